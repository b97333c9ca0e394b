#!/bin/bash
# round 6, GPU call 14: what one more L2 MISS per leaf visit costs k_trace, against one more L2 HIT (sensitivity probe)
root=$(pwd); out=$root/gpurun_out/r06n; mkdir -p $out
bash tools/exp_ab.sh $out/ab_miss.txt 64 product miss3 miss1 miss2
cd /tmp && export TMPDIR=/tmp
for v in miss1 miss2 miss3; do MTSGPU_LIB=$root/mitsuba-renderer_amd/libmtsgpu_$v.so timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $out/pmc_$v -o p -- python3 $root/tools/pmc_workload.py 64 > $out/pmc_$v.log 2>&1; done
cd $root
for v in miss1 miss2 miss3; do echo "== $v"; python3 tools/pmc_summary.py $out/pmc_$v 2>&1 | grep -A4 "k_trace"; done > $out/pmc_summary.txt; cat $out/pmc_summary.txt
find $out -name "*.db" -delete; find $out -name "*_agent_info.csv" -delete; find $out -name "*counter_collection.csv" -delete
