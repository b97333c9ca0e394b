#!/bin/bash
# round 6, GPU call 8: the 10 M-triangle regime (--grid 1000): bench line, L2 / fabric counters of the traversal launches; C3 for comparison
root=$(pwd); out=$root/gpurun_out/r06h; mkdir -p $out
timeout -k 10 600 python3 bench.py --grid 1000 --steps 3 --warmup 1 --no-cpu-baseline --no-1spp --no-group > $out/bench_10m.json 2> $out/bench_10m.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('$out/bench_10m.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['trace_ms_per_step'], r['shade_ms_per_step'], r['bytes_per_ray'], r['n_inner_per_ray'], r['n_leaf_per_ray'], r['n_idx_per_ray'], d['roofline_requests']['issued_breakdown_per_ray'])"
cd /tmp && export TMPDIR=/tmp
export MTSGPU_PMC_GRID=1000
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/pmc_tcc -o p -- python3 $root/tools/pmc_workload.py 64 > $out/pmc_tcc.log 2>&1
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $out/pmc_ea -o p -- python3 $root/tools/pmc_workload.py 64 > $out/pmc_ea.log 2>&1
timeout -k 10 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_mem -o p -- python3 $root/tools/pmc_workload.py 64 > $out/pmc_mem.log 2>&1
unset MTSGPU_PMC_GRID
cd $root
for d in pmc_tcc pmc_ea pmc_mem; do echo "== $d"; python3 tools/pmc_summary.py $out/$d 2>&1 | grep -A6 "k_trace"; done > $out/pmc_summary_10m.txt
grep STATS $out/pmc_tcc.log | cut -c1-400 >> $out/pmc_summary_10m.txt
find $out -name "*.db" -delete; find $out -name "*_agent_info.csv" -delete; find $out -name "*counter_collection.csv" -delete
cat $out/pmc_summary_10m.txt
