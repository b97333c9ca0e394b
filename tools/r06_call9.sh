#!/bin/bash
# round 6, GPU call 9: leaf records in the order of their nodes in the device tree (MTSGPU_LEAF_ORDER=1) against index-list order
root=$(pwd); out=$root/gpurun_out/r06i; mkdir -p $out
python3 tools/ab_films.py 16 512 64 sync_free=0,save=$out/ref.npy > $out/ab_films.txt 2>&1 && MTSGPU_LEAF_ORDER=1 python3 tools/ab_films.py 16 512 64 sync_free=0,ref=$out/ref.npy sync_free=1,ref=$out/ref.npy >> $out/ab_films.txt 2>&1 || { tail -5 $out/ab_films.txt; exit 1; }
grep film $out/ab_films.txt; rm -f $out/ref.npy
for rep in 1 2; do for lo in 0 1; do for g in 320 1000; do echo "== leaf_order=$lo grid=$g"; MTSGPU_LEAF_ORDER=$lo timeout -k 10 400 python3 tools/bounce_times.py 64 1024 grid=$g 2>>$out/bt.err | tail -1; done; done; done > $out/bounce_times.txt 2>&1
cat $out/bounce_times.txt
cd /tmp && export TMPDIR=/tmp
export MTSGPU_PMC_GRID=1000 MTSGPU_LEAF_ORDER=1
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/pmc_tcc -o p -- python3 $root/tools/pmc_workload.py 64 > $out/pmc_tcc.log 2>&1
cd $root; unset MTSGPU_PMC_GRID MTSGPU_LEAF_ORDER
python3 tools/pmc_summary.py $out/pmc_tcc 2>&1 | grep -A4 "k_trace" > $out/pmc_summary_10m_leaf_order.txt; cat $out/pmc_summary_10m_leaf_order.txt
find $out -name "*.db" -delete; find $out -name "*_agent_info.csv" -delete; find $out -name "*counter_collection.csv" -delete
