#!/bin/bash
# Round 4, item 1 (i): the XCD-local queue experiment, timed and with L2 hit / miss counters per launch.
#   bash tools/exp_r04a.sh      (on the GPU box, from the repo root)
root=$(pwd)
out=$root/gpurun_out/r04a
mkdir -p $out
timeout -k 10 420 python3 tools/xcd_experiment.py > $out/xcd_timing.txt 2> $out/xcd_timing.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 420 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_tcc -o p -- python3 $root/tools/xcd_experiment.py --pmc > $out/xcd_pmc.txt 2> $out/xcd_pmc.err || exit 1
cd $root
python3 tools/xcd_join.py $out/xcd_pmc.txt $out/pmc_tcc > $out/xcd_tcc_per_order.txt 2>&1
find $out -name "*.db" -delete; find $out -name "*_agent_info.csv" -delete
timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 > $out/bench_start.json 2> $out/bench_start.err
tail -3 $out/xcd_tcc_per_order.txt; cat $out/bench_start.json
