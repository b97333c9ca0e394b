#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r06c; mkdir -p $out
for k in "" "overlap=2 blocks_per_cu=3" "overlap=2 blocks_per_cu=3 dyn_div=1" "overlap=2 dyn_div=1" "overlap=2 blocks_per_cu=2" "" "overlap=2 blocks_per_cu=3" "overlap=2 blocks_per_cu=3 dyn_div=1" "overlap=2 dyn_div=1" "overlap=2 blocks_per_cu=2"; do echo "== product $k"; timeout -k 10 300 python3 tools/bounce_times.py 64 1024 $k 2>>$out/bt.err | tail -1; done > $out/bounce_times.txt 2>&1
cat $out/bounce_times.txt
