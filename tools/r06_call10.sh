#!/bin/bash
# round 6, GPU call 10: ONE traversal launch per bounce (k_trace_pair, knob merged=1): films, timing
root=$(pwd); out=$root/gpurun_out/r06j; mkdir -p $out
timeout -k 10 300 python3 tools/ab_films.py 16 512 64 sync_free=0 sync_free=0,merged=1 sync_free=0,merged=1,test_retry=1 > $out/ab_films.txt 2>&1 || { cat $out/ab_films.txt; exit 1; }
cat $out/ab_films.txt
for k in "" "merged=1" "blocks_per_cu=3" "" "merged=1" "blocks_per_cu=3"; do echo "== product $k"; timeout -k 10 300 python3 tools/bounce_times.py 64 1024 $k 2>>$out/bt.err | tail -1; done > $out/bounce_times.txt 2>&1
cat $out/bounce_times.txt
