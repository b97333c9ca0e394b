#!/bin/bash
out=gpurun_out/exp_r02h.txt
: > $out
run() { echo "== $MTSGPU_LIB $*" >> $out; "$@" 2>>gpurun_out/exp_r02h.err | tail -1 >> $out; }
run python tools/bounce_times.py 1 1024
run python tools/bounce_times.py 1 1024 plain_below=1
run python tools/bounce_times.py 1 1024 plain_below=1 desc_min=16 leaf_min=16
run python tools/bounce_times.py 1 1024 plain_below=1 desc_min=4 leaf_min=4
run python tools/bounce_times.py 1 1024 plain_below=300000
run python tools/bounce_times.py 1 1024 refill_min=48
run python tools/bounce_times.py 1 1024 refill_min=16
run python tools/bounce_times.py 1 1024 blocks_per_cu=4
cat $out
