#!/bin/bash
out=gpurun_out/exp_r02e.txt
: > $out
run() { echo "== $MTSGPU_LIB $*" >> $out; "$@" 2>>gpurun_out/exp_r02e.err | tail -1 >> $out; }
for b in 0 6 5 4 3; do run python tools/bounce_times.py 64 1024 blocks_per_cu=$b; done
cat $out
