#!/bin/bash
out=gpurun_out/exp_r02i.txt
: > $out
run() { echo "== $MTSGPU_LIB $*" >> $out; "$@" 2>>gpurun_out/exp_r02i.err | tail -1 >> $out; }
L=$PWD/mitsuba-renderer_amd
run python tools/bounce_times.py 64 1024
MTSGPU_LIB=$L/libmtsgpu_allslots.so run python tools/bounce_times.py 64 1024
run python tools/bounce_times.py 64 1024
cat $out
