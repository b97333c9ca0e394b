#!/bin/bash
out=gpurun_out/exp_r02d.txt
: > $out
run() { echo "== $MTSGPU_LIB $*" >> $out; "$@" 2>>gpurun_out/exp_r02d.err | tail -1 >> $out; }
L=$PWD/mitsuba-renderer_amd
run python tools/bounce_times.py 64 1024
for v in nt1 nt3 nt4 nt7; do MTSGPU_LIB=$L/libmtsgpu_$v.so run python tools/bounce_times.py 64 1024; done
run python tools/bounce_times.py 64 1024
cat $out
