#!/bin/bash
out=gpurun_out/exp_r02j.txt
: > $out
run() { echo "== $MTSGPU_LIB $*" >> $out; "$@" 2>>gpurun_out/exp_r02j.err | tail -1 >> $out; }
L=$PWD/mitsuba-renderer_amd
run python tools/bounce_times.py 64 1024
for v in notop top256 top128s12; do MTSGPU_LIB=$L/libmtsgpu_$v.so run python tools/bounce_times.py 64 1024; done
run python tools/bounce_times.py 64 1024
run python tools/bounce_times.py 1 1024
MTSGPU_LIB=$L/libmtsgpu_notop.so run python tools/bounce_times.py 1 1024
cat $out
