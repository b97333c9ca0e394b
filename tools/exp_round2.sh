#!/bin/bash
out=gpurun_out/exp_r02c.txt
: > $out
run() { echo "== $MTSGPU_LIB $*" >> $out; "$@" 2>>gpurun_out/exp_r02c.err | tail -1 >> $out; }
L=$PWD/mitsuba-renderer_amd
run python tools/bounce_times.py 64 1024
for v in pd12 pd24 pl12 pl24; do MTSGPU_LIB=$L/libmtsgpu_$v.so run python tools/bounce_times.py 64 1024; done
run python tools/bounce_times.py 64 1024
for k in 1 2 4; do run python tools/group_1spp.py $k; done
cat $out
