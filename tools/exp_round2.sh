#!/bin/bash
# experiment sweep on the GPU box (round 2): prints one line per configuration
out=gpurun_out/exp_r02b.txt
: > $out
run() { echo "== $*" >> $out; "$@" 2>>gpurun_out/exp_r02b.err | tail -1 >> $out; }
# 1 spp frame: host-driven vs device-driven
run python tools/bounce_times.py 1 1024 sync_free=0 overlap=0
run python tools/bounce_times.py 1 1024 sync_free=0 overlap=1
run python tools/bounce_times.py 1 1024 sync_free=1
run python tools/bounce_times.py 1 1024 sync_free=1 chunk=16
run python tools/bounce_times.py 1 1024 sync_free=1 chunk=4
# 64 spp frame: overlap on / off, shade block 256
run python tools/bounce_times.py 64 1024 overlap=0
run python tools/bounce_times.py 64 1024 overlap=1
MTSGPU_LIB=$PWD/mitsuba-renderer_amd/libmtsgpu_sb256.so run python tools/bounce_times.py 64 1024 overlap=0
# 8 spp: device-driven on a mid-size frame
run python tools/bounce_times.py 8 1024 sync_free=0 overlap=0
run python tools/bounce_times.py 8 1024 sync_free=1
cat $out
