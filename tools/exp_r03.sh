#!/bin/bash
# A/B timing of library variants on the 64-spp C3 frame:  bash tools/exp_r03.sh <out-file> <variant> [<variant> ...]
# ("product" = libmtsgpu.so); every variant runs tools/bounce_times.py 64 1024 and the whole list is run twice (drift)
out=$1; shift
: > $out
L=$PWD/mitsuba-renderer_amd
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = product ]; then lib=$L/libmtsgpu.so; else lib=$L/libmtsgpu_$v.so; fi
    echo "== $v" >> $out
    MTSGPU_LIB=$lib python tools/bounce_times.py 64 1024 2>>${out%.txt}.err | tail -1 >> $out
  done
done
cat $out
