#!/bin/bash
# 1-spp frame time of library variants:  bash tools/exp_1spp.sh <out-file> <variant> ...
out=$1; shift; : > $out
L=$PWD/mitsuba-renderer_amd
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = product ]; then lib=$L/libmtsgpu.so; else lib=$L/libmtsgpu_$v.so; fi
  echo "== $v" >> $out; MTSGPU_LIB=$lib python tools/bounce_times.py 1 1024 2>/dev/null | tail -1 >> $out
done; done; cat $out
