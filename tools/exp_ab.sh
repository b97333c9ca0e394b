#!/bin/bash
# A/B timing of library variants (tools/build_variant.sh) on a C3 frame:  bash tools/exp_ab.sh <out-file> <spp> <variant> [<variant> ...]
# ("product" = libmtsgpu.so); every variant runs tools/bounce_times.py <spp> 1024 and the whole list is run twice (drift)
out=$1; spp=$2; shift; shift
: > $out
L=$PWD/mitsuba-renderer_amd
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = product ]; then lib=$L/libmtsgpu.so; else lib=$L/libmtsgpu_$v.so; fi
    echo "== $v" >> $out
    MTSGPU_LIB=$lib timeout -k 10 300 python3 tools/bounce_times.py $spp 1024 2>>${out%.txt}.err | tail -1 >> $out
  done
done
cat $out
