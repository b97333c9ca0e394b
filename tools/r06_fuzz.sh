#!/bin/bash
# fuzz sweep: tools/r06_fuzz.sh <first seed> <count> [variant]; progress goes to stdout, the full log to gpurun_out/
root=$(pwd); out=$root/gpurun_out/r06fuzz; mkdir -p $out
lib=""; [ -n "$3" ] && lib=$root/mitsuba-renderer_amd/libmtsgpu_$3.so
MTSGPU_LIB=$lib timeout -k 10 1080 python3 tools/fuzz_parity.py $1 $2 > $out/fuzz_$1_$2_${3:-product}.txt 2> $out/fuzz_$1.err &
pid=$!
while kill -0 $pid 2>/dev/null; do sleep 30; tail -1 $out/fuzz_$1_$2_${3:-product}.txt; done
wait $pid; echo "rc=$?"; tail -2 $out/fuzz_$1_$2_${3:-product}.txt; grep -c "tree same, 0 of" $out/fuzz_$1_$2_${3:-product}.txt
