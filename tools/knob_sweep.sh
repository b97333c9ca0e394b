#!/bin/bash
# one-at-a-time sweep of k_trace's scheduling knobs on the 64-spp C3 frame (none changes a result):
#   bash tools/knob_sweep.sh gpurun_out/knobs.txt
# (refill_min set through mtsgpu_set_tuning also applies to the coherent first bounce, whose rule is 64: compare its rows with each other)
out=$1; : > $out
run() { echo "== $*" >> $out; python3 tools/bounce_times.py 64 1024 "$@" 2>>${out%.txt}.err | tail -1 >> $out; }
run
for v in 4 6 8 12 16; do run desc_min=$v; done
for v in 4 6 8 12 16; do run leaf_min=$v; done
for v in 2 3 4 6; do run dyn_div=$v; done
for v in 24 32 40; do run refill_min=$v; done
run
cat $out
