#!/bin/bash
# round 6, final GPU call: the GPU suite and smoke on the final tree, the bench lines (C3 default, C4 on one GPU), the round's profile passes
root=$(pwd); out=$root/gpurun_out/r06final; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
timeout -k 10 300 python3 __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/smoke.log | cut -c1-200
timeout -k 10 600 python3 bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('$out/bench.json')); r=d['roofline']; print('C3', d['value'], d['ms_per_step'], 'frac', r['frac'], 'trace', r['trace_ms_per_step'], 'shade', r['shade_ms_per_step'], '1spp', d['time_to_1spp_frame_ms'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], 'fabric', r.get('fabric'))"
timeout -k 10 600 python3 bench.py --spp-total 4096 --steps 1 --warmup 0 --no-cpu-baseline --no-1spp --no-group > $out/bench_c4_1gpu.json 2> $out/bench_c4.err; echo "bench c4 rc=$?"; python3 -c "
import json; d=json.load(open('$out/bench_c4_1gpu.json')); print('C4', d['value'], d['ms_per_step'], d['scaling'])"
bash tools/profile_round.sh r06 > $out/prof.log 2>&1; tail -2 $out/prof.log
cp gpurun_out/prof_r06/r06_traffic.json $out/ 2>/dev/null
timeout -k 10 600 python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-group > $out/bench_after_profile.json 2>> $out/bench.err; python3 -c "
import json; d=json.load(open('$out/bench_after_profile.json')); r=d['roofline']; print('C3 again', d['value'], 'traffic', r['traffic'], r['traffic_source'], 'fabric', r.get('fabric'))"
