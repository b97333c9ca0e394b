#!/bin/bash
# round 6, GPU call 2: the any-hit launch of bounce b BEHIND the closest-hit launch of bounce b + 1 (overlap=2): films, timing
root=$(pwd); out=$root/gpurun_out/r06b; mkdir -p $out
timeout -k 10 300 python3 tools/ab_films.py 16 512 64 sync_free=0 sync_free=0,overlap=2 sync_free=0,overlap=2,overlap_delay_us=50 sync_free=0,overlap=1 > $out/ab_films.txt 2>&1 || { cat $out/ab_films.txt; exit 1; }
cat $out/ab_films.txt
for k in "" "overlap=2" "overlap=2 overlap_delay_us=20" "overlap=2 overlap_delay_us=100" "overlap=2 dyn_div=2" "" "overlap=2" "overlap=2 overlap_delay_us=20" "overlap=2 overlap_delay_us=100" "overlap=2 dyn_div=2"; do echo "== product $k"; timeout -k 10 300 python3 tools/bounce_times.py 64 1024 $k 2>>$out/bt.err | tail -1; done > $out/bounce_times.txt 2>&1
cat $out/bounce_times.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
