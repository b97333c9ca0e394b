#!/bin/bash
# round 6, GPU call 13: the closest-hit kernel without the mailbox (host-driven bounces; tied rays traced again with it)
root=$(pwd); out=$root/gpurun_out/r06m; mkdir -p $out
timeout -k 10 300 python3 tools/ab_films.py 16 512 64 sync_free=0,mailbox_free=0 sync_free=0,mailbox_free=1 sync_free=0,mailbox_free=1,test_retry=1 sync_free=0,mailbox_free=1,merged=1 > $out/ab_films.txt 2>&1 || { cat $out/ab_films.txt; exit 1; }
cat $out/ab_films.txt
for k in "mailbox_free=0" "mailbox_free=1" "mailbox_free=0" "mailbox_free=1"; do echo "== product $k"; timeout -k 10 300 python3 tools/bounce_times.py 64 1024 $k 2>>$out/bt.err | tail -1; done > $out/bounce_times.txt 2>&1
cat $out/bounce_times.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
