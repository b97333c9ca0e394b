#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r06e; mkdir -p $out
L=$root/mitsuba-renderer_amd
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
python3 tools/ab_films.py 16 512 64 sync_free=0,save=$out/ref.npy > $out/ab_films.txt 2>&1 && MTSGPU_LIB=$L/libmtsgpu_early.so python3 tools/ab_films.py 16 512 64 sync_free=0,ref=$out/ref.npy sync_free=1,ref=$out/ref.npy >> $out/ab_films.txt 2>&1 || { cat $out/ab_films.txt; exit 1; }
cat $out/ab_films.txt; rm -f $out/ref.npy
bash tools/exp_ab.sh $out/ab_early.txt 64 product early
