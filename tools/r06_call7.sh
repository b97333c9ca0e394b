#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r06g; mkdir -p $out
L=$root/mitsuba-renderer_amd
python3 tools/ab_films.py 16 512 64 sync_free=0,save=$out/ref.npy > $out/ab_films.txt 2>&1 && for v in tf tfs; do MTSGPU_LIB=$L/libmtsgpu_$v.so python3 tools/ab_films.py 16 512 64 sync_free=0,ref=$out/ref.npy sync_free=1,ref=$out/ref.npy >> $out/ab_films.txt 2>&1 || { tail -5 $out/ab_films.txt; exit 1; }; done
grep "film" $out/ab_films.txt | head; rm -f $out/ref.npy
bash tools/exp_ab.sh $out/ab_tf.txt 64 product tf tfs
