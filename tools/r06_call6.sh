#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r06f; mkdir -p $out
L=$root/mitsuba-renderer_amd
MTSGPU_LIB=$L/libmtsgpu_tf.so timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/pytest_tf.log 2>&1; echo "pytest(tf) rc=$?"; tail -4 $out/pytest_tf.log
python3 tools/ab_films.py 16 512 64 sync_free=0,save=$out/ref.npy > $out/ab_films.txt 2>&1 && MTSGPU_DEBUG=1 MTSGPU_LIB=$L/libmtsgpu_tf.so python3 tools/ab_films.py 16 512 64 sync_free=0,ref=$out/ref.npy sync_free=1,ref=$out/ref.npy >> $out/ab_films.txt 2>&1 || { tail -5 $out/ab_films.txt; exit 1; }
grep "film\|tail filter" $out/ab_films.txt | head; rm -f $out/ref.npy
bash tools/exp_ab.sh $out/ab_tf.txt 64 product tf
for v in product tf; do echo "== $v 1spp"; if [ $v = product ]; then lib=$L/libmtsgpu.so; else lib=$L/libmtsgpu_$v.so; fi; MTSGPU_LIB=$lib python3 tools/bounce_times.py 1 1024 2>/dev/null | tail -1; done >> $out/ab_tf.txt
MTSGPU_LIB=$L/libmtsgpu_tf.so MTSGPU_DEBUG=1 python3 tools/bounce_times.py 16 1024 count=1 2>&1 | grep "tail\|lanes\|flagged" > $out/tf_counts.txt; cat $out/tf_counts.txt
tail -4 $out/ab_tf.txt
