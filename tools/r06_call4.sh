#!/bin/bash
# round 6, GPU call 4: XCD-local dealing of the traversal batches (variant xcd) -- films, timing, L2 misses; fabric counters of k_trace
root=$(pwd); out=$root/gpurun_out/r06d; mkdir -p $out
L=$root/mitsuba-renderer_amd
python3 tools/ab_films.py 16 512 64 sync_free=0,save=$out/ref.npy > $out/ab_films.txt 2>&1 && MTSGPU_LIB=$L/libmtsgpu_xcd.so python3 tools/ab_films.py 16 512 64 sync_free=0,ref=$out/ref.npy sync_free=1,ref=$out/ref.npy >> $out/ab_films.txt 2>&1 || { cat $out/ab_films.txt; exit 1; }
cat $out/ab_films.txt; rm -f $out/ref.npy
bash tools/exp_ab.sh $out/ab_xcd.txt 64 product xcd
for spp in 1; do for v in product xcd; do echo "== $v 1spp"; if [ $v = product ]; then lib=$L/libmtsgpu.so; else lib=$L/libmtsgpu_$v.so; fi; MTSGPU_LIB=$lib python3 tools/bounce_times.py 1 1024 2>/dev/null | tail -1; done; done >> $out/ab_xcd.txt
tail -4 $out/ab_xcd.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum --output-format csv -d $out/pmc_ea -o p -- python3 $root/tools/pmc_workload.py 64 > $out/pmc_ea.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/pmc_tcc -o p -- python3 $root/tools/pmc_workload.py 64 > $out/pmc_tcc.log 2>&1
export MTSGPU_LIB=$L/libmtsgpu_xcd.so
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/pmc_tcc_xcd -o p -- python3 $root/tools/pmc_workload.py 64 > $out/pmc_tcc_xcd.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum --output-format csv -d $out/pmc_ea_xcd -o p -- python3 $root/tools/pmc_workload.py 64 > $out/pmc_ea_xcd.log 2>&1
unset MTSGPU_LIB
cd $root
for d in pmc_ea pmc_tcc pmc_tcc_xcd pmc_ea_xcd; do echo "== $d"; python3 tools/pmc_summary.py $out/$d 2>&1 | grep -A6 "k_trace\|k_shade<0"; done > $out/pmc_summary.txt
find $out -name "*.db" -delete; find $out -name "*_agent_info.csv" -delete
cat $out/pmc_summary.txt
