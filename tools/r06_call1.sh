#!/bin/bash
# round 6, GPU call 1: FETCH_SIZE calibration passes, the product's frame (after the kernels.hip split), the any-hit kernel at 3
# workgroups per CU, the GPU suite
root=$(pwd); out=$root/gpurun_out/r06a; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $out/counters_list.txt 2>&1
grep -o "TCC_EA0_RDREQ[A-Za-z0-9_]*\|TCC_EA_RDREQ[A-Za-z0-9_]*\|TCC_MISS[A-Za-z0-9_]*\|TCC_HIT[A-Za-z0-9_]*\|TCC_REQ[A-Za-z0-9_]*\|TCC_BUBBLE[A-Za-z0-9_]*\|TCC_EA0_RD[A-Za-z0-9_]*" $out/counters_list.txt | sort -u > $out/tcc_names.txt
$root/tools/micro/gather_calib > $out/calib_stdout.txt 2>&1 || exit 1
cat $out/calib_stdout.txt
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o p -- $root/tools/micro/gather_calib > $out/fetch.log 2>&1
timeout -k 10 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $out/rdreq -o p -- $root/tools/micro/gather_calib > $out/rdreq.log 2>&1
timeout -k 10 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/tcc -o p -- $root/tools/micro/gather_calib > $out/tcc.log 2>&1
cd $root
python3 tools/calib_traffic.py r06a $out > $out/calib_joined.txt 2>&1; cp profiles/r06a_fetch_size_calibration.* $out/ 2>/dev/null
find $out -name "*.db" -delete
for k in "" "blocks_per_cu=3" "" "blocks_per_cu=3"; do echo "== product $k"; timeout -k 10 300 python3 tools/bounce_times.py 64 1024 $k 2>>$out/bt.err | tail -1; done > $out/bounce_times.txt 2>&1
cat $out/bounce_times.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
