#!/bin/bash
# Profile passes for one round (run on the GPU box from the repo root):  bash tools/profile_round.sh <tag>
# 1) kernel trace + stats of the bench command, 2..4) separate PMC passes over one untimed 64-spp C3 frame.
tag=${1:-r02}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-group > $out/bench_under_profiler.json 2> $out/kt.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o p -- python3 $root/tools/pmc_workload.py 64 > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o p -- python3 $root/tools/pmc_workload.py 64 > $out/pmc_write.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $out/pmc_ea -o p -- python3 $root/tools/pmc_workload.py 64 > $out/pmc_ea.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $out/pmc_sq -o p -- python3 $root/tools/pmc_workload.py 64 > $out/pmc_sq.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_mem -o p -- python3 $root/tools/pmc_workload.py 64 > $out/pmc_mem.log 2>&1
cd $root
for d in pmc_fetch pmc_write pmc_ea pmc_sq pmc_mem; do echo "== $d"; python3 tools/pmc_summary.py $out/$d 2>&1 | grep -A12 "k_trace\|k_shade"; done > $out/pmc_summary.txt
# the bench process is the one whose statistics list the replay kernel
grep -l k_replay $(find $out -name "*kernel_stats.csv") | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
python3 tools/make_traffic.py $tag $out > $out/traffic.log 2>&1; cp profiles/${tag}_traffic.json $out/ 2>/dev/null
find $out -name "*.db" -delete; find $out -name "*_agent_info.csv" -delete
ls -la $out
