#!/bin/bash
# Where the waves of k_trace wait: five PMC passes over one untimed 64-spp C3 frame (run on the GPU box from the repo root):
#   bash tools/pmc_round3.sh <tag>
tag=${1:-r03}
root=$(pwd)
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM --output-format csv -d $out/p1 -o p -- python3 $root/tools/pmc_workload.py 64 > $out/p1.log 2>&1
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_INSTS_VMEM --output-format csv -d $out/p2 -o p -- python3 $root/tools/pmc_workload.py 64 > $out/p2.log 2>&1
rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $out/p3 -o p -- python3 $root/tools/pmc_workload.py 64 > $out/p3.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TD_TD_BUSY_sum --output-format csv -d $out/p4 -o p -- python3 $root/tools/pmc_workload.py 64 > $out/p4.log 2>&1
rocprofv3 --pmc TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/p5 -o p -- python3 $root/tools/pmc_workload.py 64 > $out/p5.log 2>&1
cd $root
for d in p1 p2 p3 p4 p5; do echo "== $d"; python3 tools/pmc_summary.py $out/$d 2>&1 | grep -A10 "k_trace"; done > $out/pmc_summary.txt
find $out -name "*.db" -delete; find $out -name "*_agent_info.csv" -delete; find $out -name "*counter_collection.csv" -size +5M -delete
cat $out/pmc_summary.txt
