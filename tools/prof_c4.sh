set -e
root=$(pwd); out=$root/gpurun_out/${1:-r05j}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
C4_REPS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 $root/tools/c4_frame.py 4096 > $out/c4_frame.txt 2> $out/kt.log
cd $root
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
find $out -name "*.db" -delete; find $out -name "*_agent_info.csv" -delete; find $out/kt -name "*kernel_trace.csv" -size +20M -delete
cat $out/c4_frame.txt; cut -c1-200 $out/kernel_stats.csv | head -30
