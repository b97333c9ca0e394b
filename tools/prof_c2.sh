set -e
root=$(pwd); out=$root/gpurun_out/${1:-r05c2}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 $root/tools/config_frames.py > $out/frames.txt 2> $out/kt.log
cd $root
f=$(find $out -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv
find $out -name "*.db" -delete; find $out/kt -name "*kernel_trace.csv" -size +20M -delete
cat $out/frames.txt
python3 -c "
import csv
for r in csv.DictReader(open('$out/kernel_stats.csv')):
    n=r['Name'].split('(')[0].replace('void ','').replace('mg::','')
    if float(r['TotalDurationNs'])>2e6: print('%-28s calls %5s total %9.2f ms avg %8.3f ms' % (n[:28], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e6))
"
