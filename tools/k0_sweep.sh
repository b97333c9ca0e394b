# bash tools/k0_sweep.sh <outdir> <chunk> [<chunk> ...]: k_ld_apply / k_ld_scout times per pass for several launch chunk sizes
root=$(pwd); out=$root/gpurun_out/$1; shift; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for ch in "$@"; do
  export MTSGPU_TAB_CHUNK=$ch
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$ch -o kt -- python3 $root/tools/k0_time.py 4096 192 > $out/frames_$ch.txt 2> $out/kt_$ch.log
  f=$(find $out/kt_$ch -name "*kernel_stats.csv" | head -1)
  echo "== chunk $ch: $(tr '\n' ' ' < $out/frames_$ch.txt)"; python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'k_ld_' in r['Name']: print('   %-12s calls %4s  total %9.3f ms  avg %8.3f ms' % (r['Name'].split('(')[0].replace('mg::',''), r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e6))
" 
  rm -rf $out/kt_$ch
done 2>&1 | tee $out/sweep.txt
