"""Where a short frame goes: kernel-by-kernel timeline from a rocprofv3 --kernel-trace CSV.
  rocprofv3 --kernel-trace --output-format csv -d <dir> -o kt -- python3 tools/bounce_times.py 1 1024
  python3 tools/timeline_1spp.py <dir>          (the LAST frame of the run is analysed)"""
import csv, glob, os, sys, collections
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void mg::", "").replace("mg::", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?")))
rows.sort()
# frames start with k_ld_tables* / k_generate: take the last k_generate as the start of the last frame
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_generate")]
i0 = starts[-1]
while i0 > 0 and rows[i0 - 1][2].startswith(("k_ld_tables", "k_sample_arrays")):
    i0 -= 1
fr = rows[i0:]
t0 = fr[0][0]; t1 = max(r[1] for r in fr)
print("last frame: %d kernels, first start -> last end %.3f ms" % (len(fr), (t1 - t0) / 1e6))
by = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, q in fr:
    by[n][0] += 1; by[n][1] += (e - s) / 1e6
for n, (c, ms) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print("  %-44s x%-3d %8.3f ms" % (n[:44], c, ms))
# time covered by at least one kernel, and the idle gaps
iv = sorted((s, e) for s, e, _, _ in fr)
busy = 0; cs, ce = iv[0]
gaps = []
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs; gaps.append(s - ce); cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print("busy (union of kernels) %.3f ms, idle %.3f ms in %d gaps (median %.1f us, max %.1f us)" % (
    busy / 1e6, (t1 - t0 - busy) / 1e6, len(gaps), sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0, max(gaps) / 1e3 if gaps else 0))
if "-v" in sys.argv:
    for s, e, n, q in fr:
        print("%9.3f %9.3f  q%s %s" % ((s - t0) / 1e6, (e - s) / 1e6, q, n[:60]))
