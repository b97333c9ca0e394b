"""Calibration of rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ* for the access shapes of k_trace (tools/micro/gather_calib.hip):
joins the per-dispatch counter CSVs of the passes under <dir>/{fetch,rdreq,tcc} with the byte counts the benchmark
printed (<dir>/calib_stdout.txt) and writes profiles/<tag>_fetch_size_calibration.{txt,json}.
    python3 tools/calib_traffic.py <tag> <dir>"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, d = sys.argv[1], sys.argv[2]
REPS = 4


def dispatches(sub):
    """{counter: [(dispatch id, kernel, value)] in dispatch order}"""
    out = {}
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            out.setdefault(row["Counter_Name"], {}).setdefault(int(row["Dispatch_Id"]), [row["Kernel_Name"], 0.0])[1] += float(row["Counter_Value"])
    return {c: [(i, k, v) for i, (k, v) in sorted(m.items())] for c, m in out.items()}


expected = []
for line in open(os.path.join(d, "calib_stdout.txt")):
    p = line.split()
    if len(p) == 10 and p[0].startswith("k_"):
        expected.append(dict(shape=p[0], footprint_mb=int(p[1]), units=int(p[2]), lines=int(p[3]), bytes_if_128=int(p[4]),
                             bytes_if_64=int(p[5]), bytes_if_32=int(p[6]), ms_first=float(p[7]), ms_rest=float(p[8]), lines_per_ns=float(p[9])))
passes = {sub: dispatches(sub) for sub in ("fetch", "rdreq", "tcc") if os.path.isdir(os.path.join(d, sub))}
rows = []
for sub, counters in passes.items():
    for cname, lst in counters.items():
        per_shape = {}
        for _, k, v in lst:
            for s in ("k_g16", "k_g8", "k_g48", "k_stream"):
                if s + "(" in k or k.startswith(s) or ("::" + s) in k:
                    per_shape.setdefault(s, []).append(v)
        for e in expected:
            vals = per_shape.get(e["shape"], [])
            fi = [0, 1, 2][[32, 150, 2048].index(e["footprint_mb"])]
            grp = vals[fi * REPS:(fi + 1) * REPS]
            if len(grp) == REPS:
                e.setdefault("counters", {})[cname] = grp
lines_out = ["# %s: FETCH_SIZE / TCC_EA0_RDREQ* against known byte counts (tools/micro/gather_calib.hip; one MI355X)" % tag,
             "# every launch touches every 128-byte line of its footprint once; 4 launches back to back per row: first | mean of the other three",
             "# FETCH_SIZE is in KB (x 1024 below); 'per line' = counter bytes / lines touched"]
for e in expected:
    c = e.get("counters", {})
    s = "%-8s %5d MB  lines %9d  %.3f lines/ns" % (e["shape"], e["footprint_mb"], e["lines"], e["lines_per_ns"])
    if "FETCH_SIZE" in c:
        f = [v * 1024.0 for v in c["FETCH_SIZE"]]
        e["fetch_bytes_per_line_first"] = f[0] / e["lines"]; e["fetch_bytes_per_line_rest"] = sum(f[1:]) / 3 / e["lines"]
        s += "  FETCH_SIZE/line %6.1f | %6.1f B" % (e["fetch_bytes_per_line_first"], e["fetch_bytes_per_line_rest"])
    for cn in sorted(c):
        if cn == "FETCH_SIZE": continue
        v = c[cn]
        s += "  %s/line %.3f | %.3f" % (cn, v[0] / e["lines"], sum(v[1:]) / 3 / e["lines"])
    lines_out.append(s)
txt = "\n".join(lines_out) + "\n"
open(os.path.join(ROOT, "profiles", "%s_fetch_size_calibration.txt" % tag), "w").write(txt)
json.dump(expected, open(os.path.join(ROOT, "profiles", "%s_fetch_size_calibration.json" % tag), "w"), indent=1)
sys.stdout.write(txt)
