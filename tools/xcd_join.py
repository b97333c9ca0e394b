"""Join the labels tools/xcd_experiment.py --pmc printed with the per-dispatch TCC counters rocprofv3 collected:
python3 tools/xcd_join.py <stdout of the run> <rocprofv3 output dir>"""
import csv, glob, os, sys, collections
labels = [l[4:].rstrip() for l in open(sys.argv[1]) if l.startswith("PMC ")]
per = collections.defaultdict(dict)
for f in glob.glob(os.path.join(sys.argv[2], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_trace" not in row["Kernel_Name"]:
            continue
        d = per[int(row["Dispatch_Id"])]
        d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
ids = sorted(per)
print("%d k_trace dispatches, %d labels" % (len(ids), len(labels)))
for lab, i in zip(labels, ids):
    h, m = per[i].get("TCC_HIT_sum", 0.0), per[i].get("TCC_MISS_sum", 0.0)
    print("%-78s TCC_HIT %.4g  TCC_MISS %.4g  hit rate %.3f  misses/ray %.2f" % (lab, h, m, h / max(h + m, 1.0), m / 8e6))
