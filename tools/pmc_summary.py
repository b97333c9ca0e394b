"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel: python3 tools/pmc_summary.py <dir>..."""
import csv, glob, os, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void mg::", "").replace("mg::", "")
            tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k].add(row["Dispatch_Id"])
for k in sorted(tot):
    print(k, "dispatches", len(calls[k]))
    for c in sorted(tot[k]):
        print("   %-40s %.6g" % (c, tot[k][c]))
