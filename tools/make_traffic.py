"""profiles/<tag>_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh:
python3 tools/make_traffic.py <tag> gpurun_out/prof_<tag>   (run where the counter CSVs are)"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
tag, d = sys.argv[1], sys.argv[2]

def total(sub, counter, prefix):
    s, disp = 0.0, set()
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter and prefix in row["Kernel_Name"]:
                s += float(row["Counter_Value"]); disp.add(row["Dispatch_Id"])
    return s, len(disp)

fetch, n1 = total("pmc_fetch", "FETCH_SIZE", "k_trace")
write, n2 = total("pmc_write", "WRITE_SIZE", "k_trace")
launches = max(n1, n2, 1)
out = {
    "workload": {"grid": 320, "res": 1024, "spp": 64},
    "kernel_source_hash": bench.kernel_source_hash(),
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tools/profile_round.sh %s -> python3 tools/pmc_workload.py 64 (production kernels)" % tag,
    "kernel": "k_trace<0,false,true> + k_trace<1,false,false> (%d launches per frame)" % launches,
    "fetch_size_kb": fetch, "write_size_kb": write, "read_correction": 2.0,
    "note": "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads 1/2 of a wide coalesced stream; for these 8/16-byte gathers the factor is uncalibrated, so x2 is an upper bound. traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024 / launches",
    "launches_per_frame": launches,
    "hbm_bytes_per_launch": (2 * fetch + write) * 1024 / launches,
    "hbm_bytes_per_launch_uncorrected": (fetch + write) * 1024 / launches,
}
path = os.path.join(ROOT, "profiles", "%s_traffic.json" % tag)
json.dump(out, open(path, "w"), indent=1)
print("wrote", path, "hbm_bytes_per_launch %.4g" % out["hbm_bytes_per_launch"])
