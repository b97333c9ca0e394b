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
# L2 hits and misses of the same launches (pmc_mem pass) and the frame's ray count (STATS line of tools/pmc_workload.py)
hit, _ = total("pmc_mem", "TCC_HIT_sum", "k_trace")
miss, _ = total("pmc_mem", "TCC_MISS_sum", "k_trace")
rays = None
try:
    import ast, re
    for line in open(os.path.join(d, "pmc_mem.log")):
        if line.startswith("STATS"):
            st = ast.literal_eval(line[len("STATS "):line.index(" algorithmic_bytes")])
            rays = st["rays_closest"] + st["rays_shadow"]
except Exception:
    pass
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
    "tcc_hit": hit, "tcc_miss": miss, "rays_per_frame": rays,
    "l2_hit_rate": hit / (hit + miss) if hit + miss else None,
    "l2_miss_per_ray": miss / rays if rays else None,
}
path = os.path.join(ROOT, "profiles", "%s_traffic.json" % tag)
json.dump(out, open(path, "w"), indent=1)
print("wrote", path, "hbm_bytes_per_launch %.4g" % out["hbm_bytes_per_launch"])
