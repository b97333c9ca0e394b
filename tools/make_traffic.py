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
# fabric read requests of the same launches (pmc_ea pass): every one a 128-byte request on gfx950 (TCC_EA0_RDREQ_128B = TCC_EA0_RDREQ)
ea, _ = total("pmc_ea", "TCC_EA0_RDREQ_sum", "k_trace")
ea128, _ = total("pmc_ea", "TCC_EA0_RDREQ_128B_sum", "k_trace")
# the line rates the chip sustains for random lines (tools/micro/gather_calib.hip, profiles/r06a_fetch_size_calibration.json)
calib = {}
try:
    for e in json.load(open(os.path.join(ROOT, "profiles", "r06a_fetch_size_calibration.json"))):
        calib["%s_%dMB" % (e["shape"], e["footprint_mb"])] = e["lines_per_ns"]
except Exception:
    pass
# ... and the SUSTAINED rate (k_sustained of the same benchmark: the one-load-per-thread launches above are ramp-up limited)
try:
    calib.update(json.load(open(os.path.join(ROOT, "profiles", "r06n_sustained_line_rate.json")))["G_lines_per_s"])
except Exception:
    pass
out = {
    "workload": {"grid": 320, "res": 1024, "spp": 64},
    "kernel_source_hash": bench.kernel_source_hash(),
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_EA0_RDREQ* (separate passes), tools/profile_round.sh %s -> python3 tools/pmc_workload.py 64 (production kernels)" % tag,
    "kernel": "k_trace<0,false,true> + k_trace<1,false,false> (%d launches per frame)" % launches,
    "fetch_size_kb": fetch, "write_size_kb": write, "read_correction": 2.0,
    "read_correction_source": "measured, profiles/r06a_fetch_size_calibration.txt: FETCH_SIZE tallies 64 B per fabric read request and every request is one "
                              "128-byte line (TCC_EA0_RDREQ = TCC_EA0_RDREQ_128B = TCC_MISS = 1 per line touched) for 8-byte gathers, 16-byte gathers and "
                              "16-byte-per-lane streams alike, at 32 MB, 150 MB and 2 GB footprints; requests to a line already in flight are NOT merged",
    "note": "traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024 / launches = bytes that cross the fabric (L2 misses; Infinity-Cache hits are counted, "
            "TCC_EA0_RDREQ_DRAM = TCC_EA0_RDREQ): not all of it reaches HBM -- tree and leaf records (57 MB) live in the 256 MiB Infinity Cache",
    "launches_per_frame": launches,
    "hbm_bytes_per_launch": (2 * fetch + write) * 1024 / launches,
    "hbm_bytes_per_launch_uncorrected": (fetch + write) * 1024 / launches,
    "fabric_read_requests_per_frame": ea or None, "fabric_read_requests_128B_per_frame": ea128 or None,
    "fabric_bytes_per_frame": (2 * fetch + write) * 1024,
    "random_line_rate_G_per_s": calib,
    "tcc_hit": hit, "tcc_miss": miss, "rays_per_frame": rays,
    "l2_hit_rate": hit / (hit + miss) if hit + miss else None,
    "l2_miss_per_ray": miss / rays if rays else None,
}
path = os.path.join(ROOT, "profiles", "%s_traffic.json" % tag)
json.dump(out, open(path, "w"), indent=1)
print("wrote", path, "hbm_bytes_per_launch %.4g" % out["hbm_bytes_per_launch"])
