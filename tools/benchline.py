import sys, json
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line); r = d["roofline"]
    print("value %.1f Msamples/s  ms/step %.1f  trace %.1f  shade %.1f  frac %.3f  achieved %.0f GB/s  1spp %.2f ms" % (
        d["value"], d["ms_per_step"], r["trace_ms_per_step"], r["shade_ms_per_step"], r["frac"], r["achieved"], d["time_to_1spp_frame_ms"] or float("nan")))
    if "cpu_baseline" in d:
        print("cpu", d["cpu_baseline"])
