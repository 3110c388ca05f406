#!/usr/bin/env python3
"""Print the interesting parts of a bench.py JSON line (argument: log file)."""
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        j = json.loads(l)
        print("value", round(j["value"]), j["unit"][:12], "| ms/step", round(j["ms_per_step"], 4), "| n_gpus", j["n_gpus"])
        r = j["roofline"]
        print("roofline", {k: r.get(k) for k in ("bound", "achieved", "peak", "frac", "traffic", "avg_launch_ms", "stdev_ms", "launches", "avg_launch_ms_timed_region", "copy_floor_ms", "kernel_over_floor")})
        print(" ", j.get("traffic_note"))
        print("kernels", {k: round(v * 1e3, 1) for k, v in j["kernel_ms_per_step"].items()})
        if "other_kernels" in j:
            print("other kernels (us)", {k: round(v, 1) for k, v in j["other_kernels"].items()})
        if "cpu_baseline" in j:
            c = j["cpu_baseline"]
            print("cpu", round(c["value"], 1), c["unit"], c["cores"], c["kind"], "| gpu/cpu", round(j.get("gpu_over_cpu", 0)))
            for leg in c.get("other_legs", []):
                print("   leg", {k: leg[k] for k in leg if k != "sample" and k != "note"})
        f = j.get("fastslam")
        if f:
            print("fastslam value", round(f["value"] / 1e9, 3), "G/s | frac", round(f["roofline"]["frac"], 3), "| traffic", f["roofline"]["traffic"])
            for k, v in f["regimes"].items():
                print("   ", k, round(v["ms_per_step"] * 1e3, 1), "us", round(v["particle_steps_per_s"] / 1e9, 3), "G/s", v["resamples"])
            if f.get("weak_scaling"):
                print("    weak", f["weak_scaling"])
