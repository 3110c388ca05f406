# round 4, first GPU call: the new EKF tests (KAT-13 through the ABI, slam_ekf_state_written, 50 + 5 consecutive fp32 steps on
# the bench workload, config-1 replay with agree == total), then the default bench line with `configs` (C2, C5) and the copy floor
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_ekf.py -m gpu -q -x --timeout 800 -k "kat13 or state_written or consecutive or config1_replay or kats_through or gating_rules" > gpurun_out/r4a_pytest.log 2>&1 || { grep -v "^  File" gpurun_out/r4a_pytest.log | tail -n 60 | cut -c1-400; exit 1; }
tail -n 5 gpurun_out/r4a_pytest.log
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err || { tail -n 30 gpurun_out/r4a_bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4a_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("headline", round(d["value"]), "ms/step", round(d["ms_per_step"], 4), "dd ms", round(r["avg_launch_ms"], 4), "frac", round(r["frac"], 3),
      "floor", r.get("copy_floor_ms"), "over floor", r.get("kernel_over_floor"), "traffic", r.get("traffic"))
for k, v in d.get("configs", {}).items():
    if "error" in v:
        print(k, "ERROR", v["error"]); continue
    rr = v["roofline"]
    print(k, "ms/step", round(v["ms_per_step"], 4), "value", round(v["value"]), "dd ms", round(rr["avg_launch_ms"], 4), "frac", round(rr["frac"], 3),
          "floor", rr.get("copy_floor_ms"), "traffic", rr.get("traffic"), "cpu", (v.get("cpu_baseline") or {}).get("value"))
f = d.get("fastslam", {})
print("fastslam", f.get("error") or {k: round(v["ms_per_step"] * 1e3, 1) for k, v in f["regimes"].items()})
PY
