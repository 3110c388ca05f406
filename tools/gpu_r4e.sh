# round 4, fifth GPU call: after the shuffle-based cdf scan, the parallel table list / preloaded state words in the tail and the
# split of pf.hip: the FastSLAM GPU suite, the per-step probe at C4 (stamps of a resampling step), and the short bench
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_pf.py -m gpu -q -x --timeout 900 > gpurun_out/r4e_pf_pytest.log 2>&1 || { grep -v "^  File" gpurun_out/r4e_pf_pytest.log | tail -n 80 | cut -c1-500; exit 1; }
tail -n 3 gpurun_out/r4e_pf_pytest.log
for np in 262144 32768; do PF_PROBE_NP=$np timeout -k 10 200 python tools/pf_auto_probe.py 2>/dev/null | sed "s/^/np=$np /"; done > gpurun_out/r4e_probe.log
cut -c1-200 gpurun_out/r4e_probe.log
timeout -k 10 600 python bench.py --steps 100 --warmup 10 --no-pmc --no-cpu-baseline --no-configs > gpurun_out/r4e_bench.json 2> gpurun_out/r4e_bench.err || { tail -n 30 gpurun_out/r4e_bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4e_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("headline", round(d["value"]), "ms/step", round(d["ms_per_step"], 4), "dd avg", round(r["avg_launch_ms"], 4), "min", r.get("min_launch_ms"), "floor", r.get("copy_floor_ms"), "over floor", r.get("kernel_over_floor"), r.get("kernel_over_floor_min"))
f = d.get("fastslam", {})
print("fastslam", f.get("error") or {k: round(v["ms_per_step"] * 1e3, 1) for k, v in f["regimes"].items()})
PY
