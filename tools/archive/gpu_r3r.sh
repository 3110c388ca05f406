# round 3: bench.py's guard around the FastSLAM leg -- the normal line, then the same run with a 0.3 s deadline (the line must
# still come out, with fastslam.error, and the exit code must be 3: a hang is not a success)
mkdir -p gpurun_out
timeout -k 10 500 python bench.py --steps 40 --warmup 10 --no-pmc --no-cpu-baseline > gpurun_out/guard_a.json 2> gpurun_out/guard_a.err || { tail -n 20 gpurun_out/guard_a.err; exit 1; }
SLAM_BENCH_PF_BUDGET_S=0.3 timeout -k 10 500 python bench.py --steps 40 --warmup 10 --no-pmc --no-cpu-baseline > gpurun_out/guard_b.json 2> gpurun_out/guard_b.err; rc=$?; echo "exit code with the deadline: $rc"; [ "$rc" = 3 ] || { echo "expected exit code 3"; exit 1; }
python - <<'PY'
import json
for f in ("gpurun_out/guard_a.json", "gpurun_out/guard_b.json"):
    lines = open(f).read().strip().splitlines()
    d = json.loads(lines[-1])
    fs = d.get("fastslam", {})
    print(f, len(lines), "line(s); value", round(d["value"]), "fastslam:", fs.get("error") or {k: round(v["ms_per_step"] * 1e3, 1) for k, v in fs["regimes"].items()})
PY
