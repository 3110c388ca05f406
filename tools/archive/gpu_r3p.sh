# round 3: the grid form of the gating -- the whole GPU suite (SLAM_GATE_AUTO now takes the grid from 512 landmarks on), then
# sweep against grid at 100 .. 50k landmarks, then the bench lines for C3 and C2
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/all_pytest.log 2>&1 || { grep -v "^  File" gpurun_out/all_pytest.log | tail -n 60 | cut -c1-400; exit 1; }
tail -n 3 gpurun_out/all_pytest.log
timeout -k 10 500 python tools/bench_gate.py > gpurun_out/bench_gate.log 2>&1 || { tail -n 30 gpurun_out/bench_gate.log; exit 1; }
cat gpurun_out/bench_gate.log
for cfg in "10000 64 c3" "1000 16 c2"; do
  set -- $cfg
  timeout -k 10 600 python bench.py --landmarks $1 --obs $2 --steps 300 --warmup 30 --no-fastslam --no-cpu-baseline --no-pmc > gpurun_out/grid_$3.json 2> gpurun_out/grid_$3.err || { tail -n 20 gpurun_out/grid_$3.err; exit 1; }
  python - gpurun_out/grid_$3.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d.get("kernel_ms_per_step"))
PY
done
