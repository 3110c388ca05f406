# round 3: front-half work (C goes out before y / g, upper triangle only; W1 split over two workgroups per row block) -- the
# EKF tests, then the C3 and C2 bench lines with the per-kernel times
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_ekf.py -m gpu -q -x --timeout 800 > gpurun_out/ekf_pytest.log 2>&1 || { grep -v "^  File" gpurun_out/ekf_pytest.log | tail -n 60 | cut -c1-400; exit 1; }
tail -n 3 gpurun_out/ekf_pytest.log
for cfg in "10000 64 c3" "1000 16 c2"; do
  set -- $cfg
  timeout -k 10 600 python bench.py --landmarks $1 --obs $2 --steps 300 --warmup 30 --no-fastslam --no-cpu-baseline --no-pmc > gpurun_out/front_$3.json 2> gpurun_out/front_$3.err || { tail -n 20 gpurun_out/front_$3.err; exit 1; }
  python - gpurun_out/front_$3.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"]), round(d["ms_per_step"] * 1e3, 1), "us/step;", {k: round(v * 1e3, 1) for k, v in d["kernel_ms_per_step"].items() if v}, d["factor_phases_us"])
PY
done
