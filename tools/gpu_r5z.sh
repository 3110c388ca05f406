# round 5: the gating as ONE launch (sparse candidate lists + last-workgroup fold): the EKF GPU tests, then a short bench
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_ekf.py -x -q -m gpu > gpurun_out/r5z_tests.log 2>&1
echo "tests exit $?"; tail -5 gpurun_out/r5z_tests.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fastslam --no-pmc > gpurun_out/r5z_bench.log 2> gpurun_out/r5z_bench.err
echo "bench exit $?"
python tools/show_bench.py gpurun_out/r5z_bench.log | head -8
python - <<'PY'
import json
for l in open('gpurun_out/r5z_bench.log'):
    if l.startswith('{'):
        j = json.loads(l)
        for c in ('C2', 'C5'):
            k = j['configs'][c]
            print(c, round(k['ms_per_step']*1e3, 1), 'us/step', {a: round(b*1e3, 1) for a, b in k['kernel_ms_per_step'].items()}, 'frac', round(k['roofline']['frac'], 3))
PY
