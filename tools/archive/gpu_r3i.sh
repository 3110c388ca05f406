# round 3: PF tests after the step-kernel changes (no pose sums, preloaded state words), then the FastSLAM regimes
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_pf.py -m gpu -q -x --timeout 600 > gpurun_out/pf_pytest.log 2>&1 || { grep -v "^  File" gpurun_out/pf_pytest.log | tail -n 60 | cut -c1-300; exit 1; }
tail -n 2 gpurun_out/pf_pytest.log
for rep in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --no-pmc --landmarks 1000 --obs 16 --steps 200 --warmup 10 > gpurun_out/bench_pf.log 2>gpurun_out/bench_pf.err || { tail -n 20 gpurun_out/bench_pf.err; exit 1; }
python - <<'PY'
import json
for l in open('gpurun_out/bench_pf.log'):
    if l.startswith('{'):
        j=json.loads(l)
        print('C2 value', round(j['value']), 'us/step', round(j['ms_per_step']*1e3,1), {k: round(v*1e3,1) for k,v in j['kernel_ms_per_step'].items() if v})
        for k,v in j['fastslam']['regimes'].items(): print(' ', k, round(v['ms_per_step']*1e3,1), 'us', round(v['particle_steps_per_s']/1e9,3), 'G/s', v['resamples'])
        print(' roofline', round(j['fastslam']['roofline']['frac'],3))
PY
done
