# auto-mode FastSLAM: the PF test file, then the bench (FastSLAM sub-object is what is looked at), then the two-rank rehearsal
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_gpu_pf.py -m gpu -q -x --timeout 600 > gpurun_out/pf_pytest.log 2>&1 || { tail -n 40 gpurun_out/pf_pytest.log; exit 1; }
tail -n 2 gpurun_out/pf_pytest.log
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 200 --warmup 10 > gpurun_out/bench_pf.log 2>gpurun_out/bench_pf.err || { tail -n 20 gpurun_out/bench_pf.err; exit 1; }
python - <<'PY'
import json
for l in open('gpurun_out/bench_pf.log'):
    if l.startswith('{'):
        j=json.loads(l)
        print('EKF value', round(j['value']), 'ms', round(j['ms_per_step'],4), 'syrk', round(j['roofline']['avg_launch_ms'],4))
        for k,v in j['fastslam']['regimes'].items(): print(' ', k, round(v['ms_per_step']*1e3,1), 'us', round(v['particle_steps_per_s']/1e9,3), 'G/s', v['resamples'])
        print(' roofline', j['fastslam']['roofline']['frac'])
PY
SLAM_BENCH_REHEARSE=1 timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/rehearse.log 2>gpurun_out/rehearse.err
echo "rehearse exit $?"; tail -c 600 gpurun_out/rehearse.err
python - <<'PY'
import json
for l in open('gpurun_out/rehearse.log'):
    if l.startswith('{'):
        j=json.loads(l)
        print('rehearsal n_gpus', j['n_gpus'])
        for k,v in j['fastslam']['regimes'].items(): print(' ', k, round(v['ms_per_step']*1e3,1), 'us', v['resamples'])
        print(' weak', j['fastslam']['weak_scaling'])
PY
