# the particle-filter GPU tests, then the FastSLAM part of the bench
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_pf.py -m gpu -q -x --timeout 400 > gpurun_out/pf_pytest.log 2>&1 || { tail -n 40 gpurun_out/pf_pytest.log; exit 1; }
tail -n 3 gpurun_out/pf_pytest.log
for e in 0 1; do
  SLAMHIP_PF_EAGER=$e timeout -k 10 300 python bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l)['fastslam']['regimes']; print('eager=$e', {k: round(v['ms_per_step'],4) for k,v in j.items()})
" || exit 1
done
