# the particle-filter GPU tests, then the 2-rank rehearsal of the bench on one card (gloo): shared-memory scalars off / on
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_pf.py -m gpu -q -x --timeout 400 > gpurun_out/pf_pytest.log 2>&1 || { tail -n 40 gpurun_out/pf_pytest.log; exit 1; }
tail -n 3 gpurun_out/pf_pytest.log
for shm in 0 1; do
export SLAMHIP_SHM_SCALARS=$shm
RANKS=${RANKS:-2} bash tools/gpu_rehearse.sh > /dev/null
python - <<'PY'
import json, os
for l in open('gpurun_out/rehearse.log'):
    if l.startswith('{'):
        j = json.loads(l)['fastslam']
        print('shm', os.environ['SLAMHIP_SHM_SCALARS'], {k: round(v['ms_per_step'], 4) for k, v in j['regimes'].items()}, 'weak', j['weak_scaling'] and round(j['weak_scaling']['ms_per_step'], 4))
print(open('gpurun_out/rehearse.log').read()[-20:])
PY
done
