# round 4: the gating sweep in one launch (the last workgroup folds and compacts): EKF tests, then A/B against the two-kernel form
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_ekf.py -m gpu -q -x --timeout 600 > gpurun_out/r5m_pytest.log 2>&1 || { grep -v "^  File" gpurun_out/r5m_pytest.log | tail -n 60 | cut -c1-400; exit 1; }
tail -n 2 gpurun_out/r5m_pytest.log
run() {
  timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-fastslam --no-pmc --no-configs $2 2>>gpurun_out/r5m_exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernel_ms_per_step']; print('$1 ms/step', round(j['ms_per_step'],4), 'gate', round(k['gate']*1e3,1), 'gate_final', round(k['gate_final']*1e3,1), 'value', round(j['value']))
"
}
for rep in 1 2 3; do
  run C3_one_launch ""
  SLAMHIP_X=1024 run C3_two_kernels ""
done
for rep in 1 2; do
  run C2_one_launch "--landmarks 1000 --obs 16 --steps 300"
  SLAMHIP_X=1024 run C2_two_kernels "--landmarks 1000 --obs 16 --steps 300"
done
