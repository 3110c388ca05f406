# round 4: the diagonal tiles at the head of the claimed lists instead of their end (product library: EKF parity tests first;
# A/B in the experiments build, SLAMHIP_DIAG_LAST=1 = round 3's order)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_ekf.py -m gpu -q -x --timeout 600 2>&1 | tail -n 2
export SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so
run() {
  timeout -k 10 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-fastslam --no-pmc --no-configs 2>>gpurun_out/r4t_exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'min', round(j['roofline']['min_launch_ms'],4), 'floor', round(j['roofline']['copy_floor_ms'],4), 'ms/step', round(j['ms_per_step'],4))
"
}
for rep in 1 2 3; do
  run diag_first
  SLAMHIP_DIAG_LAST=1 run diag_last
done > gpurun_out/r4t_ab.txt 2>&1
cat gpurun_out/r4t_ab.txt
