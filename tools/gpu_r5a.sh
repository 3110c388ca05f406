# round 4: the claimed list position handed round one step earlier (product) against at the step it is needed (DD_DMA_RC=0)
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ekf.py -m gpu -q -x --timeout 600 2>&1 | tail -n 2
run() {
  timeout -k 10 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-fastslam --no-pmc --no-configs 2>>gpurun_out/r5a_exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'min', round(j['roofline']['min_launch_ms'],4), 'floor', round(j['roofline']['copy_floor_ms'],4), 'ms/step', round(j['ms_per_step'],4))
"
}
for rep in 1 2 3; do
  run claim_one_step_early
  SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_rc0.so run claim_at_the_step
done > gpurun_out/r5a_ab.txt 2>&1
cat gpurun_out/r5a_ab.txt
