# round 4: the LDS-DMA pipeline without a barrier per step (arrive / wait counters in LDS; experiments build, SLAMHIP_X=32768)
mkdir -p gpurun_out
export SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so
SLAMHIP_X=32768 timeout -k 10 600 python -m pytest tests/test_gpu_ekf.py -m gpu -q -x --timeout 600 -k "not lds_dma and not split_bf16 and not gating" 2>&1 | tail -n 3
run() {
  timeout -k 10 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-fastslam --no-pmc --no-configs 2>>gpurun_out/r4w_exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'min', round(j['roofline']['min_launch_ms'],4), 'floor', round(j['roofline']['copy_floor_ms'],4), 'ms/step', round(j['ms_per_step'],4))
"
}
for rep in 1 2 3; do
  run lds_dma
  SLAMHIP_X=32768 run lds_dma_no_barrier
done > gpurun_out/r4w_ab.txt 2>&1
cat gpurun_out/r4w_ab.txt
