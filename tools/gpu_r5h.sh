# round 4: the two row blocks' MFMA chains interleaved (build-time A/B, product code otherwise; results bit-identical)
mkdir -p gpurun_out
run() {
  timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-fastslam --no-pmc --no-configs 2>>gpurun_out/r5h_exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'min', round(j['roofline']['min_launch_ms'],4), 'floor', round(j['roofline']['copy_floor_ms'],4), 'ms/step', round(j['ms_per_step'],4))
"
}
for rep in 1 2 3; do
  run chains_one_after_the_other
  SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_il.so run chains_interleaved
done > gpurun_out/r5h_ab.txt 2>&1
cat gpurun_out/r5h_ab.txt
SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_il.so timeout -k 10 600 python -m pytest tests/test_gpu_ekf.py -m gpu -q -x --timeout 600 -k "lds_dma or split_bf16 or full_size" 2>&1 | tail -n 2
