# round 4: the LDS-DMA down-date's variants (experiments build): T1 = second row block's fragment reads behind the first one's
# MFMAs (SLAMHIP_X=1536), T2 = a tile's stores behind the next tile's first MFMAs (2560), both (3584); parity first
mkdir -p gpurun_out
export SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so
for x in 1536 2560 3584; do
SLAMHIP_X=$x timeout -k 10 600 python -m pytest tests/test_gpu_ekf.py -m gpu -q -x --timeout 600 -k "bench_workload or full_size or split_bf16 or config" 2>&1 | tail -n 2
done
run() {
  timeout -k 10 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-fastslam --no-pmc --no-configs 2>>gpurun_out/r4p_exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'min', round(j['roofline']['min_launch_ms'],4), 'floor', round(j['roofline']['copy_floor_ms'],4), 'ms/step', round(j['ms_per_step'],4))
"
}
for rep in 1 2; do
  run product_path
  SLAMHIP_X=512 run lds_dma
  SLAMHIP_X=1536 run lds_dma_T1
  SLAMHIP_X=2560 run lds_dma_T2
  SLAMHIP_X=3584 run lds_dma_T1T2
done
for x in 2560 3584; do
SLAMHIP_X=$x SLAMHIP_STAMPS=1 timeout -k 10 200 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-fastslam --no-pmc --no-configs > gpurun_out/r4p.log 2> gpurun_out/r4p.err
echo "X=$x"; grep "slamhip" gpurun_out/r4p.err
done
