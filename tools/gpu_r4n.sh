# round 4: where a wave of the LDS-DMA down-date spends its cycles (experiments build, SLAMHIP_X=512 / 1024, SLAMHIP_STAMPS=1)
mkdir -p gpurun_out
export SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so
for x in 512 1024; do
SLAMHIP_X=$x SLAMHIP_STAMPS=1 timeout -k 10 200 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-fastslam --no-pmc --no-configs > gpurun_out/r4n.log 2> gpurun_out/r4n.err
echo "X=$x"; grep "slamhip" gpurun_out/r4n.err
done
run() {
  timeout -k 10 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-fastslam --no-pmc --no-configs 2>>gpurun_out/r4n_exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'min', round(j['roofline']['min_launch_ms'],4), 'floor', round(j['roofline']['copy_floor_ms'],4), 'ms/step', round(j['ms_per_step'],4))
"
}
for rep in 1 2; do
  run product_path
  SLAMHIP_X=512 run lds_dma
  SLAMHIP_X=1024 run lds_dma_ord2
done
SLAMHIP_X=1024 timeout -k 10 600 python -m pytest tests/test_gpu_ekf.py -m gpu -q -x --timeout 600 2>&1 | tail -n 2
