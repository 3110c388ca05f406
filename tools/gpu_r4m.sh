# round 4: the LDS-DMA chunk pipeline of the down-date (experiments build, SLAMHIP_X=512): parity tests first, then A/B
mkdir -p gpurun_out
export SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so
SLAMHIP_X=512 timeout -k 10 600 python -m pytest tests/test_gpu_ekf.py -m gpu -q -x --timeout 600 > gpurun_out/r4m_pytest.log 2>&1 || { grep -v "^  File" gpurun_out/r4m_pytest.log | tail -n 40 | cut -c1-300; exit 1; }
tail -n 3 gpurun_out/r4m_pytest.log
run() {
  timeout -k 10 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-fastslam --no-pmc --no-configs 2>>gpurun_out/r4m_exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'min', round(j['roofline']['min_launch_ms'],4), 'floor', round(j['roofline']['copy_floor_ms'],4), 'ms/step', round(j['ms_per_step'],4))
"
}
for rep in 1 2 3; do
  run product_path
  SLAMHIP_X=512 run lds_dma
done > gpurun_out/r4m_exp.log 2>&1
cat gpurun_out/r4m_exp.log
