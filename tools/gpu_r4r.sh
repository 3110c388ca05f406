# round 4: the LDS-DMA chunk pipeline as the product's default: the whole GPU suite, then the product library against round 3's
# register-staged pipeline (SLAMHIP_X=512) on one box
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/r4r_pytest.log 2>&1 || { grep -v "^  File" gpurun_out/r4r_pytest.log | tail -n 80 | cut -c1-500; exit 1; }
tail -n 3 gpurun_out/r4r_pytest.log
run() {
  timeout -k 10 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-fastslam --no-pmc --no-configs 2>>gpurun_out/r4r_exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'min', round(j['roofline']['min_launch_ms'],4), 'floor', round(j['roofline']['copy_floor_ms'],4), 'ms/step', round(j['ms_per_step'],4))
"
}
for rep in 1 2 3; do
  run lds_dma_default
  SLAMHIP_X=512 run register_staged
done > gpurun_out/r4r_ab.txt 2>&1
cat gpurun_out/r4r_ab.txt
