# round 3, down-date: the claiming persistent grid (default, SLAMHIP_WGS=-1) against one workgroup per tile (0) and the static
# persistent grid (64 per list) -- experiments build (the knob is read there only), one box, two rounds; then the full-size
# update tests (the kernel's results) on the product library
mkdir -p gpurun_out
export SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so
for rep in 1 2; do for w in -1 0 64 96; do
  SLAMHIP_WGS=$w timeout -k 10 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-fastslam --no-pmc 2>>gpurun_out/dyn.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('wgs=$w syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'step_ms', round(j['ms_per_step'],4), 'value', round(j['value']), 'frac', round(j['roofline']['frac'],3))
"
done; done > gpurun_out/dyn.log 2>&1
cat gpurun_out/dyn.log
unset SLAMHIP_LIBRARY
timeout -k 10 900 python -m pytest tests/test_gpu_ekf.py -m gpu -q -x --timeout 600 -k "full_size or split_bf16 or config1 or observe or update" > gpurun_out/ekf_dd_pytest.log 2>&1 || { tail -n 40 gpurun_out/ekf_dd_pytest.log | cut -c1-300; exit 1; }
tail -n 3 gpurun_out/ekf_dd_pytest.log
