# round 4: the 16x16x32 pipeline, timing-faithful, in an otherwise PRODUCT library (-DDD_TIMING_16) against the product pipeline
# storing P back unchanged (-DDD_TIMING_BASE); diagonal tiles skipped in both; and the product itself
mkdir -p gpurun_out
run() {
  timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-fastslam --no-pmc --no-configs 2>>gpurun_out/r5k_exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'min', round(j['roofline']['min_launch_ms'],4), 'floor', round(j['roofline']['copy_floor_ms'],4))
"
}
for rep in 1 2 3; do
  run product
  SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_tBASE.so run shape_32x32x16_P_unchanged
  SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_t16.so run shape_16x16x32_P_unchanged
done
