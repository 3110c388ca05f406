# round 4: the half-tile down-date experiment (experiments build, SLAMHIP_HALF: four-wave workgroups on 64 x 128 half tiles,
# off-diagonal tiles only -- WRONG results by design, timing only) against the product kernel on the same box
mkdir -p gpurun_out
export SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so
run() {
  timeout -k 10 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-fastslam --no-pmc --no-configs 2>>gpurun_out/r4k_exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'min', round(j['roofline']['min_launch_ms'],4), 'floor', round(j['roofline']['copy_floor_ms'],4), 'ms/step', round(j['ms_per_step'],4))
"
}
for rep in 1 2; do
  run product
  SLAMHIP_DEBUG=33 run product_nostore
  SLAMHIP_HALF=2 run half_nostore
  SLAMHIP_HALF=1 run half_store_spilling
  SLAMHIP_HALF=3 run half_store_splitp
  SLAMHIP_HALF=4 run half_store_3wg
done > gpurun_out/r4k_exp.log 2>&1
cat gpurun_out/r4k_exp.log
tail -n 5 gpurun_out/r4k_exp.err
