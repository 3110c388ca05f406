# round 4: (a) where in the tile the P tile is requested (build-time A/B, chunk 8 - DD_POFF8), after the claim word's flat
# accesses and the atomic optimizer's wait were removed; (b) the copy floor with the write one unit behind the read
mkdir -p gpurun_out
run() {
  timeout -k 10 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-fastslam --no-pmc --no-configs 2>>gpurun_out/r4l_exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'min', round(j['roofline']['min_launch_ms'],4), 'floor', round(j['roofline']['copy_floor_ms'],4), 'ms/step', round(j['ms_per_step'],4))
"
}
for rep in 1 2; do
  run product_poff7
  for v in 1 2 4 8; do SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_poff$v.so run poff$v; done
  SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so run exp_floor_plain
  SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so SLAMHIP_COPY_LAG=1 run exp_floor_lag1
done > gpurun_out/r4l_exp.log 2>&1
cat gpurun_out/r4l_exp.log
