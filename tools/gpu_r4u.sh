# round 4: the step at which the LDS-DMA pipeline requests the P tile (build-time A/B, product code otherwise)
mkdir -p gpurun_out
run() {
  timeout -k 10 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-fastslam --no-pmc --no-configs 2>>gpurun_out/r4u_exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'min', round(j['roofline']['min_launch_ms'],4), 'floor', round(j['roofline']['copy_floor_ms'],4), 'ms/step', round(j['ms_per_step'],4))
"
}
for rep in 1 2 3; do
  run pch2_product
  for v in 3 4 6; do SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_pch$v.so run pch$v; done
done > gpurun_out/r4u_ab.txt 2>&1
cat gpurun_out/r4u_ab.txt
