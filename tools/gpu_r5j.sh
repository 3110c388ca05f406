# round 4: is it the box or the library?  product library and experiments build, default path, same box
mkdir -p gpurun_out
run() {
  timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-fastslam --no-pmc --no-configs 2>>gpurun_out/r5j_exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'min', round(j['roofline']['min_launch_ms'],4), 'floor', round(j['roofline']['copy_floor_ms'],4))
"
}
for rep in 1 2; do
  run product_library
  SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so run experiments_build
done
