# round 4: the software-pipelined one-workgroup-per-CU probe (experiments build, SLAMHIP_SP=1; timing only: P stored back unchanged,
# off-diagonal tiles only, static split) against the product library's kernel
mkdir -p gpurun_out
run() {
  timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-fastslam --no-pmc --no-configs 2>>gpurun_out/r5n_exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'min', round(j['roofline']['min_launch_ms'],4), 'floor', round(j['roofline']['copy_floor_ms'],4))
"
}
for rep in 1 2 3; do
  run product
  SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so SLAMHIP_SP=1 run software_pipelined_probe
done
tail -n 3 gpurun_out/r5n_exp.err | grep -v amdgpu
