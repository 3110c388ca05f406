# round 4: ONE workgroup per CU instead of two (experiments build, SLAMHIP_PER_CU=1): how much of a step do a CU's two workgroups
# hide for each other?
mkdir -p gpurun_out
export SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so
run() {
  timeout -k 10 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-fastslam --no-pmc --no-configs 2>>gpurun_out/r5c_exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'min', round(j['roofline']['min_launch_ms'],4), 'floor', round(j['roofline']['copy_floor_ms'],4))
"
}
for rep in 1 2; do
  run two_per_cu
  SLAMHIP_PER_CU=1 run one_per_cu
  SLAMHIP_X=7168 run two_per_cu_skeleton
  SLAMHIP_PER_CU=1 SLAMHIP_X=7168 run one_per_cu_skeleton
  SLAMHIP_X=6144 run two_per_cu_skeleton_mfma
  SLAMHIP_PER_CU=1 SLAMHIP_X=6144 run one_per_cu_skeleton_mfma
done > gpurun_out/r5c_ab.txt 2>&1
cat gpurun_out/r5c_ab.txt
