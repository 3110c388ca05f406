# round 4: the pipeline on the 16x16x32 MFMA shape, timing-faithful (experiments build, SLAMHIP_X=2097152: 14 fragment reads in two
# batches, 24 MFMAs, the P patch as 8 + 8 dwordx4 operations, P stored back unchanged) against the 32x32x16 pipeline storing P
# back unchanged as well (SLAMHIP_X=4194304); diagonal tiles skipped in both
mkdir -p gpurun_out
export SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so
run() {
  timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-fastslam --no-pmc --no-configs 2>>gpurun_out/r5i_exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'min', round(j['roofline']['min_launch_ms'],4), 'floor', round(j['roofline']['copy_floor_ms'],4))
"
}
for rep in 1 2 3; do
  run product_everything
  SLAMHIP_X=4194304 run shape_32x32x16_P_unchanged
  SLAMHIP_X=2097152 run shape_16x16x32_P_unchanged
done > gpurun_out/r5i_ab.txt 2>&1
cat gpurun_out/r5i_ab.txt
tail -n 3 gpurun_out/r5i_exp.err | grep -v amdgpu
