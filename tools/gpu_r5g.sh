# round 4: the step's matrix work as 24 x 16x16x32 instead of 12 x 32x32x16 bf16 MFMAs (experiments build, timing only, stores off)
mkdir -p gpurun_out
export SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so
run() {
  timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-fastslam --no-pmc --no-configs 2>>gpurun_out/r5g_exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'min', round(j['roofline']['min_launch_ms'],4), 'floor', round(j['roofline']['copy_floor_ms'],4))
"
}
for rep in 1 2; do
  SLAMHIP_X=2048 run no_stores_32x32x16
  SLAMHIP_X=1050624 run no_stores_16x16x32
  SLAMHIP_X=6144 run skeleton_mfma_32x32x16
  SLAMHIP_X=1054720 run skeleton_mfma_16x16x32
done > gpurun_out/r5g_ab.txt 2>&1
cat gpurun_out/r5g_ab.txt
for x in 6144 1054720; do
SLAMHIP_X=$x SLAMHIP_STAMPS=1 timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-fastslam --no-pmc --no-configs > gpurun_out/r5g.log 2> gpurun_out/r5g.err
echo "X=$x"; grep "slamhip" gpurun_out/r5g.err | sed 's/.*| stream/stream/'
done
