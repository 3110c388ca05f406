# round 4: the down-date's in-kernel clock (experiments build, SLAMHIP_STAMPS=1: s_memtime over s_memrealtime around the stream),
# for the whole kernel, without MFMAs, and the skeleton alone
mkdir -p gpurun_out
export SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so
for x in 0 1024 7168; do
SLAMHIP_X=$x SLAMHIP_STAMPS=1 timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-fastslam --no-pmc --no-configs > gpurun_out/r5d.log 2> gpurun_out/r5d.err
echo "X=$x"; grep "slamhip" gpurun_out/r5d.err | sed 's/.*| stream/stream/'
python -c "
import json
for l in open('gpurun_out/r5d.log'):
    if l.startswith('{'):
        j=json.loads(l); print('syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'min', round(j['roofline']['min_launch_ms'],4))
"
done
