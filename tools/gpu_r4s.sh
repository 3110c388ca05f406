# round 4: standing priorities for the two workgroups of a CU (experiments build, SLAMHIP_X 1024 / 2048)
mkdir -p gpurun_out
export SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so
run() {
  timeout -k 10 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-fastslam --no-pmc --no-configs 2>>gpurun_out/r4s_exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'min', round(j['roofline']['min_launch_ms'],4), 'floor', round(j['roofline']['copy_floor_ms'],4), 'ms/step', round(j['ms_per_step'],4))
"
}
for rep in 1 2; do
  run lds_dma
  SLAMHIP_X=1024 run prio_by_half
  SLAMHIP_X=2048 run prio_by_parity
done > gpurun_out/r4s_ab.txt 2>&1
cat gpurun_out/r4s_ab.txt
for x in 1024 2048; do
SLAMHIP_X=$x SLAMHIP_STAMPS=1 timeout -k 10 200 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-fastslam --no-pmc --no-configs > gpurun_out/r4s.log 2> gpurun_out/r4s.err
echo "X=$x"; grep "slamhip" gpurun_out/r4s.err
done
