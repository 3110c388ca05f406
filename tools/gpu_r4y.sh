# round 4: cache policy of the down-date's P stores / loads (experiments build, SLAMHIP_X bits 65536 default stores,
# 131072 sc0 stores, 262144 sc0 nt stores, 524288 default loads; the product: nt on both)
mkdir -p gpurun_out
export SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so
run() {
  timeout -k 10 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-fastslam --no-pmc --no-configs 2>>gpurun_out/r4y_exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'min', round(j['roofline']['min_launch_ms'],4), 'floor', round(j['roofline']['copy_floor_ms'],4), 'ms/step', round(j['ms_per_step'],4))
"
}
for rep in 1 2; do
  run nt_stores_nt_loads
  SLAMHIP_X=65536 run default_stores
  SLAMHIP_X=131072 run sc0_stores
  SLAMHIP_X=262144 run sc0_nt_stores
  SLAMHIP_X=524288 run default_loads
  SLAMHIP_X=589824 run default_loads_and_stores
done > gpurun_out/r4y_ab.txt 2>&1
cat gpurun_out/r4y_ab.txt
