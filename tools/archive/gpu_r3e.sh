# round 3, down-date A/B of two builds (experiments libraries): the claiming grid with the P tile requested at the tile's second
# chunk (default) against its first (DD_PCH0=1)
mkdir -p gpurun_out
for rep in 1 2 3; do for lib in libslamhip_exp.so libslamhip_exp_pch0.so; do
  SLAMHIP_LIBRARY=slam.jl_amd/$lib timeout -k 10 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-fastslam --no-pmc 2>>gpurun_out/pch.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$lib syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'step_ms', round(j['ms_per_step'],4), 'value', round(j['value']), 'frac', round(j['roofline']['frac'],3), {k: round(v*1e3,1) for k,v in j['kernel_ms_per_step'].items() if v})
"
done; done > gpurun_out/pch.log 2>&1
cat gpurun_out/pch.log
