# A/B of the down-date's tile order on one box (SLAMHIP_ORDER: 0 super-rows per XCD, 1 band-major dealt round-robin, 2 band-major with XCD = tile row % 8)
mkdir -p gpurun_out
for rep in 1 2; do for o in 0 1 2; do
  SLAMHIP_ORDER=$o timeout -k 10 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-fastslam 2>>gpurun_out/order.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('order=$o syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'step_ms', round(j['ms_per_step'],4), 'value', round(j['value']))
"
done; done > gpurun_out/order.log 2>&1
cat gpurun_out/order.log
