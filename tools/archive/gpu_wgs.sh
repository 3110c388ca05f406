# A/B: persistent (64 workgroups per XCD list) against more, shorter-lived workgroups handed out by the hardware dispatcher
mkdir -p gpurun_out
for rep in 1 2; do for cfg in "64 0" "64 1" "128 1" "256 1" "512 1" "2000 1" "2000 0"; do
  set -- $cfg
  SLAMHIP_ORDER=2 SLAMHIP_WGS=$1 SLAMHIP_X=$2 timeout -k 10 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-fastslam 2>>gpurun_out/wgs.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('wgs=$1 x=$2 syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'step_ms', round(j['ms_per_step'],4), 'value', round(j['value']))
"
done; done > gpurun_out/wgs.log 2>&1
cat gpurun_out/wgs.log
