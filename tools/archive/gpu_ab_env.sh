# generic one-box A/B of environment settings on the C3 bench: each argument is one "VAR=val VAR=val" setting ("-" = defaults)
mkdir -p gpurun_out
for rep in 1 2; do for cfg in "$@"; do
  if [ "$cfg" = "-" ]; then envs=""; else envs="$cfg"; fi
  env $envs timeout -k 10 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-fastslam 2>>gpurun_out/ab_env.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('[$cfg] syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'step_ms', round(j['ms_per_step'],4), 'value', round(j['value']))
"
done; done 2>&1 | tee gpurun_out/ab_env.log
