mkdir -p gpurun_out
rm -f gpurun_out/steps.log
for cfg in "20 3" "100 10" "20 3" "100 10" "300 20"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-fastslam 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('steps $1 warmup $2', 'step_ms', round(j['ms_per_step'],4), 'value', round(j['value']), 'syrk_ms(events)', round(j['roofline']['avg_launch_ms'],4))
" >> gpurun_out/steps.log
done
cat gpurun_out/steps.log
