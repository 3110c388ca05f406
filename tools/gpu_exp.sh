mkdir -p gpurun_out
for d in 0 0; do
  SLAMHIP_DEBUG=$d timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>>gpurun_out/exp.log | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('dbg=$d syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'step_ms', round(j['ms_per_step'],4))
" >> gpurun_out/exp.log
done
cat gpurun_out/exp.log
