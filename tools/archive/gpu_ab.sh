# A/B on ONE box: bench.py with an environment variable set / unset, alternating.  usage: gpu_ab.sh VAR [reps]
mkdir -p gpurun_out
VAR=$1; REPS=${2:-3}
rm -f gpurun_out/ab.log
for i in $(seq $REPS); do
  for mode in off on; do
    if [ $mode = on ]; then export $VAR=1; else unset $VAR; fi
    timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-fastslam 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$VAR=$mode', 'step_ms', round(j['ms_per_step'],4), 'syrk_ms(events)', round(j['roofline']['avg_launch_ms'],4), 'syrk_ms(diag)', round(j['kernel_ms_per_step']['syrk'],4))
" >> gpurun_out/ab.log
  done
done
cat gpurun_out/ab.log
