# A/B/C on ONE box over values of one environment variable.  usage: gpu_abx.sh VAR "v1 v2 v3" [reps]
mkdir -p gpurun_out
VAR=$1; VALS=$2; REPS=${3:-3}
rm -f gpurun_out/abx.log
for i in $(seq $REPS); do
  for v in $VALS; do
    export $VAR=$v
    timeout -k 10 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-fastslam 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$VAR=$v', 'step_ms', round(j['ms_per_step'],4), 'syrk_ms(events)', round(j['roofline']['avg_launch_ms'],4), 'syrk_ms(diag)', round(j['kernel_ms_per_step']['syrk'],4))
" >> gpurun_out/abx.log
  done
done
sort gpurun_out/abx.log
