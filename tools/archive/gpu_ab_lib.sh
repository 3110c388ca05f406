# one-box A/B of two builds of the library (SLAMHIP_LIBRARY): argument 1 = the other .so; C3, C5 and C2
mkdir -p gpurun_out
other=$1
run() {  # label, env..., -- bench args
  label=$1; shift
  env "$@" 2>>gpurun_out/ab_lib.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$label syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'step_ms', round(j['ms_per_step'],4), 'value', round(j['value']), 'frac', round(j['roofline']['frac'],3), 'k', {k: round(v*1e3,1) for k,v in j['kernel_ms_per_step'].items() if v})
"
}
B="timeout -k 10 300 python bench.py --no-cpu-baseline --no-fastslam --no-pmc"
for rep in 1 2; do
run "C3 new " $B --steps 60 --warmup 5
run "C3 old " SLAMHIP_LIBRARY=$other $B --steps 60 --warmup 5
done 2>&1 | tee gpurun_out/ab_lib.log
run "C2 new " $B --steps 200 --warmup 20 --landmarks 1000 --obs 16 | tee -a gpurun_out/ab_lib.log
run "C2 old " SLAMHIP_LIBRARY=$other $B --steps 200 --warmup 20 --landmarks 1000 --obs 16 | tee -a gpurun_out/ab_lib.log
run "C5 new " $B --steps 20 --warmup 3 --landmarks 50000 --obs 8 --dtype f64 --form joseph | tee -a gpurun_out/ab_lib.log
run "C5 old " SLAMHIP_LIBRARY=$other $B --steps 20 --warmup 3 --landmarks 50000 --obs 8 --dtype f64 --form joseph | tee -a gpurun_out/ab_lib.log
