# new down-date defaults (band-major tile order with XCD = tile row % 8; one workgroup per tile): EKF tests, then the
# bench at C3 (split path and fp32-MFMA path), C2 and C5, old defaults beside the new ones
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ekf.py -m gpu -q -x --timeout 400 > gpurun_out/ekf_pytest.log 2>&1 || { tail -n 30 gpurun_out/ekf_pytest.log; exit 1; }
tail -n 2 gpurun_out/ekf_pytest.log
run() {  # label, env..., -- bench args
  label=$1; shift
  env "$@" 2>>gpurun_out/r2b.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$label syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'step_ms', round(j['ms_per_step'],4), 'value', round(j['value']), 'frac', round(j['roofline']['frac'],3))
"
}
B="timeout -k 10 300 python bench.py --no-cpu-baseline --no-fastslam"
for rep in 1 2; do
run "C3 new      " $B --steps 60 --warmup 5
run "C3 old      " SLAMHIP_ORDER=0 SLAMHIP_WGS=64 $B --steps 60 --warmup 5
run "C3 fp32 new " SLAMHIP_X=8 $B --steps 60 --warmup 5
run "C3 fp32 old " SLAMHIP_X=8 SLAMHIP_ORDER=0 SLAMHIP_WGS=64 $B --steps 60 --warmup 5
run "C2 new      " $B --steps 200 --warmup 20 --landmarks 1000 --obs 16
run "C2 old      " SLAMHIP_ORDER=0 SLAMHIP_WGS=64 $B --steps 200 --warmup 20 --landmarks 1000 --obs 16
done > gpurun_out/r2b.log 2>&1
run "C5 new      " $B --steps 20 --warmup 3 --landmarks 50000 --obs 8 --dtype f64 --form joseph >> gpurun_out/r2b.log 2>&1
run "C5 old      " SLAMHIP_ORDER=0 $B --steps 20 --warmup 3 --landmarks 50000 --obs 8 --dtype f64 --form joseph >> gpurun_out/r2b.log 2>&1
cat gpurun_out/r2b.log
