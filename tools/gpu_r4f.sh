# round 4, sixth GPU call: the 4-pivot diagonal block of the factorisation (factor_diag_block4): the whole EKF GPU suite, then the
# one-box A/B against round 3's one pivot per MFMA (experiments build, SLAMHIP_FACTOR=pivot1) at C3 and C2, then the sharded
# soak (30 random configurations; aligned slices must equal the one-rank auto filter bit for bit)
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_ekf.py -m gpu -q -x --timeout 900 > gpurun_out/r4f_ekf_pytest.log 2>&1 || { grep -v "^  File" gpurun_out/r4f_ekf_pytest.log | tail -n 80 | cut -c1-500; exit 1; }
tail -n 3 gpurun_out/r4f_ekf_pytest.log
export SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so
for rep in 1 2; do for fac in pivot4 pivot1; do for cfg in "10000 64 60" "1000 16 400"; do
  set -- $cfg
  SLAMHIP_FACTOR=$fac timeout -k 10 200 python bench.py --landmarks $1 --obs $2 --steps $3 --warmup 10 --no-cpu-baseline --no-fastslam --no-pmc --no-configs 2>>gpurun_out/r4f_ab.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernel_ms_per_step']; print('N=$1 factor=$fac step_us', round(j['ms_per_step']*1e3,2), 'factor_us', round(k['factor']*1e3,2), 'eliminate_us', j['factor_phases_us']['eliminate'], 'w1_us', round(k['w1']*1e3,2), 'value', round(j['value']))
"
done; done; done > gpurun_out/r4f_factor_ab.log 2>&1
cat gpurun_out/r4f_factor_ab.log
unset SLAMHIP_LIBRARY
timeout -k 10 900 python tools/soak_sharded.py 30 4000 > gpurun_out/r4f_soak.log 2>&1; echo "soak exit $?"; tail -n 8 gpurun_out/r4f_soak.log | cut -c1-250
