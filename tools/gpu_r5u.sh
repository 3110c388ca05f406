# round 5: persistent launch, fourth form (phases as functions, state in LDS, fold three quarters through the sweep)
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests/test_gpu_pf_batch.py -x -q -m gpu > gpurun_out/r5v_tests.log 2>&1
echo "tests exit $?"
tail -12 gpurun_out/r5v_tests.log
timeout -k 10 300 python tools/pf_batch_probe.py > gpurun_out/r5v_probe.log 2>&1
echo "probe exit $?"
grep -v amdgpu.ids gpurun_out/r5v_probe.log
export SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so
for cfg in "262144 16 0" "262144 16 1" "131072 16 0" "32768 16 0"; do
  set -- $cfg
  PF_PROBE_NP=$1 PF_PROBE_K=$2 PF_PROBE_FORCE=$3 timeout -k 10 120 python tools/pf_batch_trace.py >> gpurun_out/r5v_trace.log 2>&1 || echo "FAILED $cfg" >> gpurun_out/r5v_trace.log
done
grep -v amdgpu.ids gpurun_out/r5v_trace.log | awk '/^ step/ {c++; if (c % 4 != 1) next} {print}'
