# round 5: does the persistent launch's per-step time depend on how long the run is?  (probe 66 us vs timeline tool 44 us at 262144)
mkdir -p gpurun_out
export SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so
for cfg in "40 20" "40 120" "200 20" "400 20" "10 10"; do
  set -- $cfg
  PF_PROBE_NP=262144 PF_PROBE_K=16 PF_PROBE_FORCE=0 PF_PROBE_WARM=$1 PF_PROBE_TIMED=$2 timeout -k 10 120 python tools/pf_batch_trace.py 2>&1 | grep "us per step" >> gpurun_out/r5x.log
done
unset SLAMHIP_LIBRARY
PF_PROBE_NPS=262144 PF_PROBE_STEPS=320 timeout -k 10 120 python tools/pf_batch_probe.py 2>&1 | grep "^n " >> gpurun_out/r5x.log
PF_PROBE_NPS=262144 PF_PROBE_STEPS=3840 timeout -k 10 120 python tools/pf_batch_probe.py 2>&1 | grep "^n " >> gpurun_out/r5x.log
cat gpurun_out/r5x.log
timeout -k 10 420 python -m pytest tests/test_gpu_pf_batch.py -x -q -m gpu 2>&1 | tail -3
