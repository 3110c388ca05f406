# round 5: the failing W = 8 case under forced ways (experiments build), then the deciding steps' timeline
mkdir -p gpurun_out
export SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so
for w in 1 2 4 8; do
  SLAMHIP_PB_W=$w timeout -k 10 200 python -m pytest tests/test_gpu_pf_batch.py -x -q -m gpu -k "5013-9-4 or 5013-2-16 or 40005-5-16" > gpurun_out/r5t_w$w.log 2>&1
  echo "forced W=$w exit $?: $(tail -1 gpurun_out/r5t_w$w.log)"
  grep -E "^E  .*(differ|vs)" gpurun_out/r5t_w$w.log | head -3
done
for cfg in "262144 16 1" "262144 16 -1" "32768 16 1"; do
  set -- $cfg
  PF_PROBE_NP=$1 PF_PROBE_K=$2 PF_PROBE_FORCE=$3 timeout -k 10 120 python tools/pf_batch_trace.py >> gpurun_out/r5t_trace.log 2>&1 || echo "FAILED $cfg" >> gpurun_out/r5t_trace.log
done
grep -v amdgpu.ids gpurun_out/r5t_trace.log | grep -v "^ step .*WG0"
