# round 5: why is the sequential form (W = 1) of the persistent launch slow?  Experiments build: poll sleep, forced ways, K.
mkdir -p gpurun_out
make -C slam.jl_amd/csrc exp > gpurun_out/exp_build.log 2>&1 || { tail gpurun_out/exp_build.log; exit 1; }
export SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so
export PF_PROBE_STAMPS=1 PF_PROBE_STEPS=960
for cfg in "262144 1 16 0" "262144 8 16 0" "262144 32 16 0" "262144 1 4 0" "262144 1 1 0" "131072 1 16 1" "131072 1 16 2" "65536 1 16 1"; do
  set -- $cfg
  echo "== n $1 sleep $2 K $3 W ${4}" >> gpurun_out/r5q.log
  PF_PROBE_NPS=$1 SLAMHIP_PB_SLEEP=$2 PF_PROBE_K=$3 SLAMHIP_PB_W=$4 timeout -k 10 120 python tools/pf_batch_probe.py >> gpurun_out/r5q.log 2>&1 || echo "FAILED $cfg" >> gpurun_out/r5q.log
done
grep -v amdgpu.ids gpurun_out/r5q.log
