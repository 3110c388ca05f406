# round 5: per-step, per-workgroup timeline of the persistent launch (experiments build)
mkdir -p gpurun_out
make -C slam.jl_amd/csrc exp > gpurun_out/exp_build.log 2>&1 || { tail gpurun_out/exp_build.log; exit 1; }
export SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so
for cfg in "262144 16 0" "262144 4 0" "131072 16 0" "262144 16 -1"; do
  set -- $cfg
  PF_PROBE_NP=$1 PF_PROBE_K=$2 PF_PROBE_FORCE=$3 timeout -k 10 120 python tools/pf_batch_trace.py >> gpurun_out/r5r.log 2>&1 || echo "FAILED $cfg" >> gpurun_out/r5r.log
done
grep -v amdgpu.ids gpurun_out/r5r.log
