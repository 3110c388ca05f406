# timing experiment on the particle sweep: phase stamps of the middle workgroup (libslamhip_STAMPS.so: make exp + -DPF_EXP_STAMPS)
mkdir -p gpurun_out
{
for lib in slam.jl_amd/libslamhip_STAMPS.so; do
  echo "== lib ${lib:-default}"
  SLAMHIP_LIBRARY=$lib timeout -k 10 200 python tools/pf_auto_probe.py || exit 1
done
} > gpurun_out/pfphase.log 2>&1
