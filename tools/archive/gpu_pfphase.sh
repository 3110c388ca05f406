# Timing experiment on the particle sweep: wall-clock stamps of EVERY workgroup of the last auto step (first instruction,
# map updates done, statistics stored) and of the middle workgroup's phases, printed by slam_pf_debug_stamps.
# Needs the stamps build of the library (not the product):
#   cd slam.jl_amd/csrc && make && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include \
#       -DSLAMHIP_EXPERIMENTS -DPF_EXP_STAMPS -c pf.hip -o /tmp/pf_stamps.o && \
#     hipcc --offload-arch=gfx950 -shared -fPIC -o ../libslamhip_STAMPS.so ekf_api.o ekf_gate.o ekf_strip.o ekf_update.o \
#       ekf_syrk.o /tmp/pf_stamps.o
# (add -DPF_EXP_NOOBS / -DPF_EXP_NOSTATS / -DPF_EXP_NOPREDICT for the step without its map updates / statistics / noise:
#  WRONG results, timing only)
mkdir -p gpurun_out
{
for lib in ${1:-slam.jl_amd/libslamhip_STAMPS.so}; do
  echo "== lib ${lib:-default}"
  SLAMHIP_LIBRARY=$lib timeout -k 10 200 python tools/pf_auto_probe.py || exit 1
done
} > gpurun_out/pfphase.log 2>&1
