# One-box A/B of the particle step: the product library against other builds (arguments: .so files, e.g. pf.hip compiled
# with -DPF_DEPTH=8 or -DPF_FAST_MATH=0 and linked with the other objects), tools/pf_auto_probe.py on each, twice.
# PF_PROBE_PROPOSAL=1 in the environment times the FastSLAM-2.0 step instead.
mkdir -p gpurun_out
{
for rep in 1 2; do
for lib in "" "$@"; do
  echo "== lib ${lib:-default}"
  SLAMHIP_LIBRARY=$lib timeout -k 10 200 python tools/pf_auto_probe.py || exit 1
done
done
} > gpurun_out/pfdepth.log 2>&1
