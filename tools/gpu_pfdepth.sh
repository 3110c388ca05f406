# one-box A/B of the particle sweep: the library against another build (argument 1), pf_auto_probe.py each, twice
mkdir -p gpurun_out
other=${1:-slam.jl_amd/libslamhip_d1.so}
{
for lib in "" $other "" $other; do
  echo "== lib ${lib:-default}"
  SLAMHIP_LIBRARY=$lib timeout -k 10 200 python tools/pf_auto_probe.py || exit 1
done
} > gpurun_out/pfdepth.log 2>&1
