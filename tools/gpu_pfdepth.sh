# one-box A/B of the particle sweep: the library against other builds (arguments), pf_auto_probe.py each, twice
mkdir -p gpurun_out
{
for rep in 1 2; do
for lib in "" "$@"; do
  echo "== lib ${lib:-default}"
  SLAMHIP_LIBRARY=$lib timeout -k 10 200 python tools/pf_auto_probe.py || exit 1
done
done
} > gpurun_out/pfdepth.log 2>&1
