# round 3: one-box A/B of the wave priorities in the split-bf16 down-date (experiments build): SLAMHIP_X=0 (a wave is
# favoured during its MFMAs: the product) against 128 (favoured while it issues memory / LDS operations)
mkdir -p gpurun_out
export SLAMHIP_LIBRARY=$PWD/slam.jl_amd/libslamhip_exp.so
for rep in 1 2 3; do
for x in 0 128; do
  SLAMHIP_X=$x timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-fastslam --no-cpu-baseline --no-pmc > gpurun_out/prio_$x.json 2> gpurun_out/prio_$x.err || { tail -n 5 gpurun_out/prio_$x.err; exit 1; }
  python - gpurun_out/prio_$x.json $x <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("SLAMHIP_X", sys.argv[2], "step %.1f us" % (d["ms_per_step"] * 1e3), "down-date %.1f us" % (d["roofline"]["avg_launch_ms"] * 1e3), flush=True)
PY
done
done
