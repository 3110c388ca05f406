# round 3: what the W1 kernel's epilogue costs (experiments build, timing only: the numbers are wrong with stores off)
mkdir -p gpurun_out
export SLAMHIP_LIBRARY=$PWD/slam.jl_amd/libslamhip_exp.so
for dbg in 0 1 2 3 0; do
  SLAMHIP_W1DBG=$dbg timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-fastslam --no-cpu-baseline --no-pmc > gpurun_out/w1dbg_$dbg.json 2> gpurun_out/w1dbg_$dbg.err || { tail -n 5 gpurun_out/w1dbg_$dbg.err; }
  python - gpurun_out/w1dbg_$dbg.json $dbg <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("SLAMHIP_W1DBG", sys.argv[2], {k: round(v * 1e3, 1) for k, v in d["kernel_ms_per_step"].items() if v})
PY
done
