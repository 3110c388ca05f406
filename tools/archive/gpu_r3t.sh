# round 3: kernel durations of the gating forms from the profiler (the event timers of bench_gate.py include the launch gaps)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/gateprof -o gate -- python3 $GRAFT_REPO_ROOT/tools/bench_gate.py > $GRAFT_REPO_ROOT/gpurun_out/gateprof.log 2>&1 || { tail -n 20 $GRAFT_REPO_ROOT/gpurun_out/gateprof.log; exit 1; }
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/gateprof -name "*kernel_stats.csv" | head -n 1)
[ -n "$f" ] || { echo "no kernel_stats.csv"; exit 1; }
grep -E "gate|grid|Name" "$f" < /dev/null | cut -c1-260
