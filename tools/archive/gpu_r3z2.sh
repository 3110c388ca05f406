# round 3: kernel durations of the FastSLAM regimes from the profiler
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pfprof -o pf -- python3 $GRAFT_REPO_ROOT/bench.py --landmarks 1000 --obs 16 --steps 300 --warmup 20 --no-cpu-baseline --no-pmc > $GRAFT_REPO_ROOT/gpurun_out/pfprof.log 2>&1 || { tail -n 20 $GRAFT_REPO_ROOT/gpurun_out/pfprof.log; exit 1; }
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pfprof/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    for key in ("pf_auto_resample", "pf_auto_scan1", "pf_auto_step_kernel<float, false", "pf_auto_step_kernel<float, true"):
        if key in n:
            agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    act = [d for d in v if d > 3.5]
    print(f"{k}: {len(v)} launches, {len(act)} above 3.5 us: mean {sum(act)/max(len(act),1):.1f} us, median {sorted(act)[len(act)//2] if act else 0:.1f}, max {max(act) if act else 0:.1f}")
PY
