# kernel trace of a short bench run; prints the kernels of one EKF step with start/end relative to the previous down-date
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-fastslam > $GRAFT_REPO_ROOT/gpurun_out/trace.log 2>&1
python3 - <<'PY'
import csv, glob, os, re
f = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'downdate_f32' in r['Kernel_Name']]
i0, i1 = idx[9], idx[10]
t0 = int(rows[i0]['End_Timestamp'])
for r in rows[i0:i1 + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = re.match(r'(\w+)', r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')).group(1)
    print(f"{name[:26]:28s} start {(s - t0) / 1000:8.1f}  end {(e - t0) / 1000:8.1f}  dur {(e - s) / 1000:7.1f} us  queue {r.get('Queue_Id', '?')}")
PY
grep -o '"value": [0-9.]*, "unit"' $GRAFT_REPO_ROOT/gpurun_out/trace.log | head -1
