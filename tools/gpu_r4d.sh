# round 4, fourth GPU call: the WHOLE GPU suite on the product library, the hang probe in its `pair` form (the filter's shape
# without the filter), the one-card rehearsal of the sharded bench (2 ranks: peers attached for the weak-scaling filter too?),
# and the panel-traffic experiment on the down-date (experiments build: every tile reads the image of tile (0, 0))
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/r4d_pytest.log 2>&1 || { grep -v "^  File" gpurun_out/r4d_pytest.log | tail -n 80 | cut -c1-500; exit 1; }
tail -n 3 gpurun_out/r4d_pytest.log
timeout -k 10 120 python tools/ipc_open_stack.py pair 2.5 15 > gpurun_out/r4d_ipc_pair.log 2>&1; echo "ipc pair exit $?"; tail -n 70 gpurun_out/r4d_ipc_pair.log | cut -c1-260
for n in 2; do
SLAM_BENCH_REHEARSE=1 SLAM_BENCH_TRACE=1 timeout -k 10 300 python bench.py --gpus $n --steps 40 --warmup 4 --no-cpu-baseline --landmarks 1000 --obs 16 > gpurun_out/r4d_rehearse$n.log 2>gpurun_out/r4d_rehearse$n.err
echo "rehearse $n exit $?"; grep "fastslam" gpurun_out/r4d_rehearse$n.err | tail -n 4
python - <<PY
import json
for l in open('gpurun_out/r4d_rehearse$n.log'):
    if l.startswith('{'):
        j=json.loads(l); f=j['fastslam']
        if 'error' in f: print('fastslam error', f['error']); continue
        print('rehearsal n_gpus', j['n_gpus'], 'comm', {k: (v if k not in ('backend', 'control_plane') else str(v)[:48]) for k, v in f['comm'].items()})
        for k,v in f['regimes'].items(): print(' ', k, round(v['ms_per_step']*1e3,1), 'us', v['resamples'])
        print(' weak', f['weak_scaling'])
PY
done
export SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so
for rep in 1 2; do for d in 32 65568 33 65569; do
  SLAMHIP_DEBUG=$d timeout -k 10 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-fastslam --no-pmc --no-configs 2>>gpurun_out/r4d_exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('dbg=$d syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'step_ms', round(j['ms_per_step'],4), 'floor', round(j['roofline']['copy_floor_ms'],4))
"
done; done > gpurun_out/r4d_exp.log 2>&1
cat gpurun_out/r4d_exp.log
