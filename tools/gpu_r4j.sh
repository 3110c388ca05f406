# round 4, final GPU call: the whole GPU suite, the round's profile set on the final code (r04b), the down-date's switch-off
# experiments on one box (experiments build; WRONG results by design: stores off keeps the filter valid, the others are run
# with stores off as well), and the one-card rehearsal of the sharded bench with 2 and 4 ranks
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/r4j_pytest.log 2>&1 || { grep -v "^  File" gpurun_out/r4j_pytest.log | tail -n 80 | cut -c1-500; exit 1; }
tail -n 3 gpurun_out/r4j_pytest.log
bash tools/gpu_profile_round.sh r04b 2>&1 | cut -c1-200 | grep -v "^sq\|^tcc\|^grbm" | tail -n 40
export SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so
for rep in 1 2; do for d in 33 35 37 39 65569; do
  SLAMHIP_DEBUG=$d timeout -k 10 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-fastslam --no-pmc --no-configs 2>>gpurun_out/r4j_exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('dbg=$d syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'min', round(j['roofline']['min_launch_ms'],4), 'floor', round(j['roofline']['copy_floor_ms'],4))
"
done; done > gpurun_out/r4j_exp.log 2>&1
cat gpurun_out/r4j_exp.log
unset SLAMHIP_LIBRARY
for n in 2 4; do
SLAM_BENCH_REHEARSE=1 timeout -k 10 300 python bench.py --gpus $n --steps 40 --warmup 4 --no-cpu-baseline --landmarks 1000 --obs 16 > gpurun_out/r4j_rehearse$n.log 2>gpurun_out/r4j_rehearse$n.err
echo "rehearse $n exit $?"
python - <<PY
import json
for l in open('gpurun_out/r4j_rehearse$n.log'):
    if l.startswith('{'):
        j=json.loads(l); f=j['fastslam']
        if 'error' in f: print('fastslam error', f['error']); continue
        print('rehearsal n_gpus', j['n_gpus'], 'peers', f['comm']['peers_attached'], 'halts', f['comm']['halts'], {k: round(v['ms_per_step']*1e3,1) for k,v in f['regimes'].items()}, 'weak', (f['weak_scaling'] or {}).get('peers_attached'), round((f['weak_scaling'] or {}).get('ms_per_step', 0)*1e3, 1))
PY
done
