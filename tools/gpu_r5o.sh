# round 4: one-card rehearsal of the sharded bench with 2 and 4 ranks on the final code
mkdir -p gpurun_out
for n in 2 4; do
SLAM_BENCH_REHEARSE=1 timeout -k 10 300 python bench.py --gpus $n --steps 40 --warmup 4 --no-cpu-baseline --landmarks 1000 --obs 16 > gpurun_out/r5o_rehearse$n.log 2>gpurun_out/r5o_rehearse$n.err
echo "rehearse $n exit $?"
python - <<PY
import json
for l in open('gpurun_out/r5o_rehearse$n.log'):
    if l.startswith('{'):
        j=json.loads(l); f=j['fastslam']
        if 'error' in f: print('fastslam error', f['error']); continue
        print('rehearsal n_gpus', j['n_gpus'], 'peers', f['comm']['peers_attached'], 'halts', f['comm']['halts'], {k: round(v['ms_per_step']*1e3,1) for k,v in f['regimes'].items()}, 'weak', (f['weak_scaling'] or {}).get('peers_attached'), round((f['weak_scaling'] or {}).get('ms_per_step', 0)*1e3, 1))
PY
done
