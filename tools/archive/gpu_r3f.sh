# round 3: the 2- and 4-rank rehearsal of the sharded bench on ONE card (plumbing + comm path; the numbers are not a scaling result)
mkdir -p gpurun_out
for n in 2 4; do
SLAM_BENCH_REHEARSE=1 SLAM_BENCH_TRACE=1 timeout -k 10 200 python bench.py --gpus $n --steps 40 --warmup 4 --no-cpu-baseline --landmarks 1000 --obs 16 > gpurun_out/rehearse$n.log 2>gpurun_out/rehearse$n.err
echo "rehearse $n exit $?"; grep "fastslam" gpurun_out/rehearse$n.err | tail -n 6
python - <<PY
import json
for l in open('gpurun_out/rehearse$n.log'):
    if l.startswith('{'):
        j=json.loads(l); f=j['fastslam']
        print('rehearsal n_gpus', j['n_gpus'], 'comm', {k: (v if k not in ('backend', 'control_plane') else str(v)[:48]) for k, v in f['comm'].items()})
        for k,v in f['regimes'].items(): print(' ', k, round(v['ms_per_step']*1e3,1), 'us', v['resamples'])
        print(' weak', f['weak_scaling'])
PY
done
