# round 3: rehearsal of the sharded bench on ONE card after the FastSLAM leg moved under its guard: 2 and 4 ranks; plumbing only
mkdir -p gpurun_out
for cfg in "2 262144" "4 262144"; do
set -- $cfg
SLAM_BENCH_REHEARSE=1 SLAM_BENCH_TRACE=1 SLAM_BENCH_NP=$2 timeout -k 10 240 python bench.py --gpus $1 --steps 40 --warmup 4 --no-cpu-baseline --landmarks 1000 --obs 16 > gpurun_out/rehearse$1.log 2>gpurun_out/rehearse$1.err
echo "rehearse $1 ranks x $(( $2 / $1 )) particles: exit $?"
python - <<PY
import json
for l in open('gpurun_out/rehearse$1.log'):
    if l.startswith('{'):
        j=json.loads(l); f=j['fastslam']
        print('  comm', {k: (v if k not in ('backend', 'control_plane') else str(v)[:40]) for k, v in f['comm'].items()})
        print('  ', {k: round(v['ms_per_step']*1e3,1) for k,v in f['regimes'].items()}, 'resamples', {k: v['resamples'] for k,v in f['regimes'].items()})
        print('   weak', f['weak_scaling'] and round(f['weak_scaling']['ms_per_step']*1e3,1))
PY
done
