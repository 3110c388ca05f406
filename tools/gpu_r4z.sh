# round 4: what the step structure alone costs (experiments build; as tools/gpu_r4v.sh, the fragment registers now taken as
# they are when the reads are off)
mkdir -p gpurun_out
export SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so
run() {
  timeout -k 10 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-fastslam --no-pmc --no-configs 2>>gpurun_out/r4z_exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'min', round(j['roofline']['min_launch_ms'],4), 'floor', round(j['roofline']['copy_floor_ms'],4))
"
}
for rep in 1 2; do
  SLAMHIP_X=7168 run skeleton__no_stores_ploads_mfma
  SLAMHIP_X=15360 run skeleton_without_fragment_reads
  SLAMHIP_X=23552 run skeleton_without_chunk_requests
  SLAMHIP_X=31744 run barriers_claims_and_loop_only
done > gpurun_out/r4z_ab.txt 2>&1
cat gpurun_out/r4z_ab.txt
