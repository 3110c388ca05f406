# round 4: switch-off experiments on the LDS-DMA pipeline (experiments build, SLAMHIP_X bits 1024 no MFMAs, 2048 no stores,
# 4096 no P loads, 8192 no fragment reads, 16384 no chunk requests; the filter's numbers stay valid: P is never written wrongly)
mkdir -p gpurun_out
export SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so
run() {
  timeout -k 10 200 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-fastslam --no-pmc --no-configs 2>>gpurun_out/r4v_exp.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'min', round(j['roofline']['min_launch_ms'],4), 'floor', round(j['roofline']['copy_floor_ms'],4))
"
}
for rep in 1 2; do
  run everything
  SLAMHIP_X=1024 run no_mfma
  SLAMHIP_X=2048 run no_stores
  SLAMHIP_X=3072 run no_stores_no_mfma
  SLAMHIP_X=6144 run no_stores_no_ploads
  SLAMHIP_X=7168 run skeleton__no_stores_ploads_mfma
  SLAMHIP_X=15360 run skeleton_without_fragment_reads
  SLAMHIP_X=23552 run skeleton_without_chunk_requests
  SLAMHIP_X=31744 run barriers_only
  SLAMHIP_X=24576 run p_traffic_and_mfma_no_panels__no_reads_no_requests_WRONG_RESULTS_IGNORED
done > gpurun_out/r4v_ab.txt 2>&1
cat gpurun_out/r4v_ab.txt
