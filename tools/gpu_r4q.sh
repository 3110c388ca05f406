# round 4: is the down-date held up by the memory system?  L2 -> fabric credit stalls and queue levels of the down-date next to
# the copy floor's kernel in the same run (product library; the LDS-DMA variant from the experiments build in a second pass)
mkdir -p gpurun_out/pmcq
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcq
run() {  # name, counters...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --no-configs --no-fastslam > $OUT/$name.log 2>&1
  echo "$name exit $?" >> $OUT/summary.log
}
run ea1 TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
run ea2 TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum GRBM_GUI_ACTIVE
export SLAMHIP_LIBRARY=$GRAFT_REPO_ROOT/slam.jl_amd/libslamhip_exp.so SLAMHIP_X=512
run ea1_dma TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
cat $OUT/summary.log
cd $GRAFT_REPO_ROOT
python3 - <<'PY' > gpurun_out/r4q_pmc.txt
import collections, csv, glob
for name in ['ea1', 'ea2', 'ea1_dma']:
    fs = glob.glob(f'gpurun_out/pmcq/{name}/**/*counter_collection.csv', recursive=True)
    if not fs:
        print(name, 'no file'); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name']
        short = 'downdate_f32_mfma' if 'downdate_f32' in k else 'tile_copy_floor' if 'tile_copy_floor' in k else None
        if short: agg[short][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(name, k, {c: round(sum(x) / len(x), 1) for c, x in v.items()}, 'launches', len(next(iter(v.values()))))
PY
cat gpurun_out/r4q_pmc.txt
rm -rf $OUT/ea1 $OUT/ea2 $OUT/ea1_dma
