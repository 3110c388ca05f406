# The round's profile set: (1) rocprofv3 --kernel-trace --stats of the default bench, (2) the bench JSON line of a plain
# run (with its in-run PMC traffic), (3) the PMC passes (tools/gpu_pmc.sh) folded into one summary.  Argument: tag.
tag=${1:-r02}
mkdir -p gpurun_out
bash tools/gpu_prof.sh $tag > gpurun_out/prof_$tag.sh.log 2>&1
timeout -k 10 500 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
echo "bench exit $?"
python tools/show_bench.py gpurun_out/bench_$tag.json
rm -rf gpurun_out/pmc
bash tools/gpu_pmc.sh > gpurun_out/pmc_$tag.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc > gpurun_out/pmc_${tag}_summary.txt 2>&1
cut -c1-230 gpurun_out/pmc_${tag}_summary.txt | grep -E "downdate|pf_auto_step" 
rm -rf gpurun_out/pmc                                   # (the raw counter files: gpurun copies back at most 64 MiB)
# the other configurations' bench lines (C2: 1k landmarks, 16 obs; C5: 50k landmarks, fp64, Joseph form, 8 obs)
timeout -k 10 300 python bench.py --landmarks 1000 --obs 16 --steps 200 --warmup 20 --no-fastslam > gpurun_out/bench_${tag}_c2.json 2>/dev/null
timeout -k 10 400 python bench.py --landmarks 50000 --obs 8 --dtype f64 --form joseph --steps 20 --warmup 3 --no-fastslam > gpurun_out/bench_${tag}_c5.json 2>/dev/null
python tools/show_bench.py gpurun_out/bench_${tag}_c2.json | head -4
python tools/show_bench.py gpurun_out/bench_${tag}_c5.json | head -4
