# rocprofv3 kernel trace + stats of the bench (argument: profile tag); summary copied to gpurun_out/prof_<tag>_stats.csv
tag=${1:-r02}
shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-pmc --no-configs "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
echo "rocprof exit $?"
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -name '*kernel_stats.csv' | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_stats.csv
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$tag          # (the raw trace: gpurun copies back at most 64 MiB)
cut -c1-150 $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_stats.csv | head -30
tail -c 3000 $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log | grep -o '"regimes".*"weak_scaling"' | head -3
