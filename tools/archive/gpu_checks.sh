mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/smoke.log
timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout 400 -rA > gpurun_out/pytest_gpu.log 2>&1
rc=$?
echo "pytest exit $rc" >> gpurun_out/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 400 python bench.py > gpurun_out/bench.log 2>&1
rc=$?
echo "bench exit $rc" >> gpurun_out/bench.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01zd -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
echo "rocprof exit $?" >> $GRAFT_REPO_ROOT/gpurun_out/rocprof.log
for f in $GRAFT_REPO_ROOT/gpurun_out/smoke.log $GRAFT_REPO_ROOT/gpurun_out/pytest_gpu.log $GRAFT_REPO_ROOT/gpurun_out/bench.log; do echo "== $f"; tail -n 5 $f; done
