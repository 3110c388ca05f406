# round 2, first GPU call: the whole GPU suite (incl. the new full-size C5 test and KAT-8..10 through the ABI), the bench
# line, and the self-launching two-rank rehearsal on one card.
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/smoke.log
timeout -k 10 1100 python -m pytest tests -m gpu -q --timeout 600 -rA --durations=8 > gpurun_out/pytest_gpu.log 2>&1
rc=$?
echo "pytest exit $rc" >> gpurun_out/pytest_gpu.log
tail -n 25 gpurun_out/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 400 python bench.py > gpurun_out/bench.log 2>&1
rc=$?
echo "bench exit $rc" >> gpurun_out/bench.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
SLAM_BENCH_REHEARSE=1 timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/rehearse.log 2>&1
echo "rehearse exit $?" >> gpurun_out/rehearse.log
for f in gpurun_out/smoke.log gpurun_out/bench.log gpurun_out/rehearse.log; do echo "== $f"; tail -c 1500 $f; done
