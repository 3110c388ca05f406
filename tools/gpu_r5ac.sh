# round 5: factorisation + panel + W1 in one launch with C streamed (factor_w1_kernel): the EKF tests, then the bench with the
# fused form and with round 4's two launches (SLAMHIP_X=128) on the same box
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_ekf.py -x -q -m gpu > gpurun_out/r5ac_tests.log 2>&1
rc=$?; echo "tests exit $rc"; tail -8 gpurun_out/r5ac_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-fastslam > gpurun_out/r5ac_bench.log 2> gpurun_out/r5ac_bench.err
echo "bench exit $?"
python tools/show_bench.py gpurun_out/r5ac_bench.log | tail -12
SLAMHIP_X=128 timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-fastslam > gpurun_out/r5ac_bench_two.log 2> gpurun_out/r5ac_bench_two.err
echo "bench(two launches) exit $?"
python tools/show_bench.py gpurun_out/r5ac_bench_two.log | tail -12
