# round 5: s_build_kernel without the scratch copy of the observation model: EKF tests, bench kernel table (twice)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_ekf.py -x -q -m gpu > gpurun_out/r5aj_tests.log 2>&1
rc=$?; echo "tests exit $rc"; tail -3 gpurun_out/r5aj_tests.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2; do
timeout -k 10 500 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-pmc --no-fastslam > gpurun_out/r5aj_bench$i.log 2> gpurun_out/r5aj_bench$i.err
python tools/show_bench.py gpurun_out/r5aj_bench$i.log | grep -E "^value|^kernels"
done
