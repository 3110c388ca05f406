# round 5: factor_w1_kernel (6 working waves per workgroup at C3, ready word zeroed by s_build_kernel): the EKF tests with the
# fused-against-two-launches comparison, the timeline at C3 / C2, the bench with its C2 / C5 legs
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_ekf.py -x -q -m gpu > gpurun_out/r5ag_tests.log 2>&1
rc=$?; echo "tests exit $rc"; tail -4 gpurun_out/r5ag_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/front_half_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5ag_timeline.txt
cat gpurun_out/r5ag_timeline.txt
FRONT_N=1000 FRONT_NZ=16 timeout -k 10 300 python tools/front_half_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5ag_timeline_c2.txt
cat gpurun_out/r5ag_timeline_c2.txt
timeout -k 10 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-fastslam > gpurun_out/r5ag_bench.log 2> gpurun_out/r5ag_bench.err
echo "bench exit $?"
python tools/show_bench.py gpurun_out/r5ag_bench.log | tail -14
SLAMHIP_X=128 timeout -k 10 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-fastslam > gpurun_out/r5ag_bench_two.log 2> gpurun_out/r5ag_bench_two.err
echo "bench (two launches) exit $?"
python tools/show_bench.py gpurun_out/r5ag_bench_two.log | tail -14
