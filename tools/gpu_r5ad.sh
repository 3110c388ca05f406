# round 5: timeline of factor_w1_kernel at C3 and C2; the EKF tests; the bench's kernel table
mkdir -p gpurun_out
timeout -k 10 300 python tools/front_half_timeline.py > gpurun_out/r5ad_timeline.txt 2>&1
echo "exit $?"; cat gpurun_out/r5ad_timeline.txt
FRONT_N=1000 FRONT_NZ=16 timeout -k 10 300 python tools/front_half_timeline.py > gpurun_out/r5ad_timeline_c2.txt 2>&1
echo "exit $?"; cat gpurun_out/r5ad_timeline_c2.txt
SLAMHIP_X=128 timeout -k 10 300 python tools/front_half_timeline.py > gpurun_out/r5ad_timeline_two.txt 2>&1
echo "exit $?"; cat gpurun_out/r5ad_timeline_two.txt
timeout -k 10 900 python -m pytest tests/test_gpu_ekf.py -x -q -m gpu > gpurun_out/r5ad_tests.log 2>&1
rc=$?; echo "tests exit $rc"; tail -4 gpurun_out/r5ad_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-fastslam > gpurun_out/r5ad_bench.log 2> gpurun_out/r5ad_bench.err
echo "bench exit $?"
python tools/show_bench.py gpurun_out/r5ad_bench.log | tail -12
