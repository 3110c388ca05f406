# round 5: S symmetrised by s_build_kernel (no LDS pass in front of the elimination), no pass over the matrix between the last pivot
# and the ready word (y / g read the inverse's blocks where the elimination leaves them): EKF tests, timeline, bench
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_ekf.py -x -q -m gpu > gpurun_out/r5ai_tests.log 2>&1
rc=$?; echo "tests exit $rc"; tail -4 gpurun_out/r5ai_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/front_half_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5ai_timeline.txt
cat gpurun_out/r5ai_timeline.txt
FRONT_N=1000 FRONT_NZ=16 timeout -k 10 300 python tools/front_half_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5ai_timeline_c2.txt
cat gpurun_out/r5ai_timeline_c2.txt
SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so SLAMHIP_FW1=16 timeout -k 10 200 python tools/front_half_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5ai_timeline_step.txt
cat gpurun_out/r5ai_timeline_step.txt
timeout -k 10 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-fastslam > gpurun_out/r5ai_bench.log 2> gpurun_out/r5ai_bench.err
echo "bench exit $?"
python tools/show_bench.py gpurun_out/r5ai_bench.log | tail -14
