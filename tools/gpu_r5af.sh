# round 5: factor_w1_kernel without spills (ProgC by value, operands pinned): timeline + experiments switches + tests + bench
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_ekf.py -x -q -m gpu > gpurun_out/r5af_tests.log 2>&1
rc=$?; echo "tests exit $rc"; tail -4 gpurun_out/r5af_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/front_half_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5af_timeline.txt
cat gpurun_out/r5af_timeline.txt
FRONT_N=1000 FRONT_NZ=16 timeout -k 10 300 python tools/front_half_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5af_timeline_c2.txt
cat gpurun_out/r5af_timeline_c2.txt
export SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so
rm -f gpurun_out/r5af_exp.txt
for v in "1 0" "2 0" "12 0" "0 8" "0 6"; do
  set -- $v
  echo "== SLAMHIP_FW1=$1 SLAMHIP_FW1_WPW=$2" >> gpurun_out/r5af_exp.txt
  if [ "$2" = "0" ]; then SLAMHIP_FW1=$1 timeout -k 10 300 python tools/front_half_timeline.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r5af_exp.txt
  else SLAMHIP_FW1=$1 SLAMHIP_FW1_WPW=$2 timeout -k 10 300 python tools/front_half_timeline.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r5af_exp.txt; fi
done
cat gpurun_out/r5af_exp.txt
unset SLAMHIP_LIBRARY
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-fastslam > gpurun_out/r5af_bench.log 2> gpurun_out/r5af_bench.err
echo "bench exit $?"
python tools/show_bench.py gpurun_out/r5af_bench.log | tail -12
