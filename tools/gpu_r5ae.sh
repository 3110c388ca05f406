# round 5: factor_w1_kernel with per-wave panel work (no LDS staging of C, no workgroup barrier): tests, the timeline, and the
# experiments build's switches (SLAMHIP_FW1: 1 no drain/ready inside the elimination, 2 nothing leaves before its end, 4 no W1
# stores, 8 no MFMAs; SLAMHIP_FW1_WPW: working waves per workgroup)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_ekf.py -x -q -m gpu > gpurun_out/r5ae_tests.log 2>&1
rc=$?; echo "tests exit $rc"; tail -4 gpurun_out/r5ae_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/front_half_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5ae_timeline.txt
cat gpurun_out/r5ae_timeline.txt
FRONT_N=1000 FRONT_NZ=16 timeout -k 10 300 python tools/front_half_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5ae_timeline_c2.txt
cat gpurun_out/r5ae_timeline_c2.txt
export SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so
for v in "1 0" "2 0" "4 0" "12 0" "0 8" "1 8"; do
  set -- $v
  echo "== SLAMHIP_FW1=$1 SLAMHIP_FW1_WPW=$2" >> gpurun_out/r5ae_exp.txt
  if [ "$2" = "0" ]; then SLAMHIP_FW1=$1 timeout -k 10 300 python tools/front_half_timeline.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r5ae_exp.txt
  else SLAMHIP_FW1=$1 SLAMHIP_FW1_WPW=$2 timeout -k 10 300 python tools/front_half_timeline.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r5ae_exp.txt; fi
done
cat gpurun_out/r5ae_exp.txt
unset SLAMHIP_LIBRARY
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-fastslam > gpurun_out/r5ae_bench.log 2> gpurun_out/r5ae_bench.err
echo "bench exit $?"
python tools/show_bench.py gpurun_out/r5ae_bench.log | tail -12
