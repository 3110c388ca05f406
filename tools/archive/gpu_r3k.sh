# round 3: the EKF GPU tests after the packed diagonal side array, then the gate micro-bench and the C3 bench line
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_ekf.py -m gpu -q -x --timeout 900 > gpurun_out/ekf_pytest.log 2>&1 || { grep -v "^  File" gpurun_out/ekf_pytest.log | tail -n 60 | cut -c1-300; exit 1; }
tail -n 2 gpurun_out/ekf_pytest.log
timeout -k 10 300 python tools/bench_gate.py > gpurun_out/bench_gate.log 2>&1; tail -n 12 gpurun_out/bench_gate.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-fastslam > gpurun_out/bench_side.json 2>gpurun_out/bench_side.err; python tools/show_bench.py gpurun_out/bench_side.json | head -5
SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so timeout -k 10 200 python tools/graph_observe.py > gpurun_out/graph_observe.log 2>&1; echo "graph exit $?"; grep -v amdgpu.ids gpurun_out/graph_observe.log | tail -n 8
