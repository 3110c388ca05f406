# round 3: the whole GPU suite with the grid form in (SLAM_GATE_AUTO: grid from 16384 landmarks), then sweep against grid
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/all_pytest.log 2>&1 || { grep -v "^  File" gpurun_out/all_pytest.log | tail -n 60 | cut -c1-400; exit 1; }
tail -n 3 gpurun_out/all_pytest.log
timeout -k 10 500 python tools/bench_gate.py > gpurun_out/bench_gate.log 2>&1 || { tail -n 30 gpurun_out/bench_gate.log; exit 1; }
cat gpurun_out/bench_gate.log
