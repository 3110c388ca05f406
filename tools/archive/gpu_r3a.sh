# round 3: the particle-filter GPU tests (sharded device-side resampling, C4-shape auto test), then the IPC probe
mkdir -p gpurun_out
AMD_LOG_LEVEL=1 timeout -k 10 1000 python -m pytest tests/test_gpu_pf.py -m gpu -v -x --timeout 600 > gpurun_out/pf_pytest.log 2>&1 || { grep -v "^  File" gpurun_out/pf_pytest.log | tail -n 80 | cut -c1-400; exit 1; }
grep -c PASSED gpurun_out/pf_pytest.log; tail -n 3 gpurun_out/pf_pytest.log
timeout -k 10 120 tools/ipc_probe > gpurun_out/ipc_probe.log 2>&1; echo "probe exit $?"; tail -n 4 gpurun_out/ipc_probe.log
