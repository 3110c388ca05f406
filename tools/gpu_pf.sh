# the particle-filter GPU tests, then the FastSLAM part of the bench
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_pf.py -m gpu -q -x --timeout 400 > gpurun_out/pf_pytest.log 2>&1 || { tail -n 40 gpurun_out/pf_pytest.log; exit 1; }
tail -n 3 gpurun_out/pf_pytest.log
