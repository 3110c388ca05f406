# round 3: the resample kernel's ancestor search through LDS -- the FastSLAM tests, then the FastSLAM leg of the bench
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_pf.py -m gpu -q -x --timeout 900 > gpurun_out/pf_pytest.log 2>&1 || { grep -v "^  File" gpurun_out/pf_pytest.log | tail -n 60 | cut -c1-400; exit 1; }
tail -n 3 gpurun_out/pf_pytest.log
timeout -k 10 600 python bench.py --landmarks 1000 --obs 16 --steps 400 --warmup 20 --no-cpu-baseline --no-pmc > gpurun_out/pf_bench.json 2> gpurun_out/pf_bench.err || { tail -n 20 gpurun_out/pf_bench.err; exit 1; }
python tools/show_bench.py gpurun_out/pf_bench.json | grep -A6 fastslam
