mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_ekf.py -m gpu -q -x --timeout 900 > gpurun_out/ekf_pytest.log 2>&1 || { grep -v "^  File" gpurun_out/ekf_pytest.log | tail -n 60 | cut -c1-300; exit 1; }
tail -n 2 gpurun_out/ekf_pytest.log
for rep in 1 2; do timeout -k 10 300 python bench.py --landmarks 1000 --obs 16 --steps 300 --warmup 20 --no-fastslam --no-cpu-baseline --no-pmc > gpurun_out/bench_c2.json 2>/dev/null; python tools/show_bench.py gpurun_out/bench_c2.json | head -4 | cut -c1-220; done
