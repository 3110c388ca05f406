# round 3: the whole GPU suite, then the default bench line
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/all_pytest.log 2>&1 || { grep -v "^  File" gpurun_out/all_pytest.log | tail -n 60 | cut -c1-400; exit 1; }
tail -n 3 gpurun_out/all_pytest.log
timeout -k 10 600 python bench.py > gpurun_out/bench_r03a.json 2> gpurun_out/bench_r03a.err || { tail -n 20 gpurun_out/bench_r03a.err; exit 1; }
python tools/show_bench.py gpurun_out/bench_r03a.json
