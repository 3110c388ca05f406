# round 5: the bench with the 48-launch roofline mean, the batched FastSLAM regime, and the two-rank rehearsal with the parity check
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_pf.py -x -q -m gpu -k "two_ranks_on_one_card or grid_sizes" > gpurun_out/r5y_tests.log 2>&1
echo "tests exit $?"; tail -8 gpurun_out/r5y_tests.log
timeout -k 10 700 python bench.py --steps 20 --warmup 5 > gpurun_out/r5y_bench.log 2> gpurun_out/r5y_bench.err
echo "bench exit $?"
python tools/show_bench.py gpurun_out/r5y_bench.log 2>/dev/null | head -60 || tail -c 3000 gpurun_out/r5y_bench.log
