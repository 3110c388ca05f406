# round 5: the whole GPU suite on the cleaned product library (ekf_syrk.hip without laboratory hooks), then the driver's bench command
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r5aa_tests.log 2>&1
echo "tests exit $?"; tail -6 gpurun_out/r5aa_tests.log
timeout -k 10 700 python bench.py --steps 20 --warmup 5 > gpurun_out/r5aa_bench.log 2> gpurun_out/r5aa_bench.err
echo "bench exit $?"
python tools/show_bench.py gpurun_out/r5aa_bench.log | head -24
