# round 5: the conditional resampling kernels trimmed (block offsets once, by the cdf kernel's last workgroup; three rounds of probes
# instead of ten halvings; gathers of sixteen): the particle tests, then the FastSLAM leg of the bench
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_pf.py tests/test_gpu_pf_batch.py -x -q -m gpu > gpurun_out/r5ab_tests.log 2>&1
echo "tests exit $?"; tail -5 gpurun_out/r5ab_tests.log
timeout -k 10 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-configs > gpurun_out/r5ab_bench.log 2> gpurun_out/r5ab_bench.err
echo "bench exit $?"
python tools/show_bench.py gpurun_out/r5ab_bench.log | tail -8
PF_PROBE_NPS=262144,131072,65536,32768 PF_PROBE_STEPS=3840 timeout -k 10 200 python tools/pf_batch_probe.py 2>&1 | grep "^n "
