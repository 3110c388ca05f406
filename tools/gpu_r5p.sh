# round 5: first run of the persistent K-step launch: its tests, then the probe
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests/test_gpu_pf_batch.py -x -q -m gpu > gpurun_out/r5p_tests.log 2>&1
echo "tests exit $?"
tail -15 gpurun_out/r5p_tests.log
PF_PROBE_STAMPS=1 timeout -k 10 300 python tools/pf_batch_probe.py > gpurun_out/r5p_probe.log 2>&1
echo "probe exit $?"
cat gpurun_out/r5p_probe.log
