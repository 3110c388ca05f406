# round 3: PF tests after the tagged rank exchange; then the auto step at the shard sizes of a 2 / 4 / 8-rank filter on ONE GPU
# (the per-step time a rank would have WITHOUT any exchange: the floor of strong scaling)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_pf.py -m gpu -q -x --timeout 600 > gpurun_out/pf_pytest.log 2>&1 || { grep -v "^  File" gpurun_out/pf_pytest.log | tail -n 60 | cut -c1-300; exit 1; }
tail -n 2 gpurun_out/pf_pytest.log
for np in 262144 131072 65536 32768; do PF_PROBE_NP=$np timeout -k 10 200 python tools/pf_auto_probe.py 2>/dev/null | sed "s/^/np=$np /"; done > gpurun_out/pf_shard_sizes.log
cat gpurun_out/pf_shard_sizes.log | cut -c1-260
