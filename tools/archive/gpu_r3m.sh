# round 3: the observation-parallel step kernel: PF tests, then the step time at the shard sizes with it (default threshold) and
# without it (experiments build, SLAMHIP_PF_PAR_MAX=0)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_pf.py -m gpu -q -x --timeout 600 > gpurun_out/pf_pytest.log 2>&1 || { grep -v "^  File" gpurun_out/pf_pytest.log | tail -n 60 | cut -c1-300; exit 1; }
tail -n 2 gpurun_out/pf_pytest.log
export SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so
for np in 16384 32768 65536 98304 131072 262144; do for mx in 0 1000000; do
  SLAMHIP_PF_PAR_MAX=$mx PF_PROBE_NP=$np timeout -k 10 200 python tools/pf_auto_probe.py 2>/dev/null | sed "s/^/np=$np par_max=$mx /"
done; done > gpurun_out/pf_par.log
cut -c1-150 gpurun_out/pf_par.log
