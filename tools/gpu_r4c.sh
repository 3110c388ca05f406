# round 4, third GPU call: the W-way step kernel (pf_auto_step_way_kernel) -- its tests and the invariance tests, the hang probe
# (fixed: the IPC handle is passed by value), then DESIGN section 7's shard table re-measured: the auto step at the shard sizes
# of a 1 / 2 / 4 / 8-rank filter on ONE GPU, product library (default thresholds), and the experiments build with the ways
# switched off / forced to 2 / 4 as the A/B
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_pf.py -m gpu -q -x --timeout 800 -k "ways or invariant or observation_parallel or two_ranks_on_one_card" > gpurun_out/r4c_pytest.log 2>&1 || { grep -v "^  File" gpurun_out/r4c_pytest.log | tail -n 80 | cut -c1-500; exit 1; }
tail -n 3 gpurun_out/r4c_pytest.log
timeout -k 10 120 python tools/ipc_open_stack.py 2.5 12 > gpurun_out/r4c_ipc_open_stack.log 2>&1; echo "ipc_open_stack exit $?"; tail -n 70 gpurun_out/r4c_ipc_open_stack.log | cut -c1-260
for np in 262144 196608 131072 98304 65536 49152 32768 16384; do PF_PROBE_NP=$np timeout -k 10 200 python tools/pf_auto_probe.py 2>/dev/null | sed "s/^/np=$np /"; done > gpurun_out/r4c_shard_sizes.log
cut -c1-200 gpurun_out/r4c_shard_sizes.log
export SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so
for np in 196608 131072 98304 65536; do for cfg in "0 0" "0 1000000" "1000000 1000000"; do
  set -- $cfg
  SLAMHIP_PF_WAY4_MAX=$1 SLAMHIP_PF_WAY2_MAX=$2 PF_PROBE_NP=$np timeout -k 10 200 python tools/pf_auto_probe.py 2>/dev/null | sed "s/^/np=$np way4_max=$1 way2_max=$2 /"
done; done > gpurun_out/r4c_ways_ab.log
cut -c1-160 gpurun_out/r4c_ways_ab.log
