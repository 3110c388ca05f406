# round 4, seventh GPU call: the whole GPU suite on the round's code, then the round's profile set (tools/gpu_profile_round.sh:
# rocprofv3 --kernel-trace --stats of the default bench, the plain default bench with configs + in-run PMC traffic, the PMC passes)
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/r4g_pytest.log 2>&1 || { grep -v "^  File" gpurun_out/r4g_pytest.log | tail -n 80 | cut -c1-500; exit 1; }
tail -n 3 gpurun_out/r4g_pytest.log
bash tools/gpu_profile_round.sh r04g 2>&1 | cut -c1-230
