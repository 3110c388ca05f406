# round 3: the whole GPU suite, then the round's profile set (rocprofv3 stats, bench lines for C3 / C2 / C5, PMC passes)
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/all_pytest.log 2>&1 || { grep -v "^  File" gpurun_out/all_pytest.log | tail -n 60 | cut -c1-400; exit 1; }
tail -n 3 gpurun_out/all_pytest.log
bash tools/gpu_profile_round.sh r03c
