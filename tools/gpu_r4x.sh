# round 4, closing GPU call: the whole GPU suite on the final code, then the round's profile set (r04c)
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/r4x_pytest.log 2>&1 || { grep -v "^  File" gpurun_out/r4x_pytest.log | tail -n 80 | cut -c1-500; exit 1; }
tail -n 3 gpurun_out/r4x_pytest.log
bash tools/gpu_profile_round.sh r04c 2>&1 | cut -c1-200 | grep -v "^sq\|^tcc\|^grbm" | tail -n 40
