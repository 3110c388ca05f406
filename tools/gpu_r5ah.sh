# round 5: the full GPU suite on the final code, then the round's profile set (tools/gpu_profile_round.sh ${1:-r05a})
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r5ah_tests.log 2>&1
rc=$?; echo "tests exit $rc"; tail -4 gpurun_out/r5ah_tests.log
[ $rc -eq 0 ] || exit $rc
bash tools/gpu_profile_round.sh ${1:-r05a}
