# C5 (50k landmarks, fp64, Joseph form, 8 observations): parity tests of the EKF file, then the bench line
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ekf.py -m gpu -q -x --timeout 400 > gpurun_out/c5_pytest.log 2>&1 || { tail -n 20 gpurun_out/c5_pytest.log; exit 1; }
tail -n 2 gpurun_out/c5_pytest.log
timeout -k 10 400 python bench.py --landmarks 50000 --obs 8 --dtype f64 --form joseph --no-cpu-baseline --no-fastslam --steps 20 --warmup 3 > gpurun_out/c5_bench.log 2>&1
echo "bench exit $?"; tail -n 3 gpurun_out/c5_bench.log
