# round 4: the driver's own command on the final code, timed
mkdir -p gpurun_out
s=$(date +%s.%N)
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5b_bench.json 2> gpurun_out/r5b_bench.err
rc=$?
e=$(date +%s.%N)
echo "rc $rc driver_run_s $(python3 -c "print(round($e-$s,1))")"
python tools/show_bench.py gpurun_out/r5b_bench.json | head -30
