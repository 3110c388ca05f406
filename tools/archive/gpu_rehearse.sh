# Two ranks on ONE card over gloo: checks the multi-rank plumbing of bench.py (torchrun env, sharded FastSLAM with
# record exchange, max-over-ranks timing).  Numbers from this run mean nothing.
mkdir -p gpurun_out
SLAM_BENCH_REHEARSE=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node ${RANKS:-2} --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus ${RANKS:-2} --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/rehearse.log 2>&1
echo "rehearse exit $?" >> gpurun_out/rehearse.log
tail -c 1500 gpurun_out/rehearse.log
