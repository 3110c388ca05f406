# round 4: the bit-for-bit test of the two chunk pipelines, now also at a size where a workgroup walks eight or nine tiles
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_ekf.py -m gpu -q -x --timeout 600 -k "lds_dma" 2>&1 | tail -n 5
