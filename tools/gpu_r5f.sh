# round 4: soak of the LDS-DMA pipeline against the register-staged one (bit for bit, 12 repetitions x 4 sizes)
mkdir -p gpurun_out
timeout -k 10 1000 python tools/soak_dma.py 12 > gpurun_out/r5f_soak.txt 2>&1
echo "exit $?"; tail -n 4 gpurun_out/r5f_soak.txt
