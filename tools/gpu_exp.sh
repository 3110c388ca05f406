# Timing experiments on the down-date (WRONG results by design: parts of the kernel switched off).  They live only in
# the experiments build of the library (make -C slam.jl_amd/csrc exp), selected here with SLAMHIP_LIBRARY; the product
# library neither contains these instantiations nor reads SLAMHIP_DEBUG.
mkdir -p gpurun_out
make -C slam.jl_amd/csrc exp > gpurun_out/exp_build.log 2>&1 || { tail gpurun_out/exp_build.log; exit 1; }
export SLAMHIP_LIBRARY=slam.jl_amd/libslamhip_exp.so
for d in ${SLAMHIP_EXP_LIST:-32 33 34 35}; do
  SLAMHIP_DEBUG=$d timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-fastslam 2>>gpurun_out/exp.log | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('dbg=$d syrk_ms', round(j['roofline']['avg_launch_ms'],4), 'step_ms', round(j['ms_per_step'],4))
" >> gpurun_out/exp.log
done
cat gpurun_out/exp.log
