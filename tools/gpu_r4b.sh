# round 4, second GPU call: the particle path after the chunked landmark records, the canonical statistics tree and the peer
# fences -- the whole FastSLAM GPU suite (incl. the new invariance test, the IPC generations + the weak-scaling shape attached),
# then ONE run of the hang probe (where does hipIpcOpenMemHandle of 2.5 GiB sit?), then the short bench
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_pf.py -m gpu -q -x --timeout 900 > gpurun_out/r4b_pf_pytest.log 2>&1 || { grep -v "^  File" gpurun_out/r4b_pf_pytest.log | tail -n 80 | cut -c1-500; exit 1; }
tail -n 3 gpurun_out/r4b_pf_pytest.log
timeout -k 10 120 python tools/ipc_open_stack.py 2.5 12 > gpurun_out/r4b_ipc_open_stack.log 2>&1; echo "ipc_open_stack exit $?"; tail -n 60 gpurun_out/r4b_ipc_open_stack.log | cut -c1-300
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-configs > gpurun_out/r4b_bench.json 2> gpurun_out/r4b_bench.err || { tail -n 30 gpurun_out/r4b_bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4b_bench.json").read().strip().splitlines()[-1])
f = d.get("fastslam", {})
print("fastslam", f.get("error") or {k: round(v["ms_per_step"] * 1e3, 1) for k, v in f["regimes"].items()})
PY
