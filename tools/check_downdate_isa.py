#!/usr/bin/env python3
"""Build-time check of the down-date's LDS-DMA pipeline (csrc/ekf_syrk.hip: dd_stream_dma): its hand-counted
s_waitcnt vmcnt(N) are right only if, between two barriers, a step's three chunk requests (buffer_load ... lds) stand IN FRONT
of the P tile's 32 loads / 32 stores in the wave's queue.  Compiles the file to assembly (hipcc -S, the Makefile's flags) and
looks at every stretch between two s_barrier of the product kernel downdate_f32_mfma<4, 3, true>:
  * a stretch with one chunk request (a step of the pipeline) STARTS with its three pieces: no P load / store (a buffer operation
    without `lds`) in front of them (where the kernel's inlined variants meet, the text behind them may run into another path);
  * both orders the counts rely on exist: [3 chunk pieces][32 P stores] and [3 chunk pieces][32 P loads].
usage: check_downdate_isa.py [file.s]   (without an argument: compiles slam.jl_amd/csrc/ekf_syrk.hip into a temporary file)
Exit code 0 and 'ok ...' when the order holds."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "downdate_f32_mfmaILi4ELi3ELb1EE"


def listing():
    if len(sys.argv) > 1:
        return open(sys.argv[1]).read()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "ekf_syrk.s")
        subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                        "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-S", "--cuda-device-only",
                        os.path.join(ROOT, "slam.jl_amd", "csrc", "ekf_syrk.hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
        return open(out).read()


def check(text):
    lines = text.split("\n")
    start = next(i for i, l in enumerate(lines) if KERNEL in l and l.startswith("_Z") and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    stretches, cur = [], []
    for l in lines[start:end]:
        s = l.strip()
        if s.startswith("s_barrier"):
            stretches.append(cur)
            cur = []
        elif re.match(r"buffer_(load|store)_", s):
            cur.append("dma" if re.search(r"\blds\b", s) else ("st" if s.startswith("buffer_store") else "ld"))
    stretches.append(cur)
    # a STEP of the pipeline = a stretch between two barriers with exactly one chunk request (three pieces); stretches with more
    # are the prologues (three chunks up front), where textual neighbours belong to different paths
    steps = [s for s in stretches if s.count("dma") == 3]
    problems = []
    n_st = n_ld = 0
    for k, ops in enumerate(steps):
        if ops[:3] != ["dma", "dma", "dma"]:
            problems.append(f"step {k}: a P operation stands in front of the chunk request: {ops}")
        rest = ops[3:]
        if rest == ["st"] * 32:
            n_st += 1
        elif rest == ["ld"] * 32:
            n_ld += 1
    if n_st == 0:
        problems.append("no [3 chunk pieces][32 P stores] step found")
    if n_ld == 0:
        problems.append("no [3 chunk pieces][32 P loads] step found")
    return len(steps), n_st, n_ld, problems


if __name__ == "__main__":
    n, n_st, n_ld, problems = check(listing())
    if problems:
        print("\n".join(problems))
        sys.exit(1)
    print(f"ok: {n} pipeline steps between barriers; [3 chunk pieces][32 P stores] x {n_st}, [3 chunk pieces][32 P loads] x {n_ld}; no P operation in front of a step's chunk request")
