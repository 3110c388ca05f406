import os, sys, math
import numpy as np
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
pkg = load_package()
sys.path.insert(0, "tests")
import importlib
T = importlib.import_module("tests.test_gpu_ekf")
R = T.R
m, N = 33, 1500
rng = np.random.default_rng(900 + m)
x, P = T.random_state(rng, N, spread=600.0)
res = {}
FA, FB = os.environ.get("DBG_A", "0"), os.environ.get("DBG_B", "128")
print("A =", FA, "B =", FB)
for name, flag in (("fused", FA), ("two", FB)):
    if flag is None: os.environ.pop("SLAMHIP_X", None)
    else: os.environ["SLAMHIP_X"] = flag
    st = pkg.EKFSlamState(x, P, dtype="f32", max_landmarks=N)
    r2 = np.random.default_rng(13)
    outs = []
    for step in range(3):
        xo = st.download("x").astype(np.float64)
        ids = r2.permutation(N)[:m] + 1
        st.update(T.noisy_obs(r2, xo, ids), R, ids)
        ws = []
        if os.environ.get("SLAMHIP_LIBRARY"):
            import ctypes as C
            lib = pkg._lib.lib
            lib.slam_exp_workspace.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
            for what in range(4):
                sz = C.c_size_t()
                assert lib.slam_exp_workspace(st._h, what, None, 0, C.byref(sz)) == 0
                buf = np.zeros(sz.value, dtype=np.uint8)
                assert lib.slam_exp_workspace(st._h, what, buf.ctypes.data, sz.value, None) == 0
                ws.append(buf)
        outs.append(st.download() + tuple(ws))
    res[name] = outs
    st.close()
for step in range(3):
    xa, Pa = res["fused"][step][:2]; xb, Pb = res["two"][step][:2]
    if len(res["fused"][step]) > 2:
        for what, nm in enumerate(("W1", "Wimg", "C", "g")):
            a, b = res["fused"][step][2 + what], res["two"][step][2 + what]
            dd = np.flatnonzero(a != b)
            print("  ", nm, "bytes differing", len(dd), "first at", dd[:8])
        kcap = int(round(math.sqrt(len(res["fused"][step][4]) / 8)))
        ia, ib = res["fused"][step][3], res["two"][step][3]
        for off in np.flatnonzero(ia != ib)[:6]:
            off2 = int(off) & ~1
            blk, rem = divmod(off2, 4096)              # [row block * nch + chunk][split] blocks of 4096 bytes
            rc, split = divmod(blk, 3)
            rowblk, chunk = divmod(rc, kcap // 16)
            rr, inrow = divmod(rem, 32)
            half, pos = divmod(inrow, 16)
            i8 = pos // 2
            swz = (rr >> 3) & 1
            i = ((half ^ swz) << 3) | i8
            row, col = rowblk * 128 + rr, chunk * 16 + i
            def parts(img):
                base = ((rowblk * (kcap // 16) + chunk) * 3) * 4096 + rr * 32 + half * 16 + i8 * 2
                out = []
                for sp in range(3):
                    u = int(img[base + sp * 4096]) | (int(img[base + sp * 4096 + 1]) << 8)
                    out.append(np.array([u << 16], dtype=np.uint32).view(np.float32)[0])
                return out
            W1a = res["fused"][step][2].view(np.float32).reshape(-1, 2 * kcap)
            pa, pb = parts(ia), parts(ib)
            print(f"   image byte {int(off)}: split {split} row {row} col {col}  W1 {W1a[row, col]!r}  fused h,m,l {pa} sum {np.float64(pa[0]) + np.float64(pa[1]) + np.float64(pa[2])!r}  two {pb} sum {np.float64(pb[0]) + np.float64(pb[1]) + np.float64(pb[2])!r}")
        W1a = res["fused"][step][2].view(np.float32).reshape(-1, 2 * kcap); W1b = res["two"][step][2].view(np.float32).reshape(-1, 2 * kcap)
        dr = np.argwhere(W1a != W1b)
        print("   kcap", kcap, "W1 entries differing", len(dr), dr[:10].tolist())
    d = np.argwhere(Pa != Pb)
    print("step", step, "x equal", np.array_equal(xa, xb), "P mismatches", len(d))
    if len(d):
        rows = np.unique(d[:, 0]); cols = np.unique(d[:, 1])
        print("  rows", rows[:20], "... n rows", len(rows), " cols", cols[:20], "n cols", len(cols))
        print("  tile rows", np.unique(rows >> 7), "tile cols", np.unique(cols >> 7))
        i, j = d[0]; print("  first", i, j, Pa[i, j], Pb[i, j], "max abs diff", np.abs(Pa.astype(np.float64) - Pb).max())
