"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
that include/slamhip.h (the boundary) and include/slamhip_diag.h (measurement and introspection)
declare, carries no torch / oracle dependency, and fails loudly (never falls back to a CPU path)
when no device is present."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "slamhip.h")
DIAG_HEADER = os.path.join(ROOT, "include", "slamhip_diag.h")


def declared_symbols(headers=(HEADER, DIAG_HEADER)):
    syms = set()
    for path in headers:
        text = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
        syms |= set(re.findall(r"\b(slam_[a-z0-9_]+)\s*\(", text))
    return sorted(syms)


def test_the_boundary_header_stays_small():
    """diagnostics live in slamhip_diag.h; the boundary itself is at most 60 entry points."""
    assert len(declared_symbols((HEADER,))) <= 60
    assert not set(declared_symbols((HEADER,))) & set(declared_symbols((DIAG_HEADER,)))


def test_header_declares_the_hot_path():
    syms = declared_symbols()
    for need in ("slam_ekf_create", "slam_ekf_destroy", "slam_ekf_predict", "slam_ekf_associate",
                 "slam_ekf_update", "slam_ekf_augment", "slam_ekf_nis", "slam_ekf_predict_observation",
                 "slam_ekf_set_state", "slam_ekf_get_state", "slam_last_error"):
        assert need in syms


def test_library_exports_every_declared_symbol(pkg):
    lib = ctypes.CDLL(pkg._lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    # and the Python binding covers exactly the declared surface
    assert sorted(pkg._lib.SIGNATURES) == declared_symbols()


def test_product_library_is_not_the_laboratory(pkg):
    """The product library exports exactly what the two headers declare (no slam_exp_* entry, no debug trace) and contains none of
    the experiments build's environment switches: those live in libslamhip_exp.so (make exp), which no test and no bench loads."""
    out = subprocess.run(["nm", "-D", "--defined-only", pkg._lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if " T " in line and line.split()[-1].startswith("slam_"))
    assert exported == declared_symbols(), sorted(set(exported) ^ set(declared_symbols()))
    blob = open(pkg._lib.LIB_PATH, "rb").read()
    for name in (b"SLAMHIP_DEBUG", b"SLAMHIP_FW1", b"SLAMHIP_STAMPS", b"SLAMHIP_WGS", b"SLAMHIP_PB_W", b"SLAMHIP_W1DBG", b"SLAMHIP_SP"):
        assert name not in blob, name
    for name in (b"SLAMHIP_X", b"SLAMHIP_ROCTX"):                  # (the product's own switches are there)
        assert name in blob, name


def test_library_has_no_torch_or_python_dependency(pkg):
    out = subprocess.run(["ldd", pkg._lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "libamdhip64" in out
    assert "torch" not in out and "python" not in out


def test_product_package_never_imports_the_oracle():
    pkgdir = os.path.join(ROOT, "slam.jl_amd")
    for dirpath, _dirs, files in os.walk(pkgdir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".jl")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "ekf_ref" not in src, f


def build_c_client(tmpdir):
    """tests/abi_client.c compiled AS C (c99, warnings are errors) against include/slamhip.h and linked with the library
    alone: what a cgo / ccall / JNI binding does.  Returns the executable's path."""
    exe = os.path.join(str(tmpdir), "abi_client")
    libdir = os.path.join(ROOT, "slam.jl_amd")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "abi_client.c"), "-o", exe, "-L", libdir, "-lslamhip", "-lm",
           "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return exe


def test_plain_c_client_compiles_links_and_fails_loudly_without_a_device(pkg, tmp_path):
    exe = build_c_client(tmp_path)
    if pkg.device_count() > 0:
        pytest.skip("a HIP device is present (the GPU suite runs the client)")
    res = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert res.returncode == 2 and "no HIP device" in res.stderr      # no silent CPU path behind the C ABI either


def test_no_device_means_loud_failure_not_fallback(pkg):
    if pkg.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(pkg.SlamHipError) as ei:
        pkg.EKFSlamState(np.zeros(3), np.zeros((3, 3)))
    assert ei.value.code == pkg._lib.SLAM_E_HIP
    assert "no CPU fallback" in str(ei.value)


def test_bad_arguments_are_status_codes_not_crashes(pkg):
    lib = pkg._lib.lib
    assert lib.slam_ekf_create(None, 0, 10, 0) == pkg._lib.SLAM_E_BADARG
    h = ctypes.c_void_p()
    assert lib.slam_ekf_create(ctypes.byref(h), 7, 10, 0) == pkg._lib.SLAM_E_BADARG
    assert "dtype" in pkg._lib.last_error()
    assert lib.slam_ekf_destroy(None) == 0
    assert lib.slam_ekf_predict(None, 1.0, 0.0, 4.0, None, 0.025) == pkg._lib.SLAM_E_BADARG


def test_host_helpers(pkg):
    import math
    assert pkg.mpi_to_pi(3.5 * math.pi) == pytest.approx(1.5 * math.pi)
    assert pkg.mpi_to_pi(-3.5 * math.pi) == pytest.approx(-1.5 * math.pi)
    assert pkg.mpi_to_pi(0.25) == 0.25
    from slam_jl_amd import ekf
    assert ekf._small(np.array([[1.0, 2.0], [3.0, 4.0]])).tolist() == [1.0, 3.0, 2.0, 4.0]     # column-major
    z = np.array([[10.0, 20.0, 30.0], [0.1, 0.2, 0.3]])
    assert ekf._obs(z).tolist() == [[10.0, 0.1], [20.0, 0.2], [30.0, 0.3]]
    assert ekf._obs(np.zeros((2, 0))).shape == (0, 2)


def test_batch_packing_for_the_k_step_call(pkg):
    """Host logic of FastSLAM.step_async_batch / slam_pf_step_auto_batch (no GPU): K steps packed once in the layout the entry point
    takes -- controls K x (V, G) contiguous, observations as (range, bearing) pairs padded to the widest step, ids padded with
    zeros, the force words -1 (Neff rule) / 0 / 1."""
    z1 = np.array([[10.0, 20.0, 30.0], [0.1, 0.2, 0.3]])
    z2 = np.array([[5.0], [-0.5]])
    K, vg, zz, ii, ms, stride, ff = pkg.PFShard.prepare_batch([(1.0, 0.1), (2.0, -0.2), (3.0, 0.0)], [(z1, [4, 2, 9]), (z2, [7]), (np.zeros((2, 0)), [])],
                                                              [None, False, True])
    assert K == 3 and stride == 3 and ms.tolist() == [3, 1, 0] and ms.dtype == np.int32
    assert vg.flags["C_CONTIGUOUS"] and vg.tolist() == [[1.0, 0.1], [2.0, -0.2], [3.0, 0.0]]
    assert zz.shape == (3, 3, 2) and zz[0].tolist() == [[10.0, 0.1], [20.0, 0.2], [30.0, 0.3]] and zz[1, 0].tolist() == [5.0, -0.5]
    assert not zz[1, 1:].any() and not zz[2].any()
    assert ii.dtype == np.int32 and ii.tolist() == [[4, 2, 9], [7, 0, 0], [0, 0, 0]]
    assert ff.tolist() == [-1, 0, 1] and ff.dtype == np.int32
    assert pkg.PFShard.prepare_batch([(1.0, 0.0)] * 2, [(z2, [1])] * 2)[6].tolist() == [-1, -1]          # the Neff rule at every step
    assert pkg.PFShard.prepare_batch([(1.0, 0.0)] * 2, [(z2, [1])] * 2, False)[6].tolist() == [0, 0]     # one word for all
    with pytest.raises(ValueError):
        pkg.PFShard.prepare_batch([(1.0, 0.0)] * 2, [(z2, [1])] * 2, [True])


def test_sim_driver_pieces(pkg, golden_dir):
    import math
    S = pkg.sim
    wp = S.get_waypoints(os.path.join(golden_dir, "course1.txt"))
    assert wp.shape == (2, 19) and wp[0, 0] == pytest.approx(49.954)
    pose = S.initial_pose(wp)
    assert pose[2] == pytest.approx(math.atan2(14.904 - 14.665, 41.931 - 49.954))
    lm = S.make_landmarks(500, [0.0, 100.0, 0.0, 100.0], 0.05, np.random.default_rng(0))
    assert lm.min() == 5 and lm.max() == 95 and np.all(lm == np.round(lm))
    veh = S.Vehicle(pose=pose.copy())
    S.steer(veh, wp, S.D_MIN, S.DT)
    assert veh.waypoint_id == 2 and abs(veh.target_gamma) <= veh.steer_rate * S.DT + 1e-15
    S.step_vehicle(veh, S.DT)
    assert np.hypot(*(veh.pose[:2] - pose[:2])) == pytest.approx(8 * 0.025)
    # sensor model: forward half-plane and 30 m range (sim/sim-utils.jl:20-22)
    veh.pose = np.array([50.0, 50.0, 0.0])
    lms = np.array([[60.0, 40.0, 50.0, 79.9, 80.1], [50.0, 50.0, 70.0, 50.0, 50.0]])
    z, tags = S.get_observations(veh, lms, np.zeros((2, 2)), np.random.default_rng(1))
    assert tags.tolist() == [1, 4] and z[0].tolist() == pytest.approx([10.0, 29.9])


def test_telemetry_host_helpers():
    """Row N3, host side (no GPU): laser_lines / local_to_global / dict_array restate src/common.jl:118-132,269-283
    and sim/browser/wsserver.jl:120-131; checked against hand-computed values."""
    import importlib.util
    import json
    import math
    import os
    spec = importlib.util.spec_from_file_location(
        "slam_telemetry_only", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "slam.jl_amd", "telemetry.py"))
    T = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(T)
    import numpy as np
    pose = [1.0, 2.0, math.pi / 2]
    lines = T.laser_lines(np.array([[10.0, 5.0], [0.0, math.pi / 2]]), pose)
    assert np.allclose(lines, [[1.0, 1.0], [2.0, 2.0], [1.0, -4.0], [12.0, 2.0]], atol=1e-12)
    d = T.dict_array(np.array([[10, 5, 8], [5, 3, 6]]), ["a", "b"])
    assert d == [{"a": 10.0, "b": 5.0}, {"a": 5.0, "b": 3.0}, {"a": 8.0, "b": 6.0}]      # the docstring example of the reference
    m = T.message("tracks", {"x": 1}, timestamp=3.0)
    assert json.loads(T.to_json(m)) == {"type": "tracks", "data": {"x": 1}, "timestamp": 3.0}


def _julia_ccalls():
    """(symbol, number of argument types) of every ccall in SLAMHip.jl."""
    src = open(os.path.join(ROOT, "slam.jl_amd", "SLAMHip.jl")).read()
    out = []
    for m in re.finditer(r"ccall\(\(:(slam_[a-z0-9_]+),\s*libslamhip\),\s*([A-Za-z0-9]+),\s*\(", src):
        i = m.end()                                  # just past the "(" of the argument-type tuple
        depth, j = 1, i
        while depth:
            depth += {"(": 1, ")": -1}.get(src[j], 0)
            j += 1
        body = src[i:j - 1]
        d, parts, cur = 0, [], ""
        for ch in body:                              # split at top-level commas (Ref{Ptr{Cvoid}} has none, tuples might)
            if ch in "({":
                d += 1
            elif ch in ")}":
                d -= 1
            if ch == "," and d == 0:
                parts.append(cur)
                cur = ""
            else:
                cur += ch
        if cur.strip():
            parts.append(cur)
        out.append((m.group(1), m.group(2), len([x for x in parts if x.strip()])))
    return out


def test_julia_binding_names_exported_symbols_with_the_right_arity(pkg):
    """SLAMHip.jl cannot be executed here (no julia): at least every `ccall((:name, libslamhip), ...)` in it must name a
    symbol the library exports, with as many argument types as the ctypes binding (which the GPU tests exercise) and
    Cint / Cstring as the return type; and the PF surface SURVEY 8b lists must be bound."""
    calls = _julia_ccalls()
    assert len(calls) >= 28
    sigs = pkg._lib.SIGNATURES
    lib = ctypes.CDLL(pkg._lib.LIB_PATH)
    for name, ret, nargs in calls:
        assert hasattr(lib, name), name
        assert name in sigs, name
        assert nargs == len(sigs[name][1]), (name, nargs, len(sigs[name][1]))
        assert ret == ("Cstring" if name == "slam_last_error" else "Cint"), (name, ret)
    bound = {c[0] for c in calls}
    for need in ("slam_pf_create", "slam_pf_destroy", "slam_pf_predict", "slam_pf_update_known", "slam_pf_step_auto", "slam_pf_flush",
                 "slam_pf_resample", "slam_pf_get_mean_pose", "slam_pf_get_weights", "slam_ekf_ellipses", "slam_ekf_get_block",
                 "slam_ekf_observe"):
        assert need in bound, need


def test_the_downdates_lds_dma_pipeline_keeps_its_queue_order():
    """dd_stream_dma's hand-counted s_waitcnt vmcnt(N) hold only while a step's chunk request stands in front of the P
    tile's loads / stores in the wave's memory queue.  tools/check_downdate_isa.py compiles csrc/ekf_syrk.hip to assembly (the
    Makefile's flags) and looks at the order in the product kernel; a toolchain or source change that swaps them fails here, not
    silently on the GPU."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_downdate_isa.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("ok:")
