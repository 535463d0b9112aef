"""Shared test helpers: lane-level emulator binding and comparison utilities."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (lanes per beam, elements per lane) of every compiled tiling; (6, 17) is the fat-wave tiling of beam_fat.hip
# (shared geometry + one constraint mask only)
TILINGS = [(6, 17), (8, 13), (16, 7), (32, 4), (64, 2), (64, 4), (64, 8), (64, 16)]
FAT_P = (6,)
_emul = None


def emul_lib():
    """g++ build of tests/csrc/emul_beam.cpp (the kernel's per-lane arithmetic on the CPU; test code only)."""
    global _emul
    if _emul is None:
        src = os.path.join(ROOT, "tests", "csrc", "emul_beam.cpp")
        hdr = os.path.join(ROOT, "openpystruct_amd", "csrc", "beam_math.hpp")
        so = os.path.join(ROOT, "tests", "csrc", "libemul_beam.so")
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so, src])
        _emul = ctypes.CDLL(so)
        f = _emul.emul_beam_solve_batched_f64
        f.restype = ctypes.c_int
        vp, lg = ctypes.c_void_p, ctypes.c_long
        f.argtypes = [ctypes.c_int] * 4 + [vp, lg] * 6 + [vp] * 5
    return _emul


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def emul_solve(P, M, x, E, I, fix, Fy, wy):
    I = np.ascontiguousarray(I, dtype=np.float64)
    B, Ne = I.shape
    N = Ne + 1
    x = np.ascontiguousarray(x, dtype=np.float64)
    fix = np.ascontiguousarray(fix, dtype=np.uint8)
    Fy = np.ascontiguousarray(Fy, dtype=np.float64)
    E = np.ascontiguousarray(np.atleast_1d(np.asarray(E, dtype=np.float64)))
    wy = np.ascontiguousarray(np.atleast_1d(np.asarray(wy, dtype=np.float64)))
    v = np.empty((B, N)); th = np.empty((B, N)); V = np.empty((B, Ne)); M_ = np.empty((B, Ne))
    st = np.empty(B, dtype=np.int32)
    rc = emul_lib().emul_beam_solve_batched_f64(
        P, M, B, Ne, _p(x), N if x.ndim == 2 else 0, _p(E), Ne if E.ndim == 2 else 0, _p(I), Ne,
        _p(fix), N if fix.ndim == 2 else 0, _p(Fy), N, _p(wy), Ne if wy.ndim == 2 else 0,
        _p(v), _p(th), _p(V), _p(M_), _p(st))
    assert rc == 0, rc
    return v, th, V, M_, st


def relerr(a, b):
    """max |a-b| / max |b| per beam (row), worst row."""
    a = np.asarray(a); b = np.asarray(b)
    scale = np.abs(b).max(axis=-1, keepdims=True)
    scale = np.where(scale > 0, scale, 1.0)
    return float((np.abs(a - b) / scale).max())


def load_golden(path):
    z = np.load(path)
    return {k: z[k] for k in z.files}


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def run_ranks(target, world, timeout):
    """Spawn `world` daemon processes running target(rank, world, port, queue); return their queue items.  Children are
    terminated whatever happens, so a failed rank can never leave the test run hanging at interpreter exit."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    ps = [ctx.Process(target=target, args=(r, world, port, q), daemon=True) for r in range(world)]
    [p.start() for p in ps]
    try:
        return [q.get(timeout=timeout) for _ in ps]
    finally:
        for p in ps:
            p.join(10)
            if p.is_alive():
                p.terminate()


def kappa_scaled(x, E, I, fix):
    """cond(D^-1/2 K_ff D^-1/2), D = diag(K_ff): the Jacobi-scaled condition number of one beam's free-DOF stiffness
    matrix -- what governs the rounding error of a Cholesky-type elimination in any order (van der Sluis)."""
    from oracle import beam_oracle as bo
    x = np.asarray(x, dtype=np.float64)
    N = x.shape[0]
    K, _ = bo.assemble_beam(x, E, I, np.zeros(N), 0.0)
    fix = np.asarray(fix).astype(np.int64)
    free = np.ones(2 * N, dtype=bool); free[0::2] = (fix & 1) == 0; free[1::2] = (fix & 2) == 0
    Kf = K[np.ix_(free, free)]
    d = 1.0 / np.sqrt(np.diag(Kf))
    return float(np.linalg.cond(Kf * d[:, None] * d[None, :]))


def marginal_stop_decisions(loss_history, tolerance, rel=2e-6):
    """Replays the reference's early-stop bookkeeping (SingleCore.py:211-219) on a loss history and returns the epochs (1-based) whose
    `loss < best_loss - tolerance` decision sat within `rel` * |loss| of the threshold -- float32 sums ordered differently (GPU vs CPU)
    can take such a decision the other way, which moves the stopping epoch by up to `patience`."""
    best, out = float("inf"), []
    for e, l in enumerate(loss_history):
        margin = (best - tolerance) - float(l)
        if np.isfinite(margin) and abs(margin) <= rel * abs(float(l)):
            out.append(e + 1)
        if float(l) < best - tolerance:
            best = float(l)
    return out


def assert_stop_epochs_agree(ep, ref_ep, ref_loss_history, tolerance, patience, what=""):
    """Same stopping epoch -- or, when the reference's history holds a threshold decision within float32 round-off, one that differs by
    at most `patience` epochs (one improvement counted / not counted resets or does not reset the patience counter once)."""
    if int(ep) == int(ref_ep):
        return True
    marg = marginal_stop_decisions(ref_loss_history, tolerance)
    assert marg and abs(int(ep) - int(ref_ep)) <= patience, (what, int(ep), int(ref_ep), marg[-5:])
    return False
