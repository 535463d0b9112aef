"""ctypes binding of oracle/libbeam_oracle.so (plain-C oracle).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libbeam_oracle.so")
    src = os.path.join(_HERE, "beam_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        f = _LIB.oracle_beam_solve_batched_f64
        f.restype = ctypes.c_int
        P, L = ctypes.c_void_p, ctypes.c_long
        f.argtypes = [ctypes.c_int, ctypes.c_int, P, L, P, L, P, L, P, L, P, L, P, L, P, P, P, P, P, ctypes.c_int]
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def solve_beam_batched(x, E, I, fix, Fy, wy, n_threads: int = 1, out=None):
    """Same broadcasting rules as oracle.beam_oracle.solve_beam_batched / the C-ABI."""
    I = np.ascontiguousarray(I, dtype=np.float64)
    B, Ne = I.shape
    N = Ne + 1
    x = np.ascontiguousarray(x, dtype=np.float64)
    fix = np.ascontiguousarray(fix, dtype=np.uint8)
    E = np.ascontiguousarray(np.atleast_1d(np.asarray(E, dtype=np.float64)))
    wy = np.ascontiguousarray(np.atleast_1d(np.asarray(wy, dtype=np.float64)))
    Fy = np.ascontiguousarray(Fy, dtype=np.float64)
    assert Fy.shape == (B, N)
    if out is None:
        out = (np.empty((B, N)), np.empty((B, N)), np.empty((B, Ne)), np.empty((B, Ne)), np.empty(B, dtype=np.int32))
    v, th, V, M, st = out      # pass `out` to reuse result buffers (no page faults inside a timed region)
    lib().oracle_beam_solve_batched_f64(
        B, Ne,
        _p(x), N if x.ndim == 2 else 0,
        _p(E), Ne if E.ndim == 2 else 0,
        _p(I), Ne,
        _p(fix), N if fix.ndim == 2 else 0,
        _p(Fy), N,
        _p(wy), Ne if wy.ndim == 2 else 0,
        _p(v), _p(th), _p(V), _p(M), _p(st), int(n_threads),
    )
    return v, th, V, M, st
