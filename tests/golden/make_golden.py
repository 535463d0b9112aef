"""Generates tests/golden/*.npz from the CPU oracle (oracle/beam_oracle.py).

PARITY UNPINNED (see oracle/beam_oracle.py header): openseespy is not importable here and the
reference holds no golden vectors, so these fixtures are outputs of the oracle itself, which
is pinned by closed-form beam theory (tests/test_oracle.py).  Re-run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import beam_oracle as bo  # noqa: E402


def fixed_bridge(seed, B, inertia):
    rng = np.random.default_rng(seed)
    x = np.linspace(0.0, bo.L_REF, bo.N_NODES_REF)
    fix = bo.reference_fix_mask()
    I, Fy = bo.random_cases(rng, B, inertia=inertia)
    v, th, V, M, st = bo.solve_beam_batched(x, bo.E_REF, I, fix, Fy, bo.UDL_REF)
    return dict(x=x, E=bo.E_REF, wy=bo.UDL_REF, fix=fix, I=I, Fy=Fy, v=v, theta=th, V=V, M=M, status=st)


def random_bridge(seed, B):
    """`random_bridge = 1` variant (SingleCore.py:133-151): per-beam length and 1-4 random rollers."""
    rng = np.random.default_rng(seed)
    N = bo.N_NODES_REF
    x = np.zeros((B, N))
    fix = np.zeros((B, N), dtype=np.uint8)
    Fy = np.zeros((B, N))
    I = np.exp(rng.uniform(np.log(3e-3), np.log(0.75), size=(B, N - 1)))
    for b in range(B):
        L = 15.0 + rng.uniform(0.0, 200.0)
        x[b] = np.linspace(0.0, L, N)
        nr = int(rng.integers(1, 5))
        rollers = rng.choice(np.arange(2, N), size=nr, replace=False)
        fix[b, 0] = 1
        fix[b, rollers - 1] = 1
        avail = np.array([n for n in range(2, N) if n not in rollers])
        k = int(rng.integers(1, 5))
        nodes = rng.choice(avail, size=k, replace=False)
        Fy[b, nodes - 1] = rng.uniform(bo.MAX_FORCE, bo.MIN_FORCE, size=k)
    v, th, V, M, st = bo.solve_beam_batched(x, bo.E_REF, I, fix, Fy, bo.UDL_REF)
    return dict(x=x, E=bo.E_REF, wy=bo.UDL_REF, fix=fix, I=I, Fy=Fy, v=v, theta=th, V=V, M=M, status=st)


def main():
    np.savez_compressed(os.path.join(HERE, "bridge_uniform.npz"), **fixed_bridge(20250307, 48, "uniform"))
    np.savez_compressed(os.path.join(HERE, "bridge_trajectory.npz"), **fixed_bridge(20250308, 48, "trajectory"))
    np.savez_compressed(os.path.join(HERE, "bridge_adversarial.npz"), **fixed_bridge(20250309, 24, "adversarial"))
    np.savez_compressed(os.path.join(HERE, "random_bridge.npz"), **random_bridge(20250310, 32))


if __name__ == "__main__":
    main()
