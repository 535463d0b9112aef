#!/usr/bin/env python3
"""Where the HIP sizing loop's float32 rounding first differs from the reference's.  CPU only.

The reference's update is torch CPU autograd + torch.optim.Adam (SingleCore.py:195-208); csrc/sizing_math.hpp evaluates the same mathematics in
ONE fused expression per element.  Elementwise in I, so no summation order is involved -- what differs is the association and fusing of the float32
operations.  This script runs ONE optimiser step both ways on the same inputs (epoch 0 of the reference bridge: I = 0.5, M / V from the oracle's
solve) and reports, operation by operation, how many of the 100 elements differ in at least one bit.

    framework (autograd, executed)                                  kernel (csrc/sizing_math.hpp step_case)
    g_b = ((-a) * ((M^2 / den) / den)) * 2E                         a * ((M^2) / (den * den)) * 2E
    g_s = ((((-a) * ((V^2 / GA) / GA)) * G) * k) * (0.5 * (1/sqrt I))   a * ((V^2) / (GA * GA)) * (G * k * (0.5 / sqrt I))
    g   = (1 + g_s) + g_b        (AccumulateGrad order)             (1 - b) - s
    m'  = fma(0.1, g - m, m)     (lerp_)                            0.9 m + (1 - 0.9) g
    I'  = I + ((-step) * m') / denom     (addcdiv_)                 I - step * (m' / denom)
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import beam_oracle as bo  # noqa: E402
from oracle import c_oracle as co  # noqa: E402

f32 = np.float32


def kernel_step(I, M, V, m, v, t, E=bo.E_REF, lr=0.01, gamma=0.98, aM=1e-2, aV=1e-2, k=0.03, b1=0.9, b2=0.999, eps=1e-8):
    """csrc/sizing_math.hpp step_case, operation for operation, in numpy float32 (no FMA contraction: the library is built -ffp-contract=off)."""
    G = E / 2.6
    twoE, Gf = f32(2.0 * E), f32(G)
    step = f32(f32(lr * gamma ** t) / f32(1.0 - b1 ** (t + 1)))
    bc2s = f32(np.sqrt(1.0 - b2 ** (t + 1)))
    den_b = twoE * I + f32(1e-6)
    sq = np.sqrt(I)
    den_s = Gf * (f32(k) * sq)
    g = f32(1.0) - f32(aM) * ((M * M) / (den_b * den_b)) * twoE - f32(aV) * ((V * V) / (den_s * den_s)) * (Gf * f32(k) * (f32(0.5) / sq))
    ea = f32(b1) * m + (f32(1.0) - f32(b1)) * g
    es = f32(b2) * v + (f32(1.0) - f32(b2)) * g * g
    denom = np.sqrt(es) / bc2s + f32(eps)
    return np.maximum(I - step * (ea / denom), f32(1e-8)), g, ea, es


def main():
    x = np.linspace(0.0, bo.L_REF, bo.N_NODES_REF)
    fix = bo.reference_fix_mask()
    rng = np.random.default_rng(20250307)
    I0, Fy = bo.random_cases(rng, 1, inertia="uniform")
    out = {}
    for label, Istart in (("epoch 0 (I = 0.5)", np.full(100, 0.5, dtype=f32)), ("a trajectory state", np.exp(rng.uniform(np.log(3e-3), np.log(0.75), 100)).astype(f32))):
        _, _, V, M, st = co.solve_beam_batched(x, bo.E_REF, Istart.astype(np.float64)[None, :], fix, Fy, bo.UDL_REF)
        M32, V32 = M[0].astype(f32), V[0].astype(f32)
        # the framework, executed
        It = torch.tensor(Istart, dtype=torch.float32, requires_grad=True)
        opt = torch.optim.Adam([It], lr=0.01)
        G = bo.E_REF / 2.6
        Mt, Vt = torch.tensor(M32), torch.tensor(V32)
        loss = torch.sum(It) + 1e-2 * torch.sum(Mt ** 2 / (2 * bo.E_REF * It + 1e-6)) + 1e-2 * torch.sum(Vt ** 2 / (G * (0.03 * It ** 0.5)))
        loss.backward()
        g_ref = It.grad.detach().numpy().copy()
        opt.step()
        with torch.no_grad():
            It.clamp_(min=1e-8)
        I_ref = It.detach().numpy()
        ea_ref = opt.state[It]["exp_avg"].numpy()
        es_ref = opt.state[It]["exp_avg_sq"].numpy()
        I_k, g_k, ea_k, es_k = kernel_step(Istart, M32, V32, np.zeros(100, f32), np.zeros(100, f32), 0)
        ulp = lambda a, b: float((np.abs(a.astype(np.float64) - b.astype(np.float64)) / np.spacing(np.abs(b).astype(f32))).max())  # noqa: E731
        out[label] = {"elements": 100,
                      "gradient_differs_in_a_bit": int((g_k != g_ref).sum()), "gradient_max_ulps": ulp(g_k, g_ref),
                      "exp_avg_differs": int((ea_k != ea_ref).sum()), "exp_avg_sq_differs": int((es_k != es_ref).sum()),
                      "I_after_one_step_differs": int((I_k != I_ref).sum()), "I_after_one_step_max_ulps": ulp(I_k, I_ref)}
    print(json.dumps(out, indent=1))
    return out


if __name__ == "__main__":
    main()
