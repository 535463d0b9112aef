"""Per-case CPU restatement of the reference's sizing loop -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/OpenPyStruct_BeamOpt_training_SingleCore.py:163-249 line by line
(torch CPU autograd, torch.optim.Adam, ExponentialLR, clamp, early stopping, one-step lag of
the recorded responses), with the OpenSees model build + analyze + response queries
(:176-190, :224-232) replaced by the oracle FE solve (oracle/c_oracle.py).
PARITY UNPINNED for the FE part (see oracle/beam_oracle.py: openseespy is unavailable).
The LOOP is no longer "re-typed and trusted": it is checked against fixtures the reference's OWN code produced
(r05, tests/golden/sizing_reference_{sc,mc,gpu,sc_rb,mc_rb,bo}.npz: SingleCore / MultiCore / GPU `main()` + `generate_sample`
and the BeamOpt script executed in the build container with openseespy replaced by a recorder backed by the oracle's 3-DOF
solve, tests/golden/make_sizing_golden.py) -- epoch counts equal in all 60 runs, every epoch's loss equal to 1e-6, final
float32 inertias bit-equal in >= 80 % of the cases and within 2e-5 in the rest (tests/test_sizing_golden.py).
"""
from __future__ import annotations

import numpy as np
import torch
from torch.optim.lr_scheduler import ExponentialLR

from . import beam_oracle as bo
from . import c_oracle as co


def generate_sample(node_positions, roller_nodes, force_nodes, force_values, *, E=bo.E_REF, udl=bo.UDL_REF,
                    I_0=bo.I0_REF, max_e=600, lr=0.01, gamma=0.98, alpha_moment=1e-2, alpha_shear=1e-2,
                    tolerance=5e-3, patience=5, zero_last_node=False):
    """One sample.  `zero_last_node` reproduces MultiCore.py:222-223 (last node forced to 0.0)."""
    x = np.asarray(node_positions, dtype=np.float64)
    N = x.shape[0]
    Ne = N - 1
    G = E / (2 * (1 + bo.NU_REF))
    fix = np.zeros(N, dtype=np.uint8)
    fix[0] = 1
    for r in roller_nodes:
        fix[r - 1] = 1
    Fy = np.zeros((1, N))
    for n, F in zip(force_nodes, force_values):
        Fy[0, n - 1] += F

    I_tensor = torch.tensor([I_0] * Ne, dtype=torch.float32, requires_grad=True)     # :163
    optimizer = torch.optim.Adam([I_tensor], lr=lr)                                   # :166
    scheduler = ExponentialLR(optimizer, gamma=gamma)                                 # :167
    best_loss = float("inf")
    patience_counter = 0
    epochs = 0
    loss_history = []
    for epoch in range(max_e):                                                        # :174
        optimizer.zero_grad()
        I64 = I_tensor.detach().numpy().astype(np.float64)[None, :]                   # .item() widening, :107
        v, th, V, M, st = co.solve_beam_batched(x, E, I64, fix, Fy, udl)              # :176-182
        if st[0] != 0:
            break
        bending_moments = torch.tensor(M[0], dtype=torch.float32)                     # :189
        shear_forces = torch.tensor(V[0], dtype=torch.float32)                        # :190
        bending_energy = torch.sum((bending_moments ** 2) / (2 * E * I_tensor + 1e-6))  # :195
        A_approx = 0.03 * I_tensor ** 0.5                                             # :196
        shear_energy = torch.sum(shear_forces ** 2 / (G * A_approx))                  # :197
        primary_loss = torch.sum(I_tensor)                                            # :198
        total_loss = primary_loss + alpha_moment * bending_energy + alpha_shear * shear_energy
        total_loss.backward()                                                         # :202
        loss_history.append(float(total_loss.item()))
        optimizer.step()
        scheduler.step()
        with torch.no_grad():
            I_tensor.clamp_(min=1e-8)                                                 # :208
        epochs = epoch + 1
        if total_loss.item() < best_loss - tolerance:                                 # :211
            best_loss = total_loss.item()
            patience_counter = 0
        else:
            patience_counter += 1
        if patience_counter >= patience:
            break
    rotations = th[0].copy()
    deflections = v[0].copy()
    if zero_last_node:
        rotations[-1] = 0.0
        deflections[-1] = 0.0
    return {
        "roller_x_locations": [float(x[n - 1]) for n in roller_nodes],
        "force_x_locations": [float(x[n - 1]) for n in force_nodes],
        "force_values": [float(f) for f in force_values],
        "I_values": I_tensor.detach().numpy().tolist(),
        "shear_forces": shear_forces.detach().tolist(),
        "bending_moments": bending_moments.detach().tolist(),
        "node_positions": x.tolist(),
        "roller_nodes": list(roller_nodes),
        "force_nodes": [int(n) for n in force_nodes],
        "num_nodes": N,
        "L": float(x[-1]),
        "rotations": rotations.tolist(),
        "deflections": deflections.tolist(),
        "epochs_run": epochs,
        "final_loss": float(total_loss.item()),
        "loss_history": loss_history,
    }
