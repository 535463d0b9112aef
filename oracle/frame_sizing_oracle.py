"""Per-frame CPU restatement of the reference's frame sizing loop -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/OpenPyStruct_FrameOpt_Discrete_Beta.py:141-206 line by line: `compute_combined_loss`
(:141-160: Python-float element forces against a float32 inertia tensor, `+ 1e-8` in the bending term, element-by-element
accumulation), float32 `I_tensor` (:167), `torch.optim.Adam(lr)` WITHOUT a scheduler (:170), `clamp_(min=1e-8)` (:187),
early stop on `best_loss - tolerance` with patience 10 (:192-203).  The OpenSees model build + `ops.analyze(1)` +
`ops.eleResponse(e, 'forces')` (:75-139, :181-183, :151) are replaced by the oracle's 3-DOF band solve
(oracle/beam_oracle.py::solve_model_3dof = LAPACK dpbsv).  PARITY UNPINNED for the FE part (openseespy is unavailable).
The LOOP is checked against what the reference's own script did (r05, tests/golden/sizing_reference_fr.npz: six runs of
OpenPyStruct_FrameOpt_Discrete_Beta.py executed under the recorder of tests/golden/opensees_stub.py, 1 x 2 ... 10 x 5 bays x
stories; tests/test_sizing_golden.py: loss history to 2e-6, the run that stops early at its epoch, inertias to 5e-5).
"""
from __future__ import annotations

import numpy as np
import torch

from . import beam_oracle as bo


def optimize_frame(coords, conn, fix3, nodal_loads, wy, wx, *, A=0.02, E=200e9, nu=0.3, I0=5e-4, alpha_moment=1e-2,
                   alpha_shear=1e-2, k=0.03, num_epochs=5000, lr=0.005, tolerance=1e-3, patience=10, I_init=None):
    """One frame.  Returns dict(I [Ne] float32 after the last step, I_history [epochs, Ne] (state AFTER each step),
    loss_history, epochs_run, forces of the last solve)."""
    G = E / (2 * (1 + nu))                                                             # :27
    total_elems = len(conn)
    I_values = [I0 for _ in range(total_elems)] if I_init is None else [float(v) for v in I_init]
    I_tensor = torch.tensor(I_values, dtype=torch.float32, requires_grad=True)         # :167
    optimizer = torch.optim.Adam([I_tensor], lr=lr)                                    # :170
    loss_history, I_history = [], []
    best_loss = float("inf")
    no_improve = 0
    forces = None
    for epoch in range(num_epochs):                                                    # :178
        optimizer.zero_grad()
        I64 = np.array([I_tensor[e].item() for e in range(total_elems)])               # `.item()` widening, :105
        disp, forces, st, _, _ = bo.solve_model_3dof(coords, conn, A, E, I64, fix3, nodal_loads, wy=wy, wx=wx)   # :180-182
        if st != 0:
            break
        bending_energy = 0.0
        shear_energy = 0.0
        for elem_id in range(1, total_elems + 1):                                      # :148-158
            response = [float(v) for v in forces[elem_id - 1]]
            shear_force = response[1]
            bending_moment = response[2]
            I_val = I_tensor[elem_id - 1]
            bending_energy += (bending_moment ** 2) / (2 * E * I_val + 1e-8)
            A_local = k * (I_val ** 0.5)
            shear_energy += (shear_force ** 2) / (G * A_local)
        primary_loss = torch.sum(I_tensor)
        total_loss = primary_loss + alpha_moment * bending_energy + alpha_shear * shear_energy
        total_loss.backward()                                                          # :184
        optimizer.step()
        with torch.no_grad():
            I_tensor.clamp_(min=1e-8)                                                  # :187-188
        current_loss = total_loss.item()
        loss_history.append(current_loss)
        I_history.append(I_tensor.detach().numpy().copy())
        if current_loss < best_loss - tolerance:                                       # :193-197
            best_loss = current_loss
            no_improve = 0
        else:
            no_improve += 1
        if no_improve >= patience:                                                     # :202-204
            break
    return {"I": I_tensor.detach().numpy().copy(), "I_history": np.array(I_history), "loss_history": np.array(loss_history),
            "epochs_run": len(loss_history), "forces": forces}
