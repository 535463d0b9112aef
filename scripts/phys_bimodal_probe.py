#!/usr/bin/env python3
"""The TFD + FE-residual epoch is 12.4 ms in most bench runs and 16.9 ms in some.  Per-epoch times of several trainings in one process
(argument: repeats), to be run several times as separate processes."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpystruct_amd import dataprep, runtime, sizing, train  # noqa: E402
runtime.configure()
dev = torch.device("cuda", 0)
scfg = sizing.SizingConfig()
rec = sizing.generate_dataset(50000, scfg, dev)
phys = train.PhysicsTerm(weight=1e-3, x=torch.linspace(0, scfg.L_max, scfg.num_nodes, dtype=torch.float64), E=scfg.E,
                         fix=sizing.make_cases(1, scfg).fix[0], wy=scfg.uniform_udl)
d1 = dataprep.prepare(rec, kind="tfd", n_cases=1, device=dev)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    r = train.train_surrogate("tfd", d1, train.TfdConfig(n_cases=1), device=dev, max_epochs=8, physics=phys)
    print("rep", rep, " ".join(f"{1e3 * t:.2f}" for t in r["history"]["epoch_s"]), flush=True)
