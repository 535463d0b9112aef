#!/usr/bin/env python3
"""Epoch time of the TFD loop with the FE-residual physics term (BASELINE config 4), one GPU; rocprofv3-friendly."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openpystruct_amd import dataprep, sizing, train  # noqa: E402

dev = torch.device("cuda", 0)
scfg = sizing.SizingConfig()
rec = sizing.generate_dataset(50000, scfg, dev)
phys = train.PhysicsTerm(weight=1e-3, x=torch.linspace(0, scfg.L_max, scfg.num_nodes, dtype=torch.float64), E=scfg.E,
                         fix=sizing.make_cases(1, scfg).fix[0], wy=scfg.uniform_udl)
d1 = dataprep.prepare(rec, kind="tfd", n_cases=1, device=dev)
r = train.train_surrogate("tfd", d1, train.TfdConfig(n_cases=1), device=dev, max_epochs=int(sys.argv[1]) if len(sys.argv) > 1 else 3, physics=phys)
ep = r["history"]["epoch_s"][1:]
print(json.dumps({"metric": "tfd + physics epoch time", "value": sum(ep) / len(ep), "steps_per_epoch": r["steps_per_epoch"]}))
