"""Development: stress of the cooperative chain's LL exchange - many steps at several batch sizes, models created and destroyed,
results compared bit for bit with the flag protocol every few hundred steps; prints time-outs (must be 0) and mismatches (must be 0)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from climsim_amd.mlp import MLPEmulator
rng = np.random.default_rng(0)
bad = tout = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    B = int(rng.choice([64, 200, 256, 777, 1024, 1500, 2048]))
    units = [(512,) * 5, (512, 256, 128), (128, 128), (256, 512, 512)][rep % 4]
    x = torch.randn(B, 124, device="cuda") * 0.2
    y = torch.randn(B, 128, device="cuda") * 0.05
    ws = {}
    for ll in ("1", "0"):
        os.environ["CS_COOP_LL"] = ll
        m = MLPEmulator(units=units, max_batch=2048, seed=rep, cooperative=True)
        for s in range(600):
            m.train_on_batch(x, y, 1e-3)
        torch.cuda.synchronize()
        tout += m.coop_timeouts
        ws[ll] = [w.copy() for w in m.get_weights()]
        m.close()
    same = all(np.array_equal(a, b) for a, b in zip(ws["1"], ws["0"]))
    bad += 0 if same else 1
    print(rep, B, units, "identical" if same else "MISMATCH", "timeouts so far", tout, flush=True)
print("mismatches", bad, "timeouts", tout)
sys.exit(1 if bad or tout else 0)
