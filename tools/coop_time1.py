"""Development: step time of the cfg-MLP on the cooperative chain at one batch size (env decides the variant)."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from climsim_amd import _lib
from climsim_amd.mlp import MLPEmulator
for B in [int(v) for v in sys.argv[1:]] or [1024]:
    m = MLPEmulator(units=(512,) * 5, max_batch=B, seed=0, cooperative=True)
    x = torch.randn(B, 124, device="cuda") * 0.2
    y = torch.randn(B, 128, device="cuda") * 0.05
    for _ in range(30):
        m.train_on_batch(x, y, 1e-3)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(300):
            m.train_on_batch(x, y, 1e-3)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 300)
    with _lib.profile_session(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) as prof:
        for _ in range(60):
            m.train_on_batch(x, y, 1e-3)
    m.get_weights()
    print(B, "LL=%s WARM=%s" % (os.environ.get("CS_COOP_LL", "1"), os.environ.get("CS_COOP_WARM", "0")), "step us", round(best * 1e6, 1),
          {k: round(v[0] / 60 * 1e3, 1) for k, v in prof.times.items() if v[1]}, "timeouts", m.coop_timeouts, flush=True)
    m.close()
