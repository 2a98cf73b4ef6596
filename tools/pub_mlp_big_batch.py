"""Development: published-architecture step time above 8192 columns: wide chain (CS_CHAINW_MAX_N) vs one GEMM per layer."""
import sys, time, os, torch
sys.path.insert(0, "/root/repo")
from climsim_amd.mlp import MLPEmulator
for B in [int(v) for v in sys.argv[1:]] or (12288, 16384, 32768):
    m = MLPEmulator(units=(768, 640, 512, 640, 640), activation="leakyrelu", optimizer="RAdam", max_batch=B, seed=0)
    x = (torch.rand((B, 124), device="cuda") - 0.5).contiguous(); y = (torch.rand((B, 128), device="cuda") * 0.1).contiguous()
    for _ in range(5): m.train_on_batch(x, y, 1e-3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): m.train_on_batch(x, y, 1e-3)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print(os.environ.get("CS_CHAINW_MAX_N", "8192"), B, "ms/step", round(dt * 1e3, 4), "Mcol/s", round(B / dt / 1e6, 2))
    m.close()
