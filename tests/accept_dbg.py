import sys, os, copy, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import mlp_oracle as O
from oracle.mlp_torch_cpu import TorchMLP
from climsim_amd import mlp as M
def columns(n, seed):
    xx, _ = O.synth_columns(n, seed=seed)
    A = np.random.default_rng(7).normal(0, 1 / np.sqrt(124), (124, 128)).astype(np.float32)
    yy = np.tanh(3.0 * xx @ A) * 0.3 + np.random.default_rng(seed + 1000).normal(0, 0.01, (n, 128)).astype(np.float32)
    yy[:, 120:] = np.maximum(yy[:, 120:], 0); yy[:, 60:72] = 0
    return xx, yy.astype(np.float32)
units, bs = (256, 256), 1024
steps = int(sys.argv[1])
cfg = O.MLPConfig(hidden=units, act="leakyrelu"); ws = O.glorot_init(cfg, 3)
x, y = columns(32 * bs, 31); xs, ys = columns(12 * 384, 32)
res = {}
for name, seed in (("engine", 0), ("engine_b", 1)):
    m = M.MLPEmulator(units=units, activation="leakyrelu", optimizer="Adam", max_batch=4608, seed=None)
    m.set_weights(ws)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    order = np.arange(32) if seed == 0 else np.random.default_rng(5).permutation(32)
    for it in range(steps):
        lo = order[it % 32] * bs
        m.train_on_batch(xd[lo:lo + bs], yd[lo:lo + bs], 1e-3 if it < steps * 3 // 4 else 1e-4)
    res[name] = m.predict(xs)
cpu = TorchMLP(ws, cfg)
xt, yt = torch.from_numpy(x), torch.from_numpy(y)
for it in range(steps):
    lo = (it % 32) * bs
    _, g = cpu.loss_and_grads(xt[lo:lo + bs], yt[lo:lo + bs]); cpu.adam(g, 1e-3 if it < steps * 3 // 4 else 1e-4)
with torch.no_grad():
    res["cpu"] = cpu.forward(torch.from_numpy(xs)).numpy()
for k, p in res.items():
    print(k, "mse", float(np.mean((p - ys) ** 2)), "mae", float(np.mean(np.abs(p - ys))))
for k, p in res.items():
    print(k, "heads mae", np.round(np.mean(np.abs(p - ys), axis=0)[120:] * 1e3, 2), "profile mae", round(float(np.mean(np.abs(p - ys)[:, :120])) * 1e3, 3))
