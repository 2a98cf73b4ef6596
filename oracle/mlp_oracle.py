"""CPU ORACLE for the MLP hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product path (climsim_amd/) never does and fails loudly when the HIP library is missing.

What it restates (numpy, float32 unless noted):
  * model      : baseline_models/MLP/training/HPO/baseline_v1/step2_retrain/step2_retrain.py:95-126
                 (= hpo_baseline_v1.py:75-103): Dense->act per hidden layer, Dense(128)->act,
                 heads Dense(120, linear) || Dense(8, relu), Concatenate.
  * loss       : compile(loss='mse') step2_retrain.py:160-162 -> mean over batch x 128.
  * optimisers : step2_retrain.py:140-157.  Their arithmetic lives in un-vendored wheels pinned by
                 baseline_models/MLP/env/environment.yml:119,121,287 (tensorflow 2.11.1,
                 keras 2.11.0, tensorflow-addons 0.19.0); restated here from the published update
                 rules.
  * LR schedule: tfa.optimizers.CyclicalLearningRate, triangular2 (step2_retrain.py:140-148).
  * normalise  : climsim_utils/data_utils.py:807-809 and the inf/nan->0 rule :894-897.

PARITY: PINNED (round 6) for topology, head fusion, ReLU, MSE, the hand-derived backward and the torch-Adam rule: the reference's own
torch MLP (online_testing/baseline_models/MLP_v2rh/training/mlp.py:28-67: Linear -> ReLU per hidden layer, a final Linear, ReLU on
the last 8 outputs) instantiated as 124 -> [512]*5 + [128] -> 128 IS this model with act='relu' (the two heads = one 128 x 128 layer
with ReLU on columns 120..127), and as 124 -> [768, 640, 512, 640, 640, 128] -> 128 the published one.  tests/golden/hot_mlp_golden.npz
holds what the reference computes for them at batch 8192 / 3072 (loss, predictions, every gradient, five torch.optim.Adam steps;
made by tests/golden/make_online_mlp_golden.py, which imports the reference in the build container); tests/test_hot_mlp_cpu.py holds
`forward` / `loss_and_grads(bf16=False)` / `Optimizer('AdamTorch')` to those vectors at float32 accumulation-order tolerance (3e-6;
loss 1e-6).  UNPINNED (TensorFlow / Keras cannot be installed here, the reference holds no vectors for them): the LeakyReLU / ELU
epilogues and the Keras-2.11 / tfa-0.19 optimiser rules - restated from the published definitions, cross-checked
against an independent torch-autograd implementation (oracle/mlp_torch_cpu.py) and against torch.optim with the documented
epsilon-placement differences asserted (tests/test_oracle.py); the known answers that exist (parameter count 1,753,472 and 3,503,488
forward FLOPs for the published model, step1_results.csv:170 / FLOP_calculation.ipynb nb:231) are asserted there.

`bf16=True` reproduces the rounding points of the HIP engine (operands of every contraction
rounded to bfloat16 round-to-nearest-even, fp32 accumulation, fp32 master weights) so that GPU
results can be compared at accumulation-order tolerance rather than at bf16 tolerance.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Sequence

import numpy as np

F32 = np.float32


@dataclass
class MLPConfig:
    n_in: int = 124
    hidden: Sequence[int] = (512, 512, 512, 512, 512)
    n_out_lin: int = 120
    n_out_relu: int = 8
    act: str = "leakyrelu"          # 'relu' | 'elu' | 'leakyrelu'
    alpha: float = 0.15             # LeakyReLU slope (step2_retrain.py:110)

    @property
    def n_out(self):
        return self.n_out_lin + self.n_out_relu

    @property
    def dims(self):
        """Layer widths including the 128-wide 'upper output' layer (step2_retrain.py:113)."""
        return [self.n_in, *self.hidden, self.n_out]

    def n_params(self):
        d = self.dims
        trunk = sum(d[i] * d[i + 1] + d[i + 1] for i in range(len(d) - 1))
        return trunk + self.n_out * self.n_out + self.n_out

    def fwd_flops(self):
        """2*weights + biases, the keras_flops convention (FLOP_calculation.ipynb nb:231)."""
        d = self.dims
        w = sum(d[i] * d[i + 1] for i in range(len(d) - 1)) + self.n_out * self.n_out
        b = sum(d[1:]) + self.n_out
        return 2 * w + b

    def train_flops(self):
        """fwd + wgrad + dgrad MACs x2 (no dgrad for the first layer); SURVEY.md section 8 a7."""
        d = self.dims
        w = sum(d[i] * d[i + 1] for i in range(len(d) - 1)) + self.n_out * self.n_out
        return 2 * (3 * w - d[0] * d[1])


def bf16_round(a):
    """float32 -> nearest bfloat16 (ties to even), returned as float32."""
    a = np.ascontiguousarray(a, dtype=F32)
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u >> 16) & 1) + 0x7FFF
    out = ((u + r) & 0xFFFF0000).astype(np.uint32).view(F32)
    return np.where(np.isnan(a), a, out).astype(F32)


def glorot_init(cfg: MLPConfig, seed: int) -> List[np.ndarray]:
    """Keras-ordered weight list [W0,b0,...,W_up,b_up,W_lin,b_lin,W_relu,b_relu]; glorot_uniform
    kernels (limit sqrt(6/(fan_in+fan_out))), zero biases -- the Dense defaults."""
    rng = np.random.default_rng(seed)
    ws = []

    def dense(k, n):
        lim = np.sqrt(6.0 / (k + n))
        ws.append(rng.uniform(-lim, lim, size=(k, n)).astype(F32))
        ws.append(np.zeros(n, dtype=F32))
    d = cfg.dims
    for i in range(len(d) - 1):
        dense(d[i], d[i + 1])
    dense(cfg.n_out, cfg.n_out_lin)
    dense(cfg.n_out, cfg.n_out_relu)
    return ws


def fuse_heads(ws: List[np.ndarray]):
    """Keras list -> [(W,b)] with the two heads concatenated into one 128x128 layer
    (Concatenate([lin, relu]) == one Dense with a per-column activation)."""
    pairs = [(ws[i], ws[i + 1]) for i in range(0, len(ws) - 4, 2)]
    pairs.append((np.concatenate([ws[-4], ws[-2]], axis=1), np.concatenate([ws[-3], ws[-1]])))
    return pairs


def split_heads(pairs, n_lin):
    ws = []
    for w, b in pairs[:-1]:
        ws += [w, b]
    w, b = pairs[-1]
    ws += [w[:, :n_lin], b[:n_lin], w[:, n_lin:], b[n_lin:]]
    return ws


def normalise(x, sub, div):
    """(x - sub)/div in float32 with inf/nan -> 0 (data_utils.py:807-809, :894-897)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        xn = (np.asarray(x, F32) - np.asarray(sub, F32)) / np.asarray(div, F32)
    xn[~np.isfinite(xn)] = 0
    return xn.astype(F32)


def _act(z, kind, alpha):
    if kind == "relu":
        return np.maximum(z, 0)
    if kind == "leakyrelu":
        return np.where(z > 0, z, F32(alpha) * z)
    if kind == "elu":
        return np.where(z > 0, z, np.expm1(np.minimum(z, 0)))
    raise ValueError(kind)


def _act_grad_from_h(h, kind, alpha):
    """d act / d z expressed through the activation output h (what the engine stores)."""
    if kind == "relu":
        return (h > 0).astype(F32)
    if kind == "leakyrelu":
        return np.where(h > 0, F32(1), F32(alpha)).astype(F32)
    if kind == "elu":
        return np.where(h > 0, F32(1), h + F32(1)).astype(F32)
    raise ValueError(kind)


def _mm(a, b):
    """float32 result of a float64-accumulated product (order-independent reference)."""
    return (a.astype(np.float64) @ b.astype(np.float64)).astype(F32)


def forward(ws, x, cfg: MLPConfig, bf16=False, keep=False):
    """x: (B, n_in) float32 already normalised.  Returns yhat (B,128) float32 [, activations]."""
    q = bf16_round if bf16 else (lambda a: np.asarray(a, F32))
    pairs = fuse_heads(ws)
    h = q(x)
    hs = [h]
    for w, b in pairs[:-1]:
        h = q(_act(_mm(h, q(w)) + b, cfg.act, cfg.alpha).astype(F32))
        hs.append(h)
    w, b = pairs[-1]
    z = _mm(h, q(w)) + b
    yhat = z.copy()
    yhat[:, cfg.n_out_lin:] = np.maximum(z[:, cfg.n_out_lin:], 0)
    return (yhat, hs) if keep else yhat


def loss_and_grads(ws, x, y, cfg: MLPConfig, bf16=False):
    """MSE loss (float64 scalar), Keras-ordered gradient list of d(mean sq err)/d(param)."""
    q = bf16_round if bf16 else (lambda a: np.asarray(a, F32))
    pairs = fuse_heads(ws)
    yhat, hs = forward(ws, x, cfg, bf16=bf16, keep=True)
    B = x.shape[0]
    e = yhat - np.asarray(y, F32)
    loss = float(np.mean(e.astype(np.float64) ** 2))
    mae = float(np.mean(np.abs(e.astype(np.float64))))
    dz = (2 * e).astype(F32)                                   # 1/(128 B) applied at the end
    dz[:, cfg.n_out_lin:] *= (yhat[:, cfg.n_out_lin:] > 0)     # relu head
    dz = q(dz)
    scale = F32(1.0 / (cfg.n_out * B))
    grads = [None] * len(pairs)
    for li in range(len(pairs) - 1, -1, -1):
        w, _ = pairs[li]
        h_in = hs[li]
        gw = _mm(h_in.T, dz) * scale
        gb = dz.astype(np.float64).sum(axis=0).astype(F32) * scale
        grads[li] = (gw.astype(F32), gb.astype(F32))
        if li > 0:
            dh = _mm(dz, q(w).T)
            dz = q((dh * _act_grad_from_h(h_in, cfg.act, cfg.alpha)).astype(F32))
    return loss, mae, split_heads(grads, cfg.n_out_lin), yhat


def cyclical_lr(it, init_lr=2.5e-4, max_lr=2.5e-3, step_size=16):
    """tfa CyclicalLearningRate with scale_fn = 1/2^(cycle-1), scale_mode='cycle'
    (step2_retrain.py:140-148)."""
    cycle = np.floor(1 + it / (2 * step_size))
    xx = np.abs(it / step_size - 2 * cycle + 1)
    return float(init_lr + (max_lr - init_lr) * max(0.0, 1 - xx) * (1 / 2.0 ** (cycle - 1)))


@dataclass
class Optimizer:
    """Keras 2.11 Adam / RMSprop / SGD, tfa 0.19 RectifiedAdam and torch.optim.Adam ('AdamTorch') update rules, float32 state."""
    kind: str = "Adam"
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-7
    rho: float = 0.9
    sma_threshold: float = 5.0
    it: int = 0
    m: list = field(default_factory=list)
    v: list = field(default_factory=list)

    def apply(self, ws, grads, lr):
        """One update.  Scalars are float32 and cast where TensorFlow casts them (float32 variables):
        beta^t = pow(float32(beta), float32(t)); Keras Adam multiplies by float32(1-beta) of the
        Python doubles, tfa RAdam by 1-float32(beta); RMSprop uses rsqrt(v+eps)."""
        if not self.m:
            self.m = [np.zeros_like(w) for w in ws]
            self.v = [np.zeros_like(w) for w in ws]
        t = F32(self.it + 1)
        lr = F32(lr)
        one = F32(1)
        b1, b2, eps = F32(self.beta1), F32(self.beta2), F32(self.eps)
        p1, p2 = np.power(b1, t, dtype=F32), np.power(b2, t, dtype=F32)
        out = []
        for i, (w, g) in enumerate(zip(ws, grads)):
            g = g.astype(F32)
            if self.kind == "SGD":
                w = w - lr * g
            elif self.kind == "RMSprop":
                self.v[i] = F32(self.rho) * self.v[i] + F32(1 - self.rho) * (g * g)
                w = w - (lr * g) * (one / np.sqrt(self.v[i] + eps))
            elif self.kind == "Adam":
                self.m[i] = self.m[i] + (g - self.m[i]) * F32(1 - self.beta1)
                self.v[i] = self.v[i] + (g * g - self.v[i]) * F32(1 - self.beta2)
                alpha = lr * np.sqrt(one - p2) / (one - p1)
                w = w - (self.m[i] * alpha) / (np.sqrt(self.v[i]) + eps)
            elif self.kind == "AdamTorch":
                # torch.optim.Adam (_single_tensor_adam; online_testing/baseline_models/MLP_v2rh/training/train_mlp_h5loader.py:210-211):
                # the bias corrections are Python doubles, eps sits INSIDE the correction of v (default 1e-8).  Pinned by reference
                # output: tests/test_hot_mlp_cpu.py (five steps of the reference's own optimiser on the benchmarked topologies).
                td = float(self.it + 1)
                step = F32(float(lr) / (1.0 - self.beta1 ** td))
                bc2_sqrt = F32(np.sqrt(1.0 - self.beta2 ** td))
                self.m[i] = (self.m[i] + F32(1 - self.beta1) * (g - self.m[i])).astype(F32)
                self.v[i] = (self.v[i] * b2 + (F32(1 - self.beta2) * g) * g).astype(F32)
                w = w - (step * self.m[i]) / (np.sqrt(self.v[i]) / bc2_sqrt + eps)
            elif self.kind == "RAdam":
                self.m[i] = b1 * self.m[i] + (one - b1) * g
                self.v[i] = b2 * self.v[i] + (one - b2) * (g * g)
                sma_inf = F32(2) / (one - b2) - one
                sma_t = sma_inf - F32(2) * t * p2 / (one - p2)
                m_hat = self.m[i] / (one - p1)
                if sma_t >= F32(self.sma_threshold):
                    r = np.sqrt((sma_t - F32(4)) / (sma_inf - F32(4)) * (sma_t - F32(2)) / (sma_inf - F32(2))
                                * sma_inf / sma_t)
                    w = w - lr * (r * m_hat / (np.sqrt(self.v[i] / (one - p2)) + eps))
                else:
                    w = w - lr * m_hat
            else:
                raise ValueError(self.kind)
            out.append(w.astype(F32))
        self.it += 1
        return out


def train_step(ws, opt: Optimizer, x, y, cfg: MLPConfig, lr, bf16=False):
    loss, mae, grads, _ = loss_and_grads(ws, x, y, cfg, bf16=bf16)
    return opt.apply(ws, grads, lr), loss, mae


def synth_columns(n, seed=20230614, n_in=124, n_out=128):
    """Low-res-shaped synthetic columns: normalised inputs ~N(0,0.15^2)
    clipped to [-1,1] (profiles), U(-0.5,0.5) scalars with SOLIN zeroed on half the rows;
    targets tanh(xA)*0.05 + noise, heads >= 0, upper-level moisture tendencies exactly 0."""
    rng = np.random.default_rng(seed)
    x = np.empty((n, n_in), dtype=F32)
    x[:, :120] = np.clip(rng.normal(0, 0.15, size=(n, 120)), -1, 1)
    x[:, 120:] = rng.uniform(-0.5, 0.5, size=(n, n_in - 120))
    x[rng.random(n) < 0.5, 121] = 0
    a = np.random.default_rng(20230614).normal(0, 1 / np.sqrt(n_in), size=(n_in, n_out)).astype(F32)
    y = (np.tanh(x @ a) * 0.05 + rng.normal(0, 0.01, size=(n, n_out))).astype(F32)
    y[:, 120:] = np.maximum(y[:, 120:], 0)
    y[:, 60:72] = 0
    return x, y
