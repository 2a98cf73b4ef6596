"""Many trials per GPU: the reference's hyper-parameter search trains ~8k small MLPs, several
workers per GPU, each a separate process (baseline_v1/hpo_baseline_v1.py:64-137 search space, :221-245 RandomSearch with
objective val_loss over 12 epochs, :255-260 workers_per_gpu).  A single small-batch step leaves most of an MI355X idle
(32..96 workgroups on 256 CUs, each streaming all weights: the step time is flat from 1024 to 8192 columns), so here one
process drives all trials and steps them TOGETHER: trials of one kernel family form a group (climsim_amd/group.py,
cs_mlp_group_*) whose step is one launch of all layer chains, one of all weight gradients and one of all optimisers; the
(few) families run on separate HIP streams, the data splits are shared in HBM.  `grouped=False` keeps the first form -
K engines on K streams, stepped round-robin - for comparison.

    pool = TrialPool([dict(units=(256, 384), activation="relu", optimizer="Adam", batch_size=3072), ...])
    results = pool.fit(x, y, validation_data=(xv, yv), epochs=12)        # [{"val_loss": ..., "history": ...}, ...]
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import numpy as np

from .mlp import CyclicalLearningRate, MLPEmulator

SEARCH_SPACE = {            # hpo_baseline_v1.py:66-74
    "num_layers": (2, 12), "units": (128, 1024, 128), "activation": ["relu", "elu", "leakyrelu"],
    "batch_size": [48, 96, 192, 384, 768, 1152, 1536, 2304, 3072], "optimizer": ["Adam", "RAdam", "RMSprop", "SGD"],
}


def sample_trial(rng: np.random.Generator) -> dict:
    """One draw from the reference's search space (RandomSearch)."""
    n = int(rng.integers(SEARCH_SPACE["num_layers"][0], SEARCH_SPACE["num_layers"][1] + 1))
    lo, hi, step = SEARCH_SPACE["units"]
    return {"units": tuple(int(rng.integers(lo // step, hi // step + 1)) * step for _ in range(n)),
            "activation": str(rng.choice(SEARCH_SPACE["activation"])), "optimizer": str(rng.choice(SEARCH_SPACE["optimizer"])),
            "batch_size": int(rng.choice(SEARCH_SPACE["batch_size"]))}


class TrialPool:
    def __init__(self, trials: Sequence[dict], device: Optional[int] = None, seed: int = 0, grouped: bool = True, **model_kw):
        import torch
        from .group import MLPGroup, group_by_family
        self.trials = [dict(t) for t in trials]
        self.models: List[MLPEmulator] = []
        for i, t in enumerate(self.trials):
            bs = int(t.get("batch_size", 3072))
            self.models.append(MLPEmulator(units=t["units"], activation=t.get("activation", "leakyrelu"),
                                           optimizer=t.get("optimizer", "Adam"), max_batch=max(bs, 4096), device=device,
                                           seed=seed + i, **model_kw))
        self.device = self.models[0].device
        # buckets of trials that step together: one per kernel family (grouped), or one per trial
        self.buckets = group_by_family(self.models) if grouped else [[i] for i in range(len(self.models))]
        self.groups = [MLPGroup([self.models[i] for i in b]) if (grouped and len(b) > 1) else None for b in self.buckets]
        self.streams = [torch.cuda.Stream(device=self.device) for _ in self.buckets]

    def close(self):
        for g in self.groups:
            if g is not None:
                g.close()
        for m in self.models:
            m.close()

    def fit(self, x, y, epochs: int = 12, validation_data=None, shuffle: bool = True, seed: int = 0,
            steps_per_epoch: Optional[int] = None, verbose: int = 0):
        """Train every trial for `epochs` epochs of its own batch size with the reference's cyclical schedule
        (2.5e-4..2.5e-3, step_size = 2*steps_per_epoch, hpo_baseline_v1.py:106-114); objective = final val_loss."""
        import torch
        m0 = self.models[0]
        x, y = m0._to_device(x, m0.input_length), m0._to_device(y, m0.output_length)
        val = None
        if validation_data is not None:
            val = (m0._to_device(validation_data[0], m0.input_length), m0._to_device(validation_data[1], m0.output_length))
        n = x.shape[0]
        K = len(self.models)
        bss = [int(t.get("batch_size", 3072)) for t in self.trials]
        steps = [steps_per_epoch or n // b for b in bss]
        if min(steps) < 1:
            raise ValueError("dataset smaller than one batch of some trial")
        scheds = [CyclicalLearningRate(2.5e-4, 2.5e-3, 2 * s) for s in steps]
        hist = [{"loss": [], "mae": [], "val_loss": [], "val_mae": []} for _ in range(K)]
        ep_sum = [torch.zeros(2, dtype=torch.float32, device=self.device) for _ in range(K)]
        st_loss = [torch.zeros(2, dtype=torch.float32, device=self.device) for _ in range(K)]
        gen = torch.Generator(device=self.device)
        cur = torch.cuda.current_stream(self.device)
        for epoch in range(epochs):
            perms = []
            for k in range(K):
                gen.manual_seed(seed + 1000 * k + epoch)
                perms.append(torch.randperm(n, device=self.device, generator=gen) if shuffle else torch.arange(n, device=self.device))
                ep_sum[k].zero_()
            for s in self.streams:
                s.wait_stream(cur)
            for step in range(max(steps)):                       # one step of every live trial per turn
                for b, (members, grp) in enumerate(zip(self.buckets, self.groups)):
                    live = [step < steps[k] for k in members]
                    if not any(live):
                        continue
                    with torch.cuda.stream(self.streams[b]):
                        if grp is None:
                            k = members[0]
                            idx = perms[k][step * bss[k]:(step + 1) * bss[k]]
                            self.models[k].train_on_batch(x, y, scheds[k](self.models[k].iterations), row_idx=idx, loss=st_loss[k])
                            ep_sum[k] += st_loss[k]
                        else:                                    # ONE launch per kernel kind for the whole bucket
                            idx = [perms[k][step * bss[k]:(step + 1) * bss[k]] if a else None for k, a in zip(members, live)]
                            lrs = [scheds[k](self.models[k].iterations) for k in members]
                            out = grp.train_on_batch(x, y, lrs, row_idx=idx, active=live)
                            for j, (k, a) in enumerate(zip(members, live)):
                                if a:
                                    ep_sum[k] += out[j]
            # epoch end: validation pass (the reference runs `validation_data` every epoch for every trial,
            # hpo_baseline_v1.py:139-150) - ONE launch per batch for a whole bucket (cs_mlp_group_forward), like the steps
            for b, (members, grp) in enumerate(zip(self.buckets, self.groups)):
                with torch.cuda.stream(self.streams[b]):
                    if val is not None:
                        evs = grp.evaluate(val[0], val[1]) if grp is not None else [self.models[members[0]].evaluate(val[0], val[1])]
                        for k, ev in zip(members, evs):
                            hist[k]["val_loss"].append(ev["loss"])
                            hist[k]["val_mae"].append(ev["mae"])
                    for k in members:
                        m = self.models[k]
                        tr = ep_sum[k].cpu().numpy().astype(np.float64) / (m.output_length * bss[k] * steps[k])
                        hist[k]["loss"].append(float(tr[0]))
                        hist[k]["mae"].append(float(tr[1]))
            for s in self.streams:
                cur.wait_stream(s)
            if verbose:
                print(f"epoch {epoch + 1}/{epochs} " + " ".join(f"[{k}] {h['loss'][-1]:.4g}" for k, h in enumerate(hist)), flush=True)
        out = []
        for k in range(K):
            obj = hist[k]["val_loss"][-1] if hist[k]["val_loss"] else hist[k]["loss"][-1]
            out.append({"trial": self.trials[k], "objective": obj if math.isfinite(obj) else math.inf, "history": hist[k],
                        "params": self.models[k].count_params()})
        return out
