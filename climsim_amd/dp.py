"""Data-parallel step: one process per GPU, ONE all-reduce of the flat gradient buffer per step.

The reference's baseline MLP trains on a single GPU (hpo_baseline_v1.py:264); the only gradient-
synchronous training in the repo is torch DDP/NCCL in online_testing
(train_mlp_h5loader.py:195-207: bucketed all-reduce(SUM)/world, DistributedSampler :114-126).
Here columns are i.i.d. rows, so every global batch is dealt round-robin to the ranks
(rank r takes rows r, r+world, ... of the batch's slice of the epoch permutation), each rank runs
forward/backward on its shard, the UNSCALED gradient sums are added with a single
`all_reduce(SUM)` (RCCL over xGMI with backend "nccl"; gloo on CPU for tests) and the optimiser
kernel applies 1/(128*global_batch).  Every rank then holds bit-identical weights (same reduced
buffer, same deterministic update), so no parameter broadcast is needed after the initial one.

`engine` is anything with `loss_grads(x, y, row_idx=, loss=)`, `gradient_tensor()` and
`apply_gradients(lr, grad_scale)` - MLPEmulator in production, a CPU stand-in in the gloo tests.
"""
from __future__ import annotations


def shard_of_batch(perm, step: int, global_batch: int, rank: int, world: int):
    """Rows of global batch `step` owned by `rank`: a strided view of the epoch permutation."""
    sl = perm[step * global_batch + rank:(step + 1) * global_batch:world]
    return sl.contiguous() if hasattr(sl, "contiguous") and world > 1 else sl


class DataParallel:
    def __init__(self, engine, dist=None, output_length: int = 128):
        self.engine, self.dist = engine, dist
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self.output_length = output_length
        self.grad = engine.gradient_tensor()

    def broadcast_weights(self):
        """Rank 0's weights to everyone (DDP's initial parameter broadcast)."""
        if self.dist is None:
            return
        import numpy as np
        import torch
        ws = self.engine.get_weights()
        flat = torch.from_numpy(np.concatenate([w.ravel() for w in ws])).to(self.grad.device)
        self.dist.broadcast(flat, src=0)
        flat = flat.cpu().numpy()
        out, at = [], 0
        for w in ws:
            out.append(flat[at:at + w.size].reshape(w.shape))
            at += w.size
        self.engine.set_weights(out)

    def train_step(self, x, y, perm, step: int, global_batch: int, lr: float, loss=None, normalise=False):
        """One optimiser step on global batch `step` of the permutation.  Returns the local loss sums."""
        if global_batch % self.world:
            raise ValueError("global batch must be divisible by the world size")
        idx = shard_of_batch(perm, step, global_batch, self.rank, self.world)
        out = self.engine.loss_grads(x, y, row_idx=idx, loss=loss, normalise=normalise)
        if self.dist is not None:
            self.dist.all_reduce(self.grad)                       # the ONE collective of the step
        self.engine.apply_gradients(lr, 1.0 / (self.output_length * global_batch))
        return out
