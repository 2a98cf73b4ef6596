"""Row shuffling for splits and chunks that live in HBM: the `.shuffle(buffer).batch(bs)` stage of the reference's input pipeline
(baseline_models/MLP/training/HPO/baseline_v1/step2_retrain/step2_retrain.py:266-277) as ONE small kernel.

`device_permutation(n, seed)` returns a permutation of 0..n-1 as an int64 device tensor - the row indices the engine's training
kernels gather by (`cs_permutation`, csrc/kernels.h: a keyed 4-round Feistel network over the index bits, cycle-walked into range;
a bijection by construction).  `torch.randperm` does the same job with a key-generation + radix-sort pipeline that costs 0.09 ms for
the 172,800 rows of a streamed high-res chunk - 3 % of the chunk's training time; this launch takes a few microseconds."""
from __future__ import annotations

import ctypes as C

from . import _lib


def device_permutation(n: int, seed: int, device=None, out=None):
    import torch
    if not torch.cuda.is_available():
        raise _lib.EngineError("device_permutation needs a ROCm GPU (no CPU fallback)")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if out is None:
        out = torch.empty(int(n), dtype=torch.int64, device=dev)
    elif out.dtype != torch.int64 or out.numel() < n or not out.is_contiguous():
        raise ValueError("out must be a contiguous int64 tensor of at least n elements")
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(_lib.load().cs_permutation(int(n), C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), C.c_void_p(out.data_ptr()), st))
    return out[:n]


def chunk_seed(seed: int, counter: int) -> int:
    """Seed of the `counter`-th permutation drawn under `seed` (splitmix64 step: distinct, well-mixed 64-bit keys)."""
    z = (int(seed) * 0x9E3779B97F4A7C15 + (int(counter) + 1) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z ^= z >> 30
    z = (z * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z ^= z >> 27
    z = (z * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)
