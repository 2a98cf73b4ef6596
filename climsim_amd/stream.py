"""Streamed training from raw timestep fields (BASELINE config 5: high-res 21,600-column grids).

The reference trains from files through `data_utils.load_ncdata_with_generator` -> `tf.data` `shuffle(buffer)` ->
`batch` -> `prefetch` (baseline_models/MLP/training/HPO/baseline_v1/step2_retrain/step2_retrain.py:266-277; generator
`climsim_utils/data_utils.py:791-881`): a producer fills a buffer of loaded timesteps while the model consumes batches
drawn from it.  Here the producer is the device loader (`cs_loader_stack`, one HBM-bound kernel that does tendencies,
normalisation, inf/nan -> 0, stacking and the float32 cast for a CHUNK of timesteps) on a side HIP stream, the consumer
is the training step on the main stream, and the buffer is a ring of `slots` chunks in HBM: while the engine trains on
chunk k (batches drawn from a permutation of its rows, gathered inside the first kernel), chunk k+1 is being produced.
Raw chunks may be host arrays (pinned staging + asynchronous copy on the side stream) or device tensors (a raw shard
resident in HBM, the high-res layout of SURVEY section 8d).  Under data parallelism every rank streams its own
timesteps; the step is `DataParallel.train_step`'s (one all-reduce of the flat gradient).  Ranks agree per chunk on the
common row count (the smallest rank's; surplus rows of larger chunks are dropped and counted in `rows_dropped`) and the
pass ends when the first rank runs out of chunks, so every rank issues the same collectives.

PyTorch provides streams, events and memory only; there is no CPU execution path.
"""
from __future__ import annotations

from typing import Callable, Iterable

import numpy as np


class StreamedTrainer:
    def __init__(self, model, loader, batch_size: int, slots: int = 2, dist=None, loader_on: str | None = None, carry_remainder: bool = True,
                 shuffle: str = "feistel", side_priority: int | None = None):
        """`loader_on`: "main" (default; CS_STREAM_LOADER overrides) runs the loader KERNEL on the training stream, right in front of
        the chunk's first step; host chunks are staged and copied on the side stream while the PREVIOUS chunk trains (the training
        stream waits for the copy only when it reaches the chunk - round 4: it used to wait at produce time, in front of the
        previous chunk's steps, and idled through the whole copy).  "side" runs the kernel on the side stream too.  On one GPU the
        two measure the same for device-resident chunks (round 3, bench_stream.py: 56.0 / 56.2 M columns/s; the loader is
        1.3-1.6 G columns/s); with the kernel on the training stream nothing ever runs beside a step, so a cooperative model is
        accepted.
        `carry_remainder` (round 4): the rows a chunk's last batch would leave short are carried into the next chunk's permutation
        instead of forming a partial batch per chunk - the reference batches AFTER its shuffle buffer, across file boundaries
        (`.unbatch().shuffle(...).batch(bs)`, step2_retrain.py:266-277), so only the last batch of a pass is short there too.  At
        the high-res width a chunk of 8 timesteps is 21 batches of 8192 + 768 rows: the partial step costs two thirds of a full
        one.  With one pass per chunk only (`passes_per_chunk` > 1 keeps per-chunk batching); False restores rounds 1-3.
        `shuffle`: "feistel" (default, round 4) draws a chunk's permutation with `cs_permutation` (climsim_amd/shuffle.py: one small
        kernel, seeds `chunk_seed(seed, k)` for the k-th permutation of a pass); "torch" draws `torch.randperm` from a generator
        seeded once per pass (rounds 1-3: 0.09 ms of sort kernels per high-res chunk)."""
        import os
        import torch
        self.loader_on = loader_on or os.environ.get("CS_STREAM_LOADER", "main")
        if self.loader_on not in ("main", "side", "gaps"):
            raise ValueError("loader_on must be 'main', 'side' or 'gaps'")
        # "gaps" (round 5, an experiment that is kept for its measurement - profiles/r05_stream_gaps.txt): the loader of chunk k + 1
        # is cut into `gap_slices` launches of a few timesteps each, and launch j goes out on the side stream gated on an event of the
        # training stream inside step j of chunk k: `gap_at` = "opt" (between the weight-gradient kernel and the optimiser: the slice
        # runs beside k_optimizer) or "chain" (in front of the step: the slice runs beside the layer chain).
        self.gap_at = os.environ.get("CS_STREAM_GAP", "opt")
        self.gap_slices = int(os.environ.get("CS_STREAM_SLICES", "0"))        # 0 = one slice per timestep
        if self.gap_at not in ("opt", "chain"):
            raise ValueError("CS_STREAM_GAP must be 'opt' or 'chain'")
        if slots < 2:
            raise ValueError("need at least two chunk slots (one being produced while one is consumed)")
        if batch_size > model.max_batch:
            raise ValueError(f"batch {batch_size} exceeds the engine's max_batch {model.max_batch}")
        if getattr(model, "cooperative", False) and self.loader_on != "main":
            # the loader kernel on the side stream occupies compute units while the step runs: a cooperative launch
            # (CS_FLAG_COOP), whose workgroups wait for one another, could be starved into a time-out
            raise ValueError("StreamedTrainer drives a side stream: build the model with cooperative=False")
        self.torch, self.model, self.loader, self.batch, self.slots = torch, model, loader, int(batch_size), int(slots)
        self.device = model.device
        # `side_priority`: HIP stream priority of the side stream (torch numbering: larger = lower; None = default).  Measured: no
        # effect on a pass (profiles/r04_stream_prio_probe.txt) - kept for the probe, not a tuning knob
        self.side = torch.cuda.Stream(device=self.device) if side_priority is None else torch.cuda.Stream(device=self.device, priority=int(side_priority))
        self._carrying = False
        self.dist = dist
        self.world = dist.get_world_size() if dist is not None else 1
        self._dp = None
        if dist is not None:
            from .dp import DataParallel
            self._dp = DataParallel(model, dist, model.output_length)
        self.rows_seen = 0
        self.rows_dropped = 0
        self.trace = None                    # development: set to a list to collect (label, event) marks on the training stream (tools/stream_stamps.py)
        if shuffle not in ("feistel", "torch"):
            raise ValueError("shuffle must be 'feistel' or 'torch'")
        self.shuffle = shuffle
        self._perm_count = 0
        self.carry = bool(carry_remainder)
        self._carry_x = self._carry_y = None
        self._carry_n = 0

    # ---- producer (side stream)
    def _produce(self, raw, free_event):
        torch = self.torch
        mli, mlo = raw
        main = torch.cuda.current_stream(self.device)
        device_raw = [a for a in (mli, mlo) if not isinstance(a, np.ndarray) and a is not None]
        if device_raw and self.loader_on == "side":
            # raw chunks that already live in HBM were produced on the caller's stream: the loader kernel on the side
            # stream must not start before that work has finished (host arrays are copied on the side stream itself)
            self.side.wait_stream(main)
            for a in device_raw:
                a.record_stream(self.side)
        def dev(a):
            if isinstance(a, np.ndarray):
                a = torch.from_numpy(np.ascontiguousarray(a)).pin_memory()
                return a.to(self.device, non_blocking=True)
            return a
        if self.loader_on in ("main", "gaps"):
            # Host chunks: staged and copied on the side stream NOW (gated on the chunk that used this slot: at most `slots` raw
            # chunks are ever staged, however far the host runs ahead), so the copy overlaps the steps of the chunk in front.  The
            # loader kernel and the wait for the copy are issued by _consume, behind those steps (`ready` = None marks the
            # entry as raw).
            copied = None
            if any(isinstance(a, np.ndarray) for a in (mli, mlo)):
                with torch.cuda.stream(self.side):
                    if free_event is not None:
                        self.side.wait_event(free_event)
                    mli, mlo = dev(mli), dev(mlo)
                    copied = torch.cuda.Event()
                    copied.record(self.side)
                for a in (mli, mlo):
                    if a is not None:
                        a.record_stream(main)
            return (mli, mlo), copied, None
        with torch.cuda.stream(self.side):
            if free_event is not None:
                self.side.wait_event(free_event)             # the slot's previous chunk has been consumed
            x, y = self.loader.stack_raw(dev(mli), dev(mlo), extra_rows=self.batch if self._carrying else 0)
            ready = torch.cuda.Event()
            ready.record(self.side)
        x.record_stream(main)
        y.record_stream(main)
        return x, y, ready

    def _mark(self, label):
        if self.trace is not None:
            e = self.torch.cuda.Event(enable_timing=True)
            e.record()
            self.trace.append((label, e))

    # ---- consumer (main stream)
    def _consume(self, x, y, ready, lr_of_step: Callable[[int], float], gen, passes: int, step0: int):
        torch = self.torch
        main = torch.cuda.current_stream(self.device)
        self._mark("chunk")
        if ready is None:                            # loader_on == "main": x = the raw chunk (on the device), y = its copy event or None
            if y is not None:
                main.wait_event(y)
            x, y = self.loader.stack_raw(*x, extra_rows=self.batch if self._carrying else 0)
        else:
            main.wait_event(ready)
        self._mark("loaded")
        carrying = self._carrying
        n_new = x.shape[0] - (self.batch if carrying else 0)      # rows the loader wrote; the tail is headroom for carried rows
        r = self._carry_n if carrying else 0
        if r:                                        # the previous chunk's leftover rows join this chunk's permutation
            x[n_new:n_new + r].copy_(self._carry_x[:r])
            y[n_new:n_new + r].copy_(self._carry_y[:r])
            self._carry_n = 0
        n = n_new + r
        n_have = n
        self._mark("carried_in")
        if self.dist is not None:
            # Every rank streams its own timesteps, so chunks may differ in T*ncol.  The ranks must issue the SAME number of
            # all-reduces and normalise by the SAME global row count: agree on the smallest chunk and drop the surplus
            # rows of the larger ones (one 2-int collective per chunk, like a DistributedSampler's drop_last).
            t = torch.tensor([n, -n], dtype=torch.int64, device=self.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            n_max, n = int(t[0].item()), int(-t[1].item())
            self.rows_dropped += n_have - n
            if n <= 0:
                raise ValueError("a rank produced an empty chunk")
        gap = self._gap_open() if self.loader_on == "gaps" else None
        step = step0
        # one [sum sq err, sum abs err] slot per step: the engine writes them, nothing is launched to add them up
        sums = torch.zeros((passes * ((n + self.batch - 1) // self.batch), 2), dtype=torch.float32, device=self.device)
        self._sums.append(sums)
        k = 0
        n_train = (n // self.batch) * self.batch if carrying else n      # carrying: whole batches only, the rest travels on
        for _ in range(passes):
            # (on the training stream.  Round 4 drew it on a stream of its own so that its small sort kernels would run beside an
            #  earlier chunk's steps: the pass got SLOWER, 11.7 -> 13.3 ms for 4 chunks - a layer-chain launch needs every compute
            #  unit whole, and a sort workgroup sitting on one of them costs that launch a second round of workgroups)
            if self.shuffle == "torch":
                perm = torch.randperm(n_have, device=self.device, generator=gen)
            else:
                from .shuffle import chunk_seed, device_permutation
                perm = device_permutation(n_have, chunk_seed(self._seed, self._perm_count), self.device)
                self._perm_count += 1
            if n < n_have:
                perm = perm[:n]                      # a random subset of this rank's rows, as many as the smallest rank has
            self._mark("perm")
            if carrying and n_train < n:
                left = perm[n_train:n]
                if self._carry_x is None:
                    self._carry_x = torch.empty((self.batch, x.shape[1]), dtype=torch.float32, device=self.device)
                    self._carry_y = torch.empty((self.batch, y.shape[1]), dtype=torch.float32, device=self.device)
                torch.index_select(x, 0, left, out=self._carry_x[:left.numel()])
                torch.index_select(y, 0, left, out=self._carry_y[:left.numel()])
                self._carry_n = int(left.numel())
            self._mark("carried_out")
            for lo in range(0, n_train, self.batch):
                idx = perm[lo:lo + self.batch]
                lr = lr_of_step(step)
                if gap is not None and gap["next"] < len(gap["cuts"]) - 1 and self.dist is None:
                    if self.gap_at == "chain":
                        self._gap_slice(gap, main)
                        self.model.train_on_batch(x, y, lr, row_idx=idx, loss=sums[k])
                    else:
                        self.model.loss_grads(x, y, row_idx=idx, loss=sums[k])
                        self._gap_slice(gap, main)
                        self.model.apply_gradients(lr, 1.0 / (self.model.output_length * idx.numel()))
                elif self.dist is None:
                    self.model.train_on_batch(x, y, lr, row_idx=idx, loss=sums[k])
                else:
                    self.model.loss_grads(x, y, row_idx=idx, loss=sums[k])
                    self._dp.all_reduce_grads()                                   # the ONE collective of the step
                    self.model.apply_gradients(lr, 1.0 / (self.model.output_length * idx.numel() * self.world))
                step += 1
                k += 1
        if gap is not None:                          # slices the chunk's steps did not reach, then the promise the next _consume waits for
            while gap["next"] < len(gap["cuts"]) - 1:
                self._gap_slice(gap, main)
            ready2 = torch.cuda.Event()
            ready2.record(self.side)
            self._ring[0] = (gap["x"], gap["y"], ready2)
        self._mark("steps")
        done = torch.cuda.Event()
        done.record(main)
        self.rows_seen += n_train * passes
        return done, step

    # ---- loader_on == "gaps": the next chunk's loader in slices beside this chunk's steps
    def _gap_open(self):
        ring = self._ring
        if not ring or ring[0][2] is not None:
            return None
        (mli, mlo), _, _ = ring[0]
        if mlo is None:
            return None
        # (a host chunk's copy was issued on the side stream by _produce: the slices queue behind it there)
        x2, y2, run = self.loader.stack_raw_sliced(mli, mlo, extra_rows=self.batch if self._carrying else 0)
        for a in (x2, y2, mli, mlo):
            a.record_stream(self.side)
        T = int(mli.shape[0])
        ns = min(T, self.gap_slices) if self.gap_slices > 0 else T
        cuts = [T * i // ns for i in range(ns + 1)]
        return {"x": x2, "y": y2, "run": run, "cuts": cuts, "next": 0}

    def _gap_slice(self, gap, main):
        torch = self.torch
        e = torch.cuda.Event()
        e.record(main)
        self.side.wait_event(e)
        j = gap["next"]
        with torch.cuda.stream(self.side):
            gap["run"](gap["cuts"][j], gap["cuts"][j + 1])
        gap["next"] = j + 1

    def _flush_carry(self, lr_of_step, step):
        """The rows still carried when the pass ends: its one short batch (the reference's last batch of an epoch)."""
        r, self._carry_n = self._carry_n, 0
        if not r:
            return step
        torch = self.torch
        sums = torch.zeros((1, 2), dtype=torch.float32, device=self.device)
        self._sums.append(sums)
        lr = lr_of_step(step)
        if self.dist is None:
            self.model.train_on_batch(self._carry_x[:r], self._carry_y[:r], lr, loss=sums[0])
        else:                                        # every rank carries the same count (the chunks' row counts are agreed on)
            self.model.loss_grads(self._carry_x[:r], self._carry_y[:r], loss=sums[0])
            self._dp.all_reduce_grads()
            self.model.apply_gradients(lr, 1.0 / (self.model.output_length * r * self.world))
        self.rows_seen += r
        return step + 1

    def fit_chunks(self, chunks: Iterable, learning_rate=1e-3, passes_per_chunk: int = 1, seed: int = 0):
        """Train on a stream of raw chunks `(mli_raw (T, n_in, ncol), mlo_raw (T, n_out, ncol))`.
        `learning_rate`: float or callable(step).  Returns {'loss', 'mae', 'rows', 'steps'} of the pass (means over the
        rows seen; one host synchronisation at the end)."""
        torch = self.torch
        lr_of_step = learning_rate if callable(learning_rate) else (lambda s, v=float(learning_rate): v)
        gen = torch.Generator(device=self.device)
        gen.manual_seed(seed)
        self._seed, self._perm_count = int(seed), 0
        self._sums = []
        self.rows_seen = 0
        self.rows_dropped = 0
        self._carrying = self.carry and passes_per_chunk == 1
        self._carry_n = 0
        it = iter(chunks)
        ring = self._ring = []                     # [(x, y, ready)] produced, not yet consumed
        free = []                                  # `done` events of consumed chunks, oldest first
        step = step_start = self.model.iterations
        produced = 0

        def produce_one():
            nonlocal produced
            raw = next(it, None)
            if raw is None:
                return False
            ev = free.pop(0) if produced >= self.slots else None
            ring.append(self._produce(raw, ev))
            produced += 1
            return True

        for _ in range(self.slots - 1):            # fill the ring but one slot
            if not produce_one():
                break
        while True:
            produce_one()                          # next chunk goes out on the side stream ...
            have = 1 if ring else 0
            if self.dist is not None:              # the pass ends for everyone when the first rank runs out of chunks
                t = torch.tensor([have], dtype=torch.int32, device=self.device)
                self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
                have = int(t.item())
            if not have:
                break
            x, y, ready = ring.pop(0)
            done, step = self._consume(x, y, ready, lr_of_step, gen, passes_per_chunk, step)   # ... while this one trains
            free.append(done)
            del x, y
        if self._carrying:
            self._mark("flush_begin")
            step = self._flush_carry(lr_of_step, step)
            self._mark("flush_end")
        # one device-side concatenation and ONE copy (a copy per chunk was a host synchronisation per chunk at the end of the pass)
        s = torch.cat(self._sums).double().sum(dim=0).cpu().numpy() if self._sums else np.zeros(2)
        self._sums = []
        denom = max(self.rows_seen, 1) * self.model.output_length
        return {"loss": float(s[0]) / denom, "mae": float(s[1]) / denom, "rows": self.rows_seen, "steps": step - step_start}
