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

import os


class RcclComm:
    """The native collective of the C ABI (`cs_dp_*`, include/climsim_hip.h): an RCCL communicator of the engine's own,
    bootstrapped through the torch.distributed rendezvous (rank 0's unique id is broadcast), whose all-reduce is
    issued on the stream the step's kernels run on - ordered with them without events or a second stream."""

    def __init__(self, dist, device):
        import ctypes as C
        import numpy as np
        import torch
        from . import _lib
        self._lib, self._check, self._C, self._torch = _lib.load(), _lib.check, C, torch
        self.device = device
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        self._path = path.encode() if os.path.exists(path) else None          # the RCCL torch already loaded
        # Every rank takes the same decision at every step of the setup: a rank that raised alone would leave the others
        # waiting inside a collective (ncclCommInitRank included).  So (1) everything local - loading the library, binding
        # RCCL, creating an id (every rank makes one: it proves dlopen + ncclGetUniqueId work here) - runs inside a try
        # and its outcome is agreed on with a MIN all-reduce; (2) rank 0's id is broadcast; (3) the collective init runs
        # on every rank or on none; (4) its outcome is agreed on again.
        self._c = C.c_void_p()
        idbuf = C.create_string_buffer(128)
        local_ok, err = 1, ""
        try:
            if self._lib.cs_dp_unique_id(self._path, idbuf) != 0:
                local_ok, err = 0, self._lib.cs_last_error().decode()
        except Exception as e:  # noqa: BLE001
            local_ok, err = 0, f"{type(e).__name__}: {e}"
        agreed = torch.tensor([local_ok], dtype=torch.int32, device=device)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        if int(agreed.item()) == 0:
            raise _lib.EngineError("RCCL is not usable on some rank" + (": " + err if err else ""))
        t = torch.from_numpy(np.frombuffer(idbuf.raw, dtype=np.uint8).copy()).to(device)
        dist.broadcast(t, src=0)
        ident = C.create_string_buffer(t.cpu().numpy().tobytes(), 128)
        with torch.cuda.device(device):
            rc = self._lib.cs_dp_init(C.byref(self._c), self._path, ident, dist.get_world_size(), dist.get_rank(),
                                      device.index if device.index is not None else torch.cuda.current_device())
        err = self._lib.cs_last_error().decode() if rc != 0 else ""
        agreed = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=device)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        if int(agreed.item()) == 0:
            self.close()
            raise _lib.EngineError("RCCL communicator setup failed on some rank" + (": " + err if err else ""))

    def all_reduce(self, tensor, payload: str = "fp32"):
        st = self._C.c_void_p(self._torch.cuda.current_stream(tensor.device).cuda_stream)
        fn = self._lib.cs_dp_allreduce_bf16 if payload == "bf16" else self._lib.cs_dp_allreduce
        self._check(fn(self._c, self._C.c_void_p(tensor.data_ptr()), tensor.numel(), st))

    def info(self):
        """(nranks, rank) as RCCL reports them for this communicator (ncclCommCount / ncclCommUserRank)."""
        n, r = self._C.c_int(-1), self._C.c_int(-1)
        self._check(self._lib.cs_dp_comm_info(self._c, self._C.byref(n), self._C.byref(r)))
        return int(n.value), int(r.value)

    def close(self):
        if getattr(self, "_c", None) is not None and self._c.value:
            self._lib.cs_dp_destroy(self._c)
            self._c = self._C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _DevicePointer:
    """A raw device allocation of the library as an object torch can wrap without copying (`torch.as_tensor`)."""

    def __init__(self, ptr: int, n_floats: int):
        self.__cuda_array_interface__ = {"shape": (n_floats,), "typestr": "<f4", "data": (ptr, False), "version": 2, "strides": None}


class IpcComm:
    """The one-shot all-reduce of the C ABI (`cs_dp_ipc_*`): every rank's gradient buffer is an exchange buffer the other
    ranks of the node map through HIP IPC; one kernel per rank pulls + sums its slice and pushes the sum back to everyone.
    The ENGINE'S gradient buffer is rebound to the exchange buffer, so the collective works in place.  An option
    (`DataParallel(collective="oneshot")`, CS_DP_COLLECTIVE=oneshot): it has run with two processes on one device only."""

    def __init__(self, dist, device, engine, timeout_ms: float | None = None):
        """`timeout_ms`: wall-clock bound of every wait inside the collective (default 30 s, CS_DP_IPC_TIMEOUT_MS): ranks may be
        skewed by a checkpoint, a validation pass or a lazy first-step build on one of them; align them with a barrier before
        the first step (the constructor ends with one).  A wait that runs out is counted and fails the NEXT call."""
        import ctypes as C
        import torch
        from . import _lib
        self._lib, self._check, self._C, self._torch = _lib.load(), _lib.check, C, torch
        world, rank = dist.get_world_size(), dist.get_rank()
        n = int(engine.gradient_tensor().numel())
        n4 = (n + 3) // 4 * 4
        self._c = C.c_void_p()
        self.n, self._engine, self._n_grad = n4, engine, n
        # as in RcclComm: every local step's outcome is agreed on, so that no rank raises alone
        ok, err = 1, ""
        rec = C.create_string_buffer(128)
        with torch.cuda.device(device):
            if self._lib.cs_dp_ipc_create(C.byref(self._c), world, rank, device.index if device.index is not None else torch.cuda.current_device(), n4) != 0 \
                    or self._lib.cs_dp_ipc_export(self._c, rec) != 0:
                ok, err = 0, self._lib.cs_last_error().decode()
        self._agree(dist, device, ok, "exchange buffer setup failed", err)
        if timeout_ms is not None:
            self._check(self._lib.cs_dp_ipc_set_timeout_ms(self._c, float(timeout_ms)))
        import numpy as np
        xdev = device if dist.get_backend() == "nccl" else "cpu"          # the handle records travel through the rendezvous' own medium
        mine = torch.from_numpy(np.frombuffer(rec.raw, dtype=np.uint8).copy()).to(xdev)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        allrec = C.create_string_buffer(b"".join(p.cpu().numpy().tobytes() for p in parts), 128 * world)
        with torch.cuda.device(device):
            rc = self._lib.cs_dp_ipc_connect(self._c, allrec)
        self._agree(dist, device, 1 if rc == 0 else 0, "opening the peers' buffers failed", self._lib.cs_last_error().decode() if rc else "")
        ptr, nn = C.c_void_p(), C.c_int64()
        self._check(self._lib.cs_dp_ipc_buffer(self._c, C.byref(ptr), C.byref(nn)))
        self.tensor = torch.as_tensor(_DevicePointer(ptr.value, n4), device=device)[:n]      # a view of the library's allocation
        engine.bind_gradient_tensor(self.tensor)
        dist.barrier()                                            # nobody starts a step before every rank has opened every buffer

    def _agree(self, dist, device, ok, what, err):
        t = self._torch.tensor([ok], dtype=self._torch.int32, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if int(t.item()) == 0:
            self.close()
            raise self._lib_error(f"one-shot all-reduce: {what} on some rank" + (": " + err if err else ""))

    @staticmethod
    def _lib_error(msg):
        from ._lib import EngineError
        return EngineError(msg)

    def all_reduce(self, tensor, payload: str = "fp32"):
        if payload != "fp32":
            raise ValueError("the one-shot all-reduce sums float32")
        st = self._C.c_void_p(self._torch.cuda.current_stream(tensor.device).cuda_stream)
        self._check(self._lib.cs_dp_ipc_allreduce(self._c, self.n, st))

    @property
    def timeouts(self) -> int:
        return int(self._lib.cs_dp_ipc_timeouts(self._c))

    def close(self):
        if getattr(self, "_c", None) is not None and self._c.value:
            try:
                if getattr(self, "tensor", None) is not None:
                    # the engine must not keep pointing into an allocation that is about to be freed: give it a torch tensor again
                    # (an engine that was closed first raises here - the exchange buffer is freed all the same)
                    self._torch.cuda.synchronize()
                    dev, self.tensor = self.tensor.device, None
                    self._engine.bind_gradient_tensor(self._torch.zeros(self._n_grad, dtype=self._torch.float32, device=dev))
            finally:
                self._lib.cs_dp_ipc_destroy(self._c)
                self._c = self._C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_of_batch(perm, step: int, global_batch: int, rank: int, world: int):
    """Rows of global batch `step` owned by `rank`: a strided view of the epoch permutation."""
    sl = perm[step * global_batch + rank:(step + 1) * global_batch:world]
    return sl.contiguous() if hasattr(sl, "contiguous") and world > 1 else sl


class DataParallel:
    def __init__(self, engine, dist=None, output_length: int = 128, grad_payload: str | None = None, collective: str | None = None,
                 oneshot_timeout_ms: float | None = None):
        """`grad_payload`: "fp32" (default; what DDP in the reference sends) or "bf16" - the gradient sums cross the links
        as bf16 (half the bytes: 2.39 MB for the 5x512 MLP, 26.4 MB for the CNN) and are widened back before the optimiser,
        whose moments and master weights stay float32.  Every rank receives the same reduced buffer either way, so the
        replicas stay bit-identical to each other; against fp32 sums the update carries bf16 rounding of the gradient.
        Environment default: CS_DP_PAYLOAD.
        `collective`: "rccl" (default) or "oneshot" - the one-kernel all-reduce over peer-mapped buffers (IpcComm; one node,
        at most 8 ranks, fp32 payload; an option until a measured scaling curve says otherwise).  Environment: CS_DP_COLLECTIVE.
        `oneshot_timeout_ms`: wall-clock bound of the one-shot collective's waits (default 30 s; IpcComm)."""
        self.engine, self.dist = engine, dist
        self.payload = grad_payload or os.environ.get("CS_DP_PAYLOAD", "fp32")
        if self.payload not in ("fp32", "bf16"):
            raise ValueError(f"grad_payload must be 'fp32' or 'bf16', not {self.payload!r}")
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self.output_length = output_length
        self.grad = engine.gradient_tensor()
        # GPU runs take the engine's own RCCL communicator (collective on the compute stream); torch.distributed's
        # all_reduce remains for CPU tensors (gloo tests), on request (CS_DP_NATIVE=0) and if the native setup fails
        self.native = None
        self.collective = collective or os.environ.get("CS_DP_COLLECTIVE", "rccl")
        if self.collective not in ("rccl", "oneshot"):
            raise ValueError(f"collective must be 'rccl' or 'oneshot', not {self.collective!r}")
        if self.collective == "oneshot" and dist is not None and getattr(self.grad, "is_cuda", False):
            if self.payload != "fp32":
                raise ValueError("collective='oneshot' sums float32 (grad_payload='fp32')")
            self.native = IpcComm(dist, self.grad.device, engine, timeout_ms=oneshot_timeout_ms)     # raises on EVERY rank or on none
            self.grad = engine.gradient_tensor()                      # now the exchange buffer
        elif dist is not None and getattr(self.grad, "is_cuda", False) and os.environ.get("CS_DP_NATIVE", "1") != "0":
            from ._lib import EngineError
            try:
                self.native = RcclComm(dist, self.grad.device)
            except EngineError as e:
                # RcclComm raises EngineError on EVERY rank or on none (its setup is agreed on across the group), so the
                # ranks cannot end up mixing native and torch collectives on the gradient; anything else propagates
                import warnings
                warnings.warn(f"native RCCL communicator unavailable ({e}); using torch.distributed.all_reduce")

    def close(self):
        """Destroy the native communicator (before torch.distributed is torn down and long before interpreter exit)."""
        if self.native is not None:
            self.native.close()
            self.native = None

    def all_reduce_grads(self):
        """The ONE collective of the step: SUM of the flat gradient buffer over all ranks, in place."""
        if self.native is not None:
            self.native.all_reduce(self.grad, self.payload)
        elif self.dist is not None:
            if self.payload == "bf16":
                import torch
                half = self.grad.to(torch.bfloat16)               # round-to-nearest-even, as cs_dp_allreduce_bf16 packs
                self.dist.all_reduce(half)
                self.grad.copy_(half)
            else:
                self.dist.all_reduce(self.grad)

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
        self.all_reduce_grads()                                   # the ONE collective of the step
        self.engine.apply_gradients(lr, 1.0 / (self.output_length * global_batch))
        return out
