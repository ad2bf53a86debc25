"""One-shot all-reduce for the latency-class messages of tensor-parallel decode (csrc/awq_oneshot.hip, include/awq_cdna4.h).

Protocol (identical in the HIP kernel and in `HostMailbox`, the numpy restatement the CPU tests drive over POSIX shared memory):
every rank owns an exchange buffer  data[2 halves][W slots][max_bytes] + flags[2][W];  in round e a rank stores its partial into
slot `rank` of half e & 1 of EVERY rank's buffer, then raises flag (e & 1, rank) = e in every buffer, waits for the W flags of its
own buffer and reduces the W slots in rank order with fp32 accumulation and one rounding.  Half e & 1 is reused in round e + 2,
which a rank can only enter after it saw every peer's flag of round e + 1 -- raised after that peer finished reading round e.

`OneShotAllReduce` wires the GPU kernel to a torch.distributed group: buffers are allocated fine-grained, exported with hipIpc
handles (all_gather_object over the group) and opened on every peer.  It is the DEFAULT reducer of the tensor-parallel paths for
messages up to `max_bytes` when world > 1 (`AWQ_ONESHOT=0` is the escape hatch; `make_reducer` falls back to RCCL when the exchange
buffers cannot be set up -- no fine-grained memory, no hipIpc -- and every rank agrees on that through one all-reduce); messages
above `max_bytes` always go to RCCL (bandwidth-bound: ring / direct RS+AG territory).  A round whose peer flag does not arrive within
the spin bound poisons its output with NaNs and sets a sticky status word: `check()` (after a synchronize) raises.
"""
from __future__ import annotations

import ctypes
import os
import time

import numpy as np
import torch

MAX_WORLD = 8


class HostMailbox:
    """The protocol on host memory: `bufs[q]` is rank q's exchange buffer as a uint8 numpy array (shared between the processes)."""

    def __init__(self, bufs, rank: int, world: int, max_bytes: int):
        assert 1 <= world <= MAX_WORLD and max_bytes % 16 == 0
        self.bufs, self.rank, self.world, self.max_bytes = bufs, rank, world, max_bytes
        self.round = 0

    @staticmethod
    def buffer_bytes(world: int, max_bytes: int) -> int:
        return 2 * world * max_bytes + 256

    def _data(self, q, half, slot, nbytes):
        off = (half * self.world + slot) * self.max_bytes
        return self.bufs[q][off: off + nbytes]

    def _flags(self, q):
        off = 2 * self.world * self.max_bytes
        return self.bufs[q][off: off + 64].view(np.uint32)

    def all_reduce(self, x: torch.Tensor, timeout_s: float = 20.0, out_dtype=None) -> torch.Tensor:
        """x: fp16 / bf16 CPU tensor; returns the sum over ranks (fp32 accumulation in rank order, one rounding).
        x float32 + out_dtype = T: the fp32 form (awq_oneshot_allreduce_f32): unrounded partials in, T(sum) out."""
        if x.dtype == torch.float32:
            return self._all_reduce_f32(x, timeout_s, out_dtype or torch.float32)
        self.round += 1
        e, half = self.round, self.round & 1
        raw = x.contiguous().view(torch.int16).numpy().view(np.uint8).reshape(-1)
        assert raw.size <= self.max_bytes
        for q in range(self.world):                      # (1) my partial into slot `rank` of every buffer
            self._data(q, half, self.rank, raw.size)[:] = raw
        for q in range(self.world):                      # (2) then the flags (numpy stores of a process are program ordered)
            self._flags(q)[half * self.world + self.rank] = e
        mine = self._flags(self.rank)
        t_end = time.time() + timeout_s
        while not all(int(mine[half * self.world + q]) == e for q in range(self.world)):   # (3)
            if time.time() > t_end:
                raise TimeoutError(f"rank {self.rank}: round {e} flags {mine[half * self.world: half * self.world + self.world]}")
            time.sleep(0.0005)
        acc = torch.zeros(x.numel(), dtype=torch.float32)
        for q in range(self.world):                      # (4) rank order
            part = torch.from_numpy(self._data(self.rank, half, q, raw.size).copy().view(np.int16)).view(x.dtype)
            acc += part.float()
        return acc.to(x.dtype).reshape(x.shape)

    def _all_reduce_f32(self, x, timeout_s, out_dtype):
        self.round += 1
        e, half = self.round, self.round & 1
        raw = x.contiguous().numpy().view(np.uint8).reshape(-1)
        assert raw.size <= self.max_bytes
        for q in range(self.world):
            self._data(q, half, self.rank, raw.size)[:] = raw
        for q in range(self.world):
            self._flags(q)[half * self.world + self.rank] = e
        mine = self._flags(self.rank)
        t_end = time.time() + timeout_s
        while not all(int(mine[half * self.world + q]) == e for q in range(self.world)):
            if time.time() > t_end:
                raise TimeoutError(f"rank {self.rank}: round {e} flags {mine[half * self.world: half * self.world + self.world]}")
            time.sleep(0.0005)
        acc = torch.zeros(x.numel(), dtype=torch.float32)
        for q in range(self.world):                      # rank order, fp32
            acc += torch.from_numpy(self._data(self.rank, half, q, raw.size).copy().view(np.float32))
        return acc.to(out_dtype).reshape(x.shape)


class OneShotAllReduce:
    """GPU side: one exchange buffer per rank, peers mapped through hipIpc.  `__call__(t)` returns the reduced tensor (new
    storage) for messages up to max_bytes and falls back to `dist.all_reduce` (in place) above."""

    def __init__(self, group=None, max_bytes: int = 64 * 1024, device=None, device_epoch: bool = True):
        import torch.distributed as dist
        from . import _capi
        self.dist, self.group, self.L = dist, group, _capi.lib()
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        spin = os.environ.get("AWQ_ONESHOT_SPIN_LIMIT")  # polls (~1 us each) before a round is declared lost; default 40 M (library)
        if spin:
            _capi.check(self.L.awq_oneshot_set_spin_limit(int(spin)))
        assert self.world <= MAX_WORLD and max_bytes % 16 == 0
        self.max_bytes = max_bytes
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.round = 0
        # device_epoch: the round counter lives in the exchange buffer and the kernel advances it (round argument 0) -- safe inside a
        # captured and replayed hipGraph; False: the host passes 1, 2, 3, ... (eager launches only)
        self.device_epoch = device_epoch
        # every step that can fail on one rank only (no fine-grained memory, no hipIpc) is followed by an exchange of the outcomes, so
        # that ALL ranks raise together and nobody is left waiting inside a collective
        self.local, self.opened = None, []
        with torch.cuda.device(self.device):
            raw, err = None, None
            try:
                buf = ctypes.c_void_p()
                _capi.check(self.L.awq_oneshot_alloc(ctypes.byref(buf), self.world, max_bytes))
                self.local = buf
                handle = ctypes.create_string_buffer(64)
                _capi.check(self.L.awq_oneshot_ipc_export(buf, handle))
                raw = bytes(handle.raw)
            except Exception as e:  # noqa: BLE001
                err = f"{type(e).__name__}: {e}"
            handles = [None] * self.world
            dist.all_gather_object(handles, raw, group=group)
            if any(h is None for h in handles):
                self.close()
                raise RuntimeError(f"one-shot all-reduce: exchange buffer unavailable on rank(s) {[q for q, h in enumerate(handles) if h is None]}"
                                   + (f" (here: {err})" if err else ""))
            ptrs = (ctypes.c_void_p * MAX_WORLD)()
            try:
                for q in range(self.world):
                    if q == self.rank:
                        ptrs[q] = self.local.value
                    else:
                        p = ctypes.c_void_p()
                        _capi.check(self.L.awq_oneshot_ipc_open(ctypes.create_string_buffer(handles[q], 64), ctypes.byref(p)))
                        self.opened.append(p)
                        ptrs[q] = p.value
            except Exception as e:  # noqa: BLE001
                err = f"{type(e).__name__}: {e}"
            outcomes = [None] * self.world
            dist.all_gather_object(outcomes, err, group=group)
            if any(o is not None for o in outcomes):
                self.close()
                raise RuntimeError(f"one-shot all-reduce: peer buffers could not be mapped: {[o for o in outcomes if o]}")
            self.ptrs = ptrs
            self.status = torch.zeros(1, dtype=torch.int32, device=self.device)
        dist.barrier(group=group)  # every buffer is zeroed and mapped before the first round

    def serves(self, t: torch.Tensor) -> bool:
        """whether this reducer exchanges `t` itself (else the group's all-reduce does): size class, dtype, and the DEVICE its buffers
        and launch stream live on (a tensor elsewhere has no ordering against that stream)"""
        return (t.is_cuda and t.device == self.device and t.dtype in (torch.float16, torch.bfloat16, torch.float32) and t.is_contiguous()
                and t.numel() % 8 == 0 and 0 < t.numel() * t.element_size() <= self.max_bytes and self.local is not None)

    def reduce_f32(self, y32: torch.Tensor, out_dtype, bias=None) -> torch.Tensor:
        """fp32 partials of a row split -> T(sum over ranks) (+ bias in T).  Latency-class messages go through the exchange buffers
        (sum, rounding and bias inside the one kernel); larger ones are summed by the group's all-reduce ON THE FLOAT TENSOR and rounded
        once by awq_round_bias_f32."""
        from . import _capi, ops
        if not self.serves(y32) or (bias is not None and (y32.numel() % bias.numel() or bias.numel() % 8)):
            self.dist.all_reduce(y32, group=self.group)
            return ops.round_bias_f32(y32, out_dtype, bias)
        self.round += 1
        out = torch.empty(y32.shape, dtype=out_dtype, device=y32.device)
        with torch.cuda.device(self.device):
            _capi.check(self.L.awq_oneshot_allreduce_f32(self.ptrs, y32.data_ptr(), bias.data_ptr() if bias is not None else None,
                                                         bias.numel() if bias is not None else 0, out.data_ptr(), y32.numel(),
                                                         0 if out_dtype == torch.float16 else 1, self.rank, self.world,
                                                         0 if self.device_epoch else self.round, self.max_bytes, self.status.data_ptr(),
                                                         torch.cuda.current_stream(self.device).cuda_stream))
        return out

    def __call__(self, t: torch.Tensor) -> torch.Tensor:
        from . import _capi
        if t.dtype == torch.float32 or not self.serves(t):
            self.dist.all_reduce(t, group=self.group)
            return t
        self.round += 1
        out = torch.empty_like(t)
        with torch.cuda.device(self.device):
            _capi.check(self.L.awq_oneshot_allreduce(self.ptrs, t.data_ptr(), out.data_ptr(), t.numel(), 0 if t.dtype == torch.float16 else 1,
                                                     self.rank, self.world, 0 if self.device_epoch else self.round, self.max_bytes, self.status.data_ptr(),
                                                     torch.cuda.current_stream(self.device).cuda_stream))
        return out

    def check(self):
        """raise if a round timed out waiting for a peer (call after a synchronize)"""
        if int(self.status.item()) != 0:
            raise RuntimeError("one-shot all-reduce: a peer's flag did not arrive within the spin bound")

    def close(self):
        for p in self.opened:
            self.L.awq_oneshot_ipc_close(p)
        self.opened = []
        if self.local is not None:
            self.L.awq_oneshot_free(self.local)
            self.local = None


def enabled_by_env() -> bool:
    return os.environ.get("AWQ_ONESHOT", "1") != "0"


def make_reducer(dist, group=None, max_bytes: int = 64 * 1024, device=None):
    """The default small-message reducer of a tensor-parallel group: a OneShotAllReduce if EVERY rank could build one, else None
    (= torch.distributed.all_reduce).  Collective: every rank of the group must call it."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    world = dist.get_world_size(group)
    if world <= 1 or world > MAX_WORLD or not enabled_by_env() or not torch.cuda.is_available():
        return None
    import sys
    try:
        red = OneShotAllReduce(group, max_bytes, device)  # (raises on EVERY rank or on none: see __init__)
    except Exception as e:  # noqa: BLE001 -- no fine-grained memory / hipIpc on this box: RCCL serves the messages
        print(f"[oneshot] rank {dist.get_rank(group)}: {e}; using RCCL for the small messages", file=sys.stderr)
        return None
    why = validate_reducer(red, dist, group)
    if why is not None:  # (agreed by all ranks: every rank drops the reducer or none does)
        print(f"[oneshot] rank {dist.get_rank(group)}: self-check failed ({why}); using RCCL for the small messages", file=sys.stderr)
        red.close()
        return None
    return red


def validate_reducer(red, dist, group=None, rounds: int = 6):
    """A few rounds of the one-shot reducer against dist.all_reduce on rank-dependent data (both message sizes the decode paths send and
    a ragged one), then ONE collective agreement: returns None if every rank saw every round match (sums of small integers: exact in
    T whatever the order) and no round timed out, else a reason string -- on EVERY rank.  The peer-mapped path has only ever run
    between processes on one GPU in the test suite; a fabric where peer stores are not visible to a polling kernel shows up here as
    a timeout / mismatch and costs the fallback, not the run."""
    why = None
    dev, rank = red.device, red.rank
    cdev = torch.device("cpu") if dist.get_backend(group) == "gloo" else dev  # (the reference sum travels over the group's own backend)
    for r in range(rounds):
        n = (4096, 8192, 8 * (1 + r))[r % 3]
        if 2 * n > red.max_bytes:
            n = red.max_bytes // 2
        dt = torch.bfloat16 if r & 1 else torch.float16
        t = ((torch.arange(n, device=dev, dtype=torch.float32) % 7) + rank + r).to(dt)
        ref = t.float().to(cdev)
        dist.all_reduce(ref, group=group)  # (outside the try: every rank issues the same sequence of collectives whatever its one-shot rounds do)
        ref = ref.to(dev).to(dt)
        if why is not None:
            continue
        try:
            got = red(t.clone())
            got32 = red.reduce_f32(t.float(), dt) if 4 * n <= red.max_bytes else ref  # the fp32-partial form the row splits send
            torch.cuda.synchronize(dev)
            if int(red.status.item()) != 0:
                why = f"round {r}: a peer flag timed out"
            elif not torch.equal(got, ref):
                why = f"round {r}: {int((got != ref).sum().item())} of {n} elements differ from the group's all-reduce"
            elif not torch.equal(got32, ref):
                why = f"round {r}: fp32 form: {int((got32 != ref).sum().item())} of {n} elements differ from the group's all-reduce"
        except Exception as e:  # noqa: BLE001
            why = f"round {r}: {type(e).__name__}: {e}"
    bad = torch.tensor([0 if why is None else 1], dtype=torch.int32,
                       device="cpu" if dist.get_backend(group) == "gloo" else red.device)
    dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=group)
    if int(bad.item()) != 0:
        return why or "another rank's self-check failed"
    return None
