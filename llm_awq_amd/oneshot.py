"""One-shot all-reduce for the latency-class messages of tensor-parallel decode (csrc/awq_oneshot.hip, include/awq_cdna4.h).

Protocol (identical in the HIP kernel and in `HostMailbox`, the numpy restatement the CPU tests drive over POSIX shared memory):
every rank owns an exchange buffer  data[2 halves][W slots][max_bytes] + flags[2][W];  in round e a rank stores its partial into
slot `rank` of half e & 1 of EVERY rank's buffer, then raises flag (e & 1, rank) = e in every buffer, waits for the W flags of its
own buffer and reduces the W slots in rank order with fp32 accumulation and one rounding.  Half e & 1 is reused in round e + 2,
which a rank can only enter after it saw every peer's flag of round e + 1 -- raised after that peer finished reading round e.

`OneShotAllReduce` wires the GPU kernel to a torch.distributed group: buffers are allocated fine-grained, exported with hipIpc
handles (all_gather_object over the group) and opened on every peer.  It is OPT-IN (`AWQ_ONESHOT=1` or an explicit object handed to
TPWQLinear): the round-end scaling bench runs on hardware this repository's author could not reach, and an RCCL all-reduce is the
path that is known to work there; messages above `max_bytes` always go to RCCL (bandwidth-bound: ring / direct RS+AG territory).
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

    def all_reduce(self, x: torch.Tensor, timeout_s: float = 20.0) -> torch.Tensor:
        """x: fp16 / bf16 CPU tensor; returns the sum over ranks (fp32 accumulation in rank order, one rounding)."""
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


class OneShotAllReduce:
    """GPU side: one exchange buffer per rank, peers mapped through hipIpc.  `__call__(t)` returns the reduced tensor (new
    storage) for messages up to max_bytes and falls back to `dist.all_reduce` (in place) above."""

    def __init__(self, group=None, max_bytes: int = 64 * 1024, device=None, device_epoch: bool = True):
        import torch.distributed as dist
        from . import _capi
        self.dist, self.group, self.L = dist, group, _capi.lib()
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        assert self.world <= MAX_WORLD and max_bytes % 16 == 0
        self.max_bytes = max_bytes
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.round = 0
        # device_epoch: the round counter lives in the exchange buffer and the kernel advances it (round argument 0) -- safe inside a
        # captured and replayed hipGraph; False: the host passes 1, 2, 3, ... (eager launches only)
        self.device_epoch = device_epoch
        with torch.cuda.device(self.device):
            buf = ctypes.c_void_p()
            _capi.check(self.L.awq_oneshot_alloc(ctypes.byref(buf), self.world, max_bytes))
            self.local = buf
            handle = ctypes.create_string_buffer(64)
            _capi.check(self.L.awq_oneshot_ipc_export(buf, handle))
            handles = [None] * self.world
            dist.all_gather_object(handles, bytes(handle.raw), group=group)
            self.opened = []
            ptrs = (ctypes.c_void_p * MAX_WORLD)()
            for q in range(self.world):
                if q == self.rank:
                    ptrs[q] = buf.value
                else:
                    p = ctypes.c_void_p()
                    _capi.check(self.L.awq_oneshot_ipc_open(ctypes.create_string_buffer(handles[q], 64), ctypes.byref(p)))
                    self.opened.append(p)
                    ptrs[q] = p.value
            self.ptrs = ptrs
            self.status = torch.zeros(1, dtype=torch.int32, device=self.device)
        dist.barrier(group=group)  # every buffer is zeroed and mapped before the first round

    def __call__(self, t: torch.Tensor) -> torch.Tensor:
        from . import _capi
        nbytes = t.numel() * t.element_size()
        if t.dtype not in (torch.float16, torch.bfloat16) or nbytes > self.max_bytes or t.numel() % 8 or not t.is_contiguous():
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
    return os.environ.get("AWQ_ONESHOT", "0") == "1"
