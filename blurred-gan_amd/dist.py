"""Data-parallel plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" == RCCL over xGMI on
ROCm; "gloo" for the CPU tests).  The reference never ran multi-GPU (wgan.py:89 "TODO: Distributed
training"); the build defines DP == the single-device step at the global batch: replicas
shard the minibatch, parameters and Adam state are replicated, and the flat gradient buffer of each
network is SUM all-reduced once per optimiser step (losses are already written for SUM semantics,
wgan.py:130,157)."""
from __future__ import annotations

import os

import torch
import torch.distributed as td


def is_initialized():
    return td.is_available() and td.is_initialized()


def rank():
    return td.get_rank() if is_initialized() else 0


def world_size():
    return td.get_world_size() if is_initialized() else 1


def local_rank():
    """Device index of this process.  One process per GPU; when a node exposes fewer devices than ranks (the
    2-ranks-on-one-card rehearsal of tests/test_dp_gpu.py and of bench.py) ranks share devices round-robin."""
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    return lr % n if n > 0 else lr


def init_from_env(backend=None):
    """Joins the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws <= 1 or is_initialized():
        return world_size()
    use_cuda = torch.cuda.is_available()
    if use_cuda:
        torch.cuda.set_device(local_rank())
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # BGAN_DIST_BACKEND=gloo: rehearsal of the multi-rank path where RCCL cannot run (several ranks on one card)
    backend = backend or os.environ.get("BGAN_DIST_BACKEND") or ("nccl" if use_cuda else "gloo")
    td.init_process_group(backend=backend)
    return world_size()


def all_reduce_sum_(flat):
    """In-place SUM all-reduce of a flat gradient buffer (no-op for a single replica)."""
    if world_size() > 1:
        if flat.is_cuda and td.get_backend() == "gloo":      # CPU test harness: stage through the host
            host = flat.cpu()
            td.all_reduce(host, op=td.ReduceOp.SUM)
            flat.copy_(host)
        else:
            td.all_reduce(flat, op=td.ReduceOp.SUM)          # RCCL ring/tree over xGMI on the GPU box
    return flat


class GradReducer:
    """SUM all-reduce of one network's flat gradient buffer, issued in buckets WHILE the backward that fills the buffer is
    still running: the engine reports each layer's slice as soon as its last contribution has been enqueued, adjacent slices
    are merged into buckets of >= ``bucket_bytes`` and each bucket goes out as an asynchronous collective (RCCL runs it on
    its own stream behind the kernels already enqueued; xGMI is point-to-point, so a few multi-MB messages per network beat
    one message per layer).  ``finish()`` reduces whatever was not reported and waits; Adam runs after it.
    With one replica every call is a no-op."""

    def __init__(self, flat, n, bucket_bytes=4 << 20):
        self.flat, self.n = flat, int(n)
        self.bucket = max(1, int(bucket_bytes) // 4)
        self.active = world_size() > 1
        self.pending = None          # (lo, hi) merged, not yet issued
        self.covered = []            # issued ranges
        self.works = []
        self.n_collectives = 0

    def _issue(self, lo, hi):
        if hi <= lo:
            return
        seg = self.flat[lo:hi]
        self.n_collectives += 1
        if seg.is_cuda and td.get_backend() == "gloo":          # CPU-side rehearsal: stage through the host, synchronous
            host = seg.cpu()
            td.all_reduce(host, op=td.ReduceOp.SUM)
            seg.copy_(host)
        else:
            self.works.append(td.all_reduce(seg, op=td.ReduceOp.SUM, async_op=True))
        self.covered.append((lo, hi))

    def ready(self, lo, hi):
        """Elements [lo, hi) of the buffer are final (their last kernel has been enqueued on the current stream).
        Anything already handed over (issued or pending) is clipped away: no element is ever reduced twice."""
        if not self.active:
            return
        lo, hi = max(0, int(lo)), min(self.n, int(hi))
        taken = sorted(self.covered + ([self.pending] if self.pending is not None else []))
        pieces, pos = [], lo
        for a, b in taken:
            if b <= pos or a >= hi:
                continue
            if a > pos:
                pieces.append((pos, a))
            pos = max(pos, b)
        if pos < hi:
            pieces.append((pos, hi))
        for a, b in pieces:
            if self.pending is not None and (a == self.pending[1] or b == self.pending[0]):   # adjacent: grow the bucket
                self.pending = (min(a, self.pending[0]), max(b, self.pending[1]))
            else:
                if self.pending is not None:
                    self._issue(*self.pending)
                self.pending = (a, b)
            if self.pending[1] - self.pending[0] >= self.bucket:
                self._issue(*self.pending)
                self.pending = None

    def finish(self):
        if not self.active:
            return
        if self.pending is not None:
            self._issue(*self.pending)
            self.pending = None
        pos = 0                                                   # whatever no layer reported (padding, unused variables)
        for lo, hi in sorted(self.covered):
            if lo > pos:
                self._issue(pos, lo)
            pos = max(pos, hi)
        if pos < self.n:
            self._issue(pos, self.n)
        for w in self.works:
            w.wait()
        self.works = []


def max_over_ranks(value: float) -> float:
    """MAX of a host scalar over the replicas (bench timing)."""
    if world_size() <= 1:
        return float(value)
    dev = "cuda" if (td.get_backend() == "nccl") else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    td.all_reduce(t, op=td.ReduceOp.MAX)
    return float(t.item())


def broadcast_(flat, src=0):
    if world_size() > 1:
        td.broadcast(flat, src=src)
    return flat


def barrier():
    if world_size() > 1:
        td.barrier()


def shard(batch, r=None, n=None):
    """The slice of a global batch this replica trains on (equal shards)."""
    r = rank() if r is None else r
    n = world_size() if n is None else n
    per = batch.shape[0] // n
    return batch[r * per:(r + 1) * per]
