"""Data-parallel plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" == RCCL over xGMI on
ROCm; "gloo" for the CPU tests).  The reference never ran multi-GPU (wgan.py:89 "TODO: Distributed
training"); the build defines DP == the single-device step at the global batch: replicas
shard the minibatch, parameters and Adam state are replicated, and the flat gradient buffer of each
network is SUM all-reduced once per optimiser step (losses are already written for SUM semantics,
wgan.py:130,157)."""
from __future__ import annotations

import datetime as _dt
import os

import torch
import torch.distributed as td


def is_initialized():
    return td.is_available() and td.is_initialized()


def rank():
    return td.get_rank() if is_initialized() else 0


def world_size():
    return td.get_world_size() if is_initialized() else 1


def collectives_active():
    """True when gradient / BatchNorm-statistics exchange has to run: more than one replica, or a one-rank group with
    BGAN_DP_FORCE_COLLECTIVES=1 (the RCCL rehearsal of tests/test_dp_gpu.py on a one-GPU box: every collective is issued
    and waited for exactly as with N ranks, its result is the identity)."""
    return world_size() > 1 or (is_initialized() and os.environ.get("BGAN_DP_FORCE_COLLECTIVES") == "1")


def local_rank():
    """Device index of this process (one process per GPU).

    * LOCAL_RANK < visible devices: that device (torchrun on a whole node).
    * exactly ONE visible device and a per-rank mask in the environment (ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES /
      CUDA_VISIBLE_DEVICES, e.g. SLURM --gpus-per-task=1): device 0 -- every rank sees its own card as the only one.
    * anything else with more local ranks than devices is an error: two RCCL ranks on one card hang inside the first
      collective instead of failing.  Sharing a card round-robin is a REHEARSAL mode and needs the gloo backend
      (BGAN_DIST_BACKEND=gloo, with or without BGAN_DIST_SHARE_DEVICES=1); BGAN_DIST_SHARE_DEVICES=1 with the RCCL backend is
      refused for the same reason."""
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n > 0 and lr >= n:
        if os.environ.get("BGAN_DIST_BACKEND") == "gloo":
            return lr % n
        masked = any(os.environ.get(v) for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
        if n == 1 and masked and os.environ.get("BGAN_DIST_SHARE_DEVICES") != "1":
            return 0
        raise RuntimeError(f"LOCAL_RANK={lr} but only {n} GPU(s) are visible: launch one process per GPU "
                           f"(--nproc-per-node <= {n}), or mask one device per rank (ROCR_VISIBLE_DEVICES); to rehearse several "
                           "ranks on one card set BGAN_DIST_BACKEND=gloo (RCCL ranks sharing a card hang in their first collective)")
    return lr


def init_from_env(backend=None):
    """Joins the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    forced = os.environ.get("BGAN_DP_FORCE_COLLECTIVES") == "1" and "RANK" in os.environ   # one-rank rehearsal of the collective path
    if (ws <= 1 and not forced) or is_initialized():
        return world_size()
    use_cuda = torch.cuda.is_available()
    if use_cuda:
        torch.cuda.set_device(local_rank())
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # BGAN_DIST_BACKEND=gloo: rehearsal of the multi-rank path where RCCL cannot run (several ranks on one card)
    backend = backend or os.environ.get("BGAN_DIST_BACKEND") or ("nccl" if use_cuda else "gloo")
    # The rendezvous store comes FIRST and the device check runs on it BEFORE the process group exists: with device_id= the RCCL
    # communicator is created inside init_process_group, so two ranks that ended up on one card would hang in there and a check
    # placed after it would never run.  torch.distributed.rendezvous is the public env:// handler (it knows whether
    # torchrun's agent already hosts the store); the same store is then handed to init_process_group.
    rank = int(os.environ.get("RANK", "0"))
    store = None
    try:
        store, rank, ws = next(iter(td.rendezvous("env://", rank, ws)))
        store.set_timeout(_dt.timedelta(seconds=int(os.environ.get("BGAN_DIST_TIMEOUT_S", "600"))))
    except Exception:                   # an exotic launcher: fall back to init's own rendezvous, check afterwards
        store = None
    if store is not None:
        _check_one_device_per_rank(backend, store, rank, ws)
    kw = {}
    if use_cuda and backend == "nccl":
        # bind the communicator to this rank's card up front: no "guessing device ID based on global rank" (and no hang when
        # the rank -> GPU mapping is not the identity)
        kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
    if store is not None:
        td.init_process_group(backend=backend, store=store, rank=rank, world_size=ws, **kw)
    else:
        td.init_process_group(backend=backend, **kw)
        _check_one_device_per_rank(backend)
    return world_size()


def _device_identity():
    """(host, physical card) of this process's current device: uuid and PCI domain:bus:device as the runtime reports them, or ""
    when it reports neither (an identity that cannot be told apart must not be mistaken for a shared card)."""
    import socket
    pr = torch.cuda.get_device_properties(torch.cuda.current_device())
    uuid = str(getattr(pr, "uuid", "") or "")
    if not uuid.strip("0-"):
        uuid = ""
    pci_parts = [getattr(pr, a, None) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")]
    pci = "" if any(v is None for v in pci_parts) else ":".join(str(v) for v in pci_parts)
    if not uuid and not pci:
        return ""
    return f"{socket.gethostname()}|{uuid}|{pci}"


def _distinct_devices(ids):
    """ids: one identity string per rank ("" = unknown, not compared).  Raises when two ranks resolved to the same physical card."""
    seen = {}
    for r, ident in enumerate(ids):
        if not ident:
            continue
        if ident in seen:
            raise RuntimeError(f"ranks {seen[ident]} and {r} both run on device {ident}: RCCL ranks sharing a card hang in their "
                               "first collective.  Launch one process per GPU, or give every rank its own device mask "
                               "(ROCR_VISIBLE_DEVICES); to rehearse several ranks on one card use BGAN_DIST_BACKEND=gloo")
        seen[ident] = r


def _check_one_device_per_rank(backend, store=None, me=None, n=None):
    """After set_device and BEFORE the communicator exists: every rank publishes (host, card) through the rendezvous store (TCP,
    no GPU involved) and reads everybody else's.  A per-rank mask and a box-wide mask look the same to local_rank(); this is
    where the second case -- two ranks silently on one card -- turns into an error instead of a hang."""
    if backend != "nccl" and os.environ.get("BGAN_DIST_CHECK_DEVICES") != "1":
        return
    if os.environ.get("BGAN_DIST_CHECK_DEVICES") == "0":
        return
    if store is None:                   # fallback path only (no explicit store could be made): the group's own store
        get = getattr(td.distributed_c10d, "_get_default_store", None)
        if get is None:
            return
        store, me, n = get(), td.get_rank(), td.get_world_size()
    store.set(f"bgan_device_of_rank_{me}", _device_identity() or "?")
    ids = [store.get(f"bgan_device_of_rank_{r}").decode() for r in range(n)]
    ids = ["" if i == "?" else i for i in ids]
    # Nobody may leave (and, raising, tear its end of the store down -- rank 0 usually HOSTS it) before everybody has read every
    # identity: count the readers in, then let rank 0 wait for the others' acknowledgement; after its acknowledgement a rank does
    # not touch the store again.
    import time
    store.add("bgan_device_check_read", 1)
    t0 = time.time()
    while store.add("bgan_device_check_read", 0) < n and time.time() - t0 < 120:
        time.sleep(0.01)
    if me != 0:
        store.add("bgan_device_check_ack", 1)
    else:
        while store.add("bgan_device_check_ack", 0) < n - 1 and time.time() - t0 < 120:
            time.sleep(0.01)
    _distinct_devices(ids)


class _Done:
    def wait(self):
        return None


class _RecordedWork:
    """A collective issued while a step program is being recorded (program.Recorder): issuing it and waiting for it become
    host actions of the program, re-run at the same points of every replay."""

    def __init__(self, rec, issue):
        self._rec, self._issue = rec, issue
        self._cur = issue()
        rec.host_action(self._reissue)

    def _reissue(self):
        self._cur = self._issue()

    def _rewait(self):
        self._cur.wait()

    def wait(self):
        self._cur.wait()
        from . import program
        if program.active() is self._rec:
            self._rec.host_action(self._rewait)


def _recorder():
    from . import program
    return program.active()


def all_reduce_sum_async(flat):
    """Enqueues the in-place SUM all-reduce of ``flat`` behind the kernels enqueued so far and returns a handle whose
    ``wait()`` orders the current stream after it: work enqueued between the two overlaps the exchange (the BatchNorm
    backward statistics travel while the layer above computes its filter gradient).  No-op for a single replica."""
    if not collectives_active():
        return _Done()
    rec = _recorder()
    if rec is not None:
        return _RecordedWork(rec, lambda: _all_reduce_sum_async(flat))
    return _all_reduce_sum_async(flat)


def _all_reduce_sum_async(flat):
    if _use_abi_comm(flat):
        return AbiComm.get().all_reduce_async(flat)
    if flat.is_cuda and td.get_backend() == "gloo":      # CPU-side rehearsal: stage through the host, synchronous
        host = flat.cpu()
        td.all_reduce(host, op=td.ReduceOp.SUM)
        flat.copy_(host)
        return _Done()
    return td.all_reduce(flat, op=td.ReduceOp.SUM, async_op=True)     # RCCL over xGMI, on RCCL's own stream


def all_reduce_sum_(flat):
    """In-place SUM all-reduce of a flat buffer, IN LINE on the current stream (no-op for a single replica).  The forward SyncBN
    exchanges of the G-step have nothing to overlap with, so they skip the collective stream and its two event hops (round 3):
    the ABI communicator takes the compute stream as its stream argument, and torch's process group runs a synchronous
    (``async_op=False``) collective on the caller's current stream."""
    if not collectives_active():
        return flat
    rec = _recorder()
    if rec is not None:
        rec.host_action(lambda: _all_reduce_sum_inline(flat))
    _all_reduce_sum_inline(flat)
    return flat


def _all_reduce_sum_inline(flat):
    inline = os.environ.get("BGAN_DP_INLINE_SYNC", "1") != "0"      # 0: back on the collective stream with event hops (both routes)
    if _use_abi_comm(flat) and inline:
        AbiComm.get().all_reduce_inline(flat)
    elif flat.is_cuda and td.get_backend() != "gloo" and inline and not _use_abi_comm(flat):
        td.all_reduce(flat, op=td.ReduceOp.SUM, async_op=False)
    else:
        _all_reduce_sum_async(flat).wait()


class AbiComm:
    """The C ABI's own communicator (include/bgan.h: bg_comm_init / bg_allreduce_sum_f32), i.e. RCCL called from
    libbgan_hip.so instead of through torch.  The unique id travels over the already-initialised torch.distributed group
    (any backend); collectives run on a private stream ordered after the kernels enqueued so far, like torch's own.
    Selected with BGAN_DP_COLLECTIVE=abi; one communicator per process."""

    _inst = None

    def __init__(self):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        idbuf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        if rank() == 0:
            _lib.check(lib.bg_comm_unique_id(idbuf), "bg_comm_unique_id")
        box = [bytes(idbuf.raw)]
        if world_size() > 1:
            td.broadcast_object_list(box, src=0)
        self._lib, self._check = lib, _lib.check
        self._h = C.c_void_p()
        _lib.check(lib.bg_comm_init(C.byref(self._h), rank(), world_size(), box[0]), "bg_comm_init")
        self.stream = torch.cuda.Stream()

    @classmethod
    def get(cls):
        if cls._inst is None:
            cls._inst = cls()
        return cls._inst

    def all_reduce_async(self, seg):
        """Enqueues the in-place SUM of ``seg`` behind the current stream's work; returns an object with .wait()."""
        ev = torch.cuda.Event()
        ev.record()
        self.stream.wait_event(ev)
        self._check(self._lib.bg_allreduce_sum_f32(self._h, seg.data_ptr(), seg.numel(), self.stream.cuda_stream),
                    "bg_allreduce_sum_f32")
        done = torch.cuda.Event()
        done.record(self.stream)
        return _AbiWork(done)

    def all_reduce_inline(self, seg):
        """In-place SUM of ``seg`` as the next operation of the CURRENT stream (RCCL orders it against this communicator's
        collectives on the private stream by itself)."""
        self._check(self._lib.bg_allreduce_sum_f32(self._h, seg.data_ptr(), seg.numel(), torch.cuda.current_stream().cuda_stream),
                    "bg_allreduce_sum_f32")

    def close(self):
        if self._h:
            self._check(self._lib.bg_comm_destroy(self._h), "bg_comm_destroy")
            self._h = None
        AbiComm._inst = None


class _AbiWork:
    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


def _use_abi_comm(t):
    return t.is_cuda and os.environ.get("BGAN_DP_COLLECTIVE", "") == "abi"


class GradReducer:
    """SUM all-reduce of one network's flat gradient buffer, issued in buckets WHILE the backward that fills the buffer is
    still running: the engine reports each layer's slice as soon as its last contribution has been enqueued, adjacent slices
    are merged into buckets of >= ``bucket_bytes`` and each bucket goes out as an asynchronous collective (RCCL runs it on
    its own stream behind the kernels already enqueued; xGMI is point-to-point, so a few multi-MB messages per network beat
    one message per layer).  ``finish()`` reduces whatever was not reported and waits; Adam runs after it.
    With one replica every call is a no-op."""

    def __init__(self, flat, n, bucket_bytes=None):
        self.flat, self.n = flat, int(n)
        if bucket_bytes is None:
            # xGMI is point-to-point and a ring is bound by one link: a few 16 MB messages per network (critic 17 MB, generator
            # 47 MB) keep the per-message latency small against the transfer and still overlap the tail of the backward
            bucket_bytes = int(float(os.environ.get("BGAN_DP_BUCKET_MB", "16")) * (1 << 20))
        self.bucket = max(1, int(bucket_bytes) // 4)
        self.active = collectives_active()
        self.pending = None          # (lo, hi) merged, not yet issued
        self.covered = []            # issued ranges
        self.works = []
        self.n_collectives = 0

    def _issue(self, lo, hi):
        if hi <= lo:
            return
        seg = self.flat[lo:hi]
        self.n_collectives += 1
        from . import ops
        rec = _recorder()
        if rec is not None:      # step program: this bucket goes out at the same point of every replay
            rec.host_action(lambda: self._issue_seg(seg))
        with ops.trace_range("all_reduce"):
            self._issue_seg(seg)
        self.covered.append((lo, hi))

    def _issue_seg(self, seg):
        if _use_abi_comm(seg):
            self.works.append(AbiComm.get().all_reduce_async(seg))
        elif seg.is_cuda and td.get_backend() == "gloo":        # CPU-side rehearsal: stage through the host, synchronous
            host = seg.cpu()
            td.all_reduce(host, op=td.ReduceOp.SUM)
            seg.copy_(host)
        else:
            self.works.append(td.all_reduce(seg, op=td.ReduceOp.SUM, async_op=True))

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
        rec = _recorder()
        if rec is not None:
            rec.host_action(self._wait_all)
        self._wait_all()

    def _wait_all(self):
        for w in self.works:
            w.wait()
        self.works = []


def _quiesce_abi():
    """Two communicators must never have collectives in flight at once (the classic RCCL/NCCL cross-communicator deadlock):
    before anything goes through torch's communicator, the C ABI's private stream is drained."""
    if AbiComm._inst is not None:
        AbiComm._inst.stream.synchronize()


def evidence():
    """What the collective layer itself says about the group, for a bench line: the backend, the number of
    peers that TOOK PART in a collective (a SUM all-reduce of ones through the very communicator the gradients use -- not
    WORLD_SIZE from the environment), every rank's physical device (uuid | PCI id, all-gathered), and, when the C ABI's
    communicator is the route, ncclCommCount / ncclCommUserRank as RCCL reports them.  {} when no group was joined."""
    if not is_initialized():
        return {}
    _quiesce_abi()
    backend = td.get_backend()
    on_gpu = backend == "nccl"
    t = torch.ones(1, dtype=torch.float32, device="cuda" if on_gpu else "cpu")
    td.all_reduce(t)
    ident = _device_identity() if torch.cuda.is_available() else ""
    ids = [None] * td.get_world_size()
    td.all_gather_object(ids, ident)
    out = {"backend": backend + (" (RCCL)" if on_gpu else ""), "world_size_env": int(os.environ.get("WORLD_SIZE", "1")),
           "collective_nranks": int(round(float(t.item()))), "devices": ids,
           "distinct_devices": len({i for i in ids if i}), "collective_route": os.environ.get("BGAN_DP_COLLECTIVE") or "torch"}
    if on_gpu:
        out["rccl_nranks"] = out["collective_nranks"]
    if AbiComm._inst is not None:
        import ctypes as C
        n, r = C.c_int(), C.c_int()
        AbiComm._inst._check(AbiComm._inst._lib.bg_comm_query(AbiComm._inst._h, C.byref(n), C.byref(r)), "bg_comm_query")
        out["abi_comm"] = {"nranks": n.value, "rank": r.value}
    return out


def max_over_ranks(value: float) -> float:
    """MAX of a host scalar over the replicas (bench timing)."""
    if world_size() <= 1:
        return float(value)
    _quiesce_abi()
    dev = "cuda" if (td.get_backend() == "nccl") else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    td.all_reduce(t, op=td.ReduceOp.MAX)
    return float(t.item())


def broadcast_(flat, src=0):
    if world_size() > 1:
        _quiesce_abi()
        td.broadcast(flat, src=src)
    return flat


def broadcast_object(obj, src=0):
    """A small Python object from rank ``src`` to everyone (run directory names in the demos)."""
    if world_size() <= 1:
        return obj
    _quiesce_abi()
    box = [obj]
    td.broadcast_object_list(box, src=src)
    return box[0]


def barrier():
    if collectives_active():          # also in the one-rank rehearsal: its first collective is what creates the communicator
        _quiesce_abi()
        td.barrier()


def shutdown():
    """Leaves the process group (no-op when none was joined): RCCL communicators are torn down before interpreter exit."""
    if AbiComm._inst is not None:
        AbiComm._inst.close()
    if is_initialized():
        td.destroy_process_group()


def shard(batch, r=None, n=None):
    """The slice of a global batch this replica trains on (equal shards)."""
    r = rank() if r is None else r
    n = world_size() if n is None else n
    per = batch.shape[0] // n
    return batch[r * per:(r + 1) * per]
