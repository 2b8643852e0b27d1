"""Step programs on the host side: record a step's launch list once, replay it with one call into the library
(include/bgan.h ``bg_program_*`` / ``bg_dstep`` / ``bg_gstep``; the reference's per-batch Python pass over TF ops,
wgan.py:86-114,132-172, happens ONCE per program here).

``StepPrograms.run(kind, key, fn)`` drives one kind of step ("d" / "g") of one model:
  first call with a key    -> ``fn()`` eagerly (contexts, workspaces and optimiser slots get allocated);
  second call              -> ``fn()`` eagerly again, but every launch is also recorded (``Recorder``);
  every later call         -> the recorded program is replayed: the per-step scalars are written into the program's slots
                              (Adam's lr_t, the RNG counter offsets -- the host counters advance exactly as the eager path
                              advances them, so eager and replayed steps can be mixed freely and stay bit-identical), then ONE
                              ctypes call issues every kernel.  Data-parallel collectives are host actions: the replay is
                              split at the node indices where they were issued while recording.
The key holds everything that shapes the launch list (batch, tap count of the blur, switches, hyper-parameters baked into kernel
arguments, dirty flags of the transposed weight copies); a new key is a new program."""
from __future__ import annotations

import ctypes as C
import math
import os
import threading
import warnings
from collections import OrderedDict

from . import _lib

N_SLOTS = 64


class _ThreadState(threading.local):
    """The Recorder the CURRENT THREAD records into (ops / dist / wgan consult it).  Per thread, like the library's own recorder
    state (csrc/program.hip: thread_local): a prefetch or metric thread that issues library calls while another thread records
    neither lands in that program's keep list nor announces binds to a recorder that is not its own, and two models may record in
    two threads at once."""
    rec = None


_tls = _ThreadState()


def active():
    return _tls.rec


def __getattr__(name):                 # program._active: the historical spelling, read-only
    if name == "_active":
        return _tls.rec
    raise AttributeError(name)


def enabled_by_env():
    return os.environ.get("BGAN_NO_STEP_REPLAY") != "1"


def graph_by_env():
    return os.environ.get("BGAN_STEP_GRAPH") == "1"


class Recorder:
    def __init__(self):
        self.lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self.lib.bg_program_create(C.byref(h), N_SLOTS), "bg_program_create")
        self.h = h
        self.f64 = (C.c_double * N_SLOTS).from_address(self.lib.bg_program_slots_f64(h))
        self.u64 = (C.c_uint64 * N_SLOTS).from_address(self.lib.bg_program_slots_u64(h))
        self.n_slots = 0
        self.keep = []          # every tensor whose address the program holds: alive as long as the program
        self.updates = []       # callables run before a replay, in recording order
        self.actions = []       # (node index, callable): host work (collectives) between two stretches of launches
        self.exit_state = []    # (obj, attr, value): host flags as the recorded step left them
        self.result = None
        self.n_nodes = self.n_launches = 0
        self.use_graph = graph_by_env()

    def close(self):
        if self.h:
            self.lib.bg_program_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- while recording
    def __enter__(self):
        assert _tls.rec is None, "this thread is already recording a step program"
        _lib.check(self.lib.bg_program_record_begin(self.h), "bg_program_record_begin")
        _tls.rec = self
        return self

    def __exit__(self, et, ev, tb):
        _tls.rec = None
        rc = self.lib.bg_program_record_end(self.h)
        if et is None:
            _lib.check(rc, "bg_program_record_end")
            self.n_nodes = self.lib.bg_program_size(self.h)
            self.n_launches = self.lib.bg_program_launches(self.h)
            # every announced slot must have become a binding of a launch argument: a dropped one would replay the recording
            # step's lr_t / Philox offset for ever (same noise and masks every step) without any error
            nb = self.lib.bg_program_binds(self.h)
            if nb != self.n_slots:
                raise _lib.BgError(f"step program: {self.n_slots} per-step values were announced but {nb} launch arguments were bound")
        return False

    def _slot(self):
        s = self.n_slots
        if s >= N_SLOTS:
            raise _lib.BgError("step program: out of slots")
        self.n_slots += 1
        return s

    def bind_rng(self, obj, attr, inc):
        """The next uniform / keep-mask launch draws at ``getattr(obj, attr)``, which advances by ``inc`` per step."""
        s = self._slot()
        _lib.check(self.lib.bg_program_bind_next(_lib.BIND_RNG_OFFSET, s), "bg_program_bind_next")
        u64 = self.u64

        def update():
            v = getattr(obj, attr)
            u64[s] = v
            setattr(obj, attr, v + inc)
        self.updates.append(update)

    def bind_adam(self, opt, b1, b2):
        """The next bg_adam_f32 launch takes lr_t of ``opt``'s next iteration."""
        s = self._slot()
        _lib.check(self.lib.bg_program_bind_next(_lib.BIND_ADAM_LR, s), "bg_program_bind_next")
        f64 = self.f64

        def update():
            opt.iterations += 1
            t = opt.iterations
            f64[s] = float(opt.learning_rate) * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
        self.updates.append(update)

    def host_action(self, fn):
        """``fn`` (a collective, a wait) runs at this point of every replay; the caller runs it itself now."""
        self.actions.append((self.lib.bg_program_size(self.h), fn))

    # ---- replay
    def replay(self, stream):
        for u in self.updates:
            u()
        lib, h = self.lib, self.h
        run = lib.bg_program_graph_launch if self.use_graph else lib.bg_program_replay
        if not self.actions:
            rc = run(h, 0, -1, stream)
            if rc:
                _lib.check(rc, "bg_program_replay")
        else:
            pos = 0
            for idx, fn in self.actions:
                if idx > pos:
                    rc = run(h, pos, idx, stream)
                    if rc:
                        _lib.check(rc, "bg_program_replay")
                    pos = idx
                fn()
            if pos < self.n_nodes:
                rc = run(h, pos, -1, stream)
                if rc:
                    _lib.check(rc, "bg_program_replay")
        for obj, attr, val in self.exit_state:
            setattr(obj, attr, val)
        return self.result


class StepPrograms:
    """The programs of one model, by key.  ``warm`` eager runs precede the recording run of a key."""

    def __init__(self, capacity=24):
        self.entries = OrderedDict()
        self.capacity = capacity
        self.stats = {"eager": 0, "recorded": 0, "replayed": 0, "evicted": 0, "rerecorded": 0}
        self.last_was_replay = False
        self._evicted = set()       # hashes of keys whose recorded program was dropped by the LRU rule
        self._warned = False

    def clear(self):
        for e in self.entries.values():
            if isinstance(e, Recorder):
                e.close()
        self.entries.clear()

    def run(self, key, fn, stream, exit_state=None):
        e = self.entries.get(key)
        self.last_was_replay = isinstance(e, Recorder)
        if isinstance(e, Recorder):
            self.entries.move_to_end(key)
            self.stats["replayed"] += 1
            return e.replay(stream)
        if e is None:
            if hash(key) in self._evicted:
                # the working set of keys is larger than the cache: every return to an evicted key costs an eager step and a
                # recording (e.g. persistent_input=True with a ring of more device buffers than `capacity` programs)
                self.stats["rerecorded"] += 1
                if not self._warned:
                    self._warned = True
                    warnings.warn(f"step programs: a key evicted from the cache of {self.capacity} programs is in use again -- the steps "
                                  "cycle through more distinct (batch shape, input buffer, tap count, ...) combinations than the cache "
                                  "holds and are re-recorded continuously; raise StepPrograms.capacity or feed fewer distinct input "
                                  "buffers (persistent_input=True keys programs on the batch's device address)", RuntimeWarning, stacklevel=3)
            self.entries[key] = 0
            self.stats["eager"] += 1
            self._trim()
            return fn()
        rec = Recorder()
        with rec:
            rec.result = fn()
        if exit_state is not None:
            rec.exit_state = exit_state()
        self.entries[key] = rec
        self.stats["recorded"] += 1
        return rec.result

    def _trim(self):
        while len(self.entries) > self.capacity:
            k, e = self.entries.popitem(last=False)
            if isinstance(e, Recorder):
                self.stats["evicted"] += 1
                self._evicted.add(hash(k))
                e.close()
