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
from collections import OrderedDict

from . import _lib

N_SLOTS = 64
_active = None              # the Recorder the current thread records into (ops / dist / wgan consult it)


def active():
    return _active


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
        global _active
        assert _active is None, "a step program is already being recorded"
        _lib.check(self.lib.bg_program_record_begin(self.h), "bg_program_record_begin")
        _active = self
        return self

    def __exit__(self, et, ev, tb):
        global _active
        _active = None
        rc = self.lib.bg_program_record_end(self.h)
        if et is None:
            _lib.check(rc, "bg_program_record_end")
            self.n_nodes = self.lib.bg_program_size(self.h)
            self.n_launches = self.lib.bg_program_launches(self.h)
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
        self.stats = {"eager": 0, "recorded": 0, "replayed": 0}
        self.last_was_replay = False

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
            _, e = self.entries.popitem(last=False)
            if isinstance(e, Recorder):
                e.close()
