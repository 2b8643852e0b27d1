"""Step programs (include/bgan.h bg_program_* / bg_dstep / bg_gstep; blurred_gan_amd/program.py): the launch list of a
discriminator_step / generator_step (wgan.py:132-172) recorded once and replayed with one call must leave EXACTLY the state the
eager path leaves -- weights, Adam slots, BatchNorm statistics, metrics, RNG stream positions -- with sigma changing every step
(so the tap VALUES change under the recorded program, and the tap COUNT changes mid-run: a new program) and the build's own RNG."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = {"tiny": (8, 8, 3), "mnist": (28, 28, 1), "celeba64": (64, 64, 3), "celeba128": (128, 128, 3)}


def _make(arch, B, seed, tmp_path, replay, sigma0, **kw):
    import blurred_gan_amd as bg
    from blurred_gan_amd import models
    bg.set_seed(seed)
    gen, disc = models.DCGANGenerator(arch=arch), models.DCGANDiscriminator(arch=arch)
    hp = bg.BlurredWGANGP.HyperParameters(initial_blur_std=sigma0, global_batch_size=B, batch_size=B)
    return bg.BlurredWGANGP(gen, disc, hp, bg.TrainingConfig(log_dir=str(tmp_path / "log")), step_replay=replay, **kw)


def _same_state(a, b):
    for ma, mb in ((a.generator, b.generator), (a.discriminator, b.discriminator)):
        for name in ("theta", "state", "m", "v", "grad"):
            assert torch.equal(getattr(ma.store, name), getattr(mb.store, name)), name
        assert ma.optimizer.iterations == mb.optimizer.iterations
        assert ma.net().rng_offset == mb.net().rng_offset
    assert a._rng_off == b._rng_off and int(a.n_batches) == int(b.n_batches) and int(a.n_img) == int(b.n_img)


@pytest.mark.parametrize("arch,B,sigmas", [
    ("tiny", 4, [0.9, 0.88, 0.86, 0.84, 0.82, 0.5, 0.49, 0.48, 0.47]),             # 5 taps ... then 3 taps: a second program
    ("mnist", 8, [2.0, 1.98, 1.96, 1.94, 1.92, 1.9, 1.0, 0.99, 0.98, 0.97]),       # 13 taps ... then 7
    ("celeba64", 8, [5.0, 4.99, 4.98, 4.97, 4.96, 4.95]),                            # 31 taps throughout
    ("celeba128", 4, [5.0, 4.99, 4.98, 4.97, 4.96]),                                 # the verbatim 128-pixel stacks (16-channel kernels)
])
def test_replayed_steps_equal_eager_steps_bit_for_bit(tmp_path, arch, B, sigmas):
    eager = _make(arch, B, 21, tmp_path, False, sigmas[0])
    prog = _make(arch, B, 21, tmp_path, True, sigmas[0])
    g = torch.Generator().manual_seed(3)
    for i, s in enumerate(sigmas):
        reals = (torch.rand(B, *SHAPES[arch], generator=g) * 2 - 1).cuda()
        eager.std.assign(s)
        prog.std.assign(s)
        me = eager.train_on_batch(reals)
        mp_ = prog.train_on_batch(reals.clone())          # another address every step: the program must not care
        assert me == mp_, (i, me, mp_)
        _same_state(eager, prog)
        assert torch.equal(eager.images[0], prog.images[0]) and torch.equal(eager.images[1], prog.images[1])
    st = prog._programs.stats
    assert st["replayed"] >= 4, st                       # D and G programs replayed (a changed tap count costs one eager + one recording step)
    assert eager._programs.stats["replayed"] == 0
    recs = [e for e in prog._programs.entries.values() if hasattr(e, "n_launches")]
    assert recs and all(r.n_launches > 10 for r in recs)


def test_persistent_input_buffers_are_recorded_directly(tmp_path):
    """persistent_input=True: a loader that refills its own device buffers in place (two of them, alternating): the programs are
    recorded on those buffers (one program per address, no staging buffer, no copy) and leave the eager path's state."""
    arch, B = "mnist", 8
    eager = _make(arch, B, 11, tmp_path, False, 1.0)
    prog = _make(arch, B, 11, tmp_path, True, 1.0, persistent_input=True)
    g = torch.Generator().manual_seed(4)
    bufs = [torch.empty(B, 28, 28, 1, device="cuda") for _ in range(2)]
    for i in range(10):
        batch = (torch.rand(B, 28, 28, 1, generator=g) * 2 - 1).cuda()
        bufs[i % 2].copy_(batch)
        assert eager.train_on_batch(batch) == prog.train_on_batch(bufs[i % 2]), i
        _same_state(eager, prog)
    assert prog._reals_stage is None
    st = prog._programs.stats                            # per address one eager and one recording step (D and G), one more eager
    assert st["replayed"] >= 10 and st["recorded"] == 4, st   # D-step while the generator's transposed kernels were still dirty


def test_eager_and_replayed_steps_can_be_mixed(tmp_path):
    """discriminator_step / generator_step called directly (eager) between replayed train_on_batch calls, generate_samples in
    between, d_steps_per_g_step = 2 (batches without a G-step: other dirty flags, another program)."""
    arch, B = "mnist", 8
    a = _make(arch, B, 5, tmp_path, False, 1.0)
    b = _make(arch, B, 5, tmp_path, True, 1.0)
    for m in (a, b):
        m.d_steps_per_g_step = 2
    g = torch.Generator().manual_seed(9)
    data = [(torch.rand(B, 28, 28, 1, generator=g) * 2 - 1).cuda() for _ in range(12)]
    z = torch.rand(B, 100, generator=g).cuda()
    for i, reals in enumerate(data):
        if i == 7:                                        # a hand-driven step in the middle, on both
            for m in (a, b):
                m.discriminator_step(reals)
                m.generator_step()
                m.n_batches.assign_add(1)
            _same_state(a, b)
            continue
        ra, rb = a.train_on_batch(reals), b.train_on_batch(reals)
        assert ra == rb, (i, ra, rb)
        if i % 3 == 0:
            assert torch.equal(a.generate_samples(z), b.generate_samples(z))
        _same_state(a, b)
    assert b._programs.stats["replayed"] >= 6


def test_program_api_replay_and_graph_launch_match_direct_calls():
    """The C ABI by itself: record two launches (one with a bound RNG offset), replay by node range, as bg_dstep, and as a hipGraph."""
    import ctypes as C
    from blurred_gan_amd import _lib, ops
    lib = _lib.load()
    x = torch.zeros(1000, device="cuda")
    y = torch.zeros(1000, device="cuda")
    h = C.c_void_p()
    _lib.check(lib.bg_program_create(C.byref(h), 8), "create")
    u64 = (C.c_uint64 * 8).from_address(lib.bg_program_slots_u64(h))
    s = ops._stream()
    _lib.check(lib.bg_program_record_begin(h), "begin")
    _lib.check(lib.bg_program_bind_next(_lib.BIND_RNG_OFFSET, 2), "bind")
    ops.uniform(x, 77, 5)
    ops.scale_(x, 2.0)
    _lib.check(lib.bg_program_record_end(h), "end")
    assert lib.bg_program_launches(h) == 2 and lib.bg_program_binds(h) == 1 and lib.bg_program_size(h) >= 2
    want5 = ops.scale_(ops.uniform(torch.zeros(1000, device="cuda"), 77, 5), 2.0)
    assert torch.equal(x, want5)                          # the recording step itself ran from the recorded nodes
    want9 = ops.scale_(ops.uniform(torch.zeros(1000, device="cuda"), 77, 9), 2.0)
    u64[2] = 9
    _lib.check(lib.bg_dstep(h, s), "bg_dstep")
    assert torch.equal(x, want9)
    u64[2] = 5
    _lib.check(lib.bg_program_graph_launch(h, 0, -1, s), "graph")
    assert torch.equal(x, want5)
    u64[2] = 9
    _lib.check(lib.bg_program_graph_launch(h, 0, -1, s), "graph (bound argument refreshed)")
    assert torch.equal(x, want9)
    # a replay under the profiling hooks yields the same records as the eager calls
    ops.prof_reset(); ops.prof_enable(True)
    _lib.check(lib.bg_gstep(h, s), "bg_gstep")
    ops.prof_enable(False)
    names = [r[0] for r in ops.prof_records()]
    ops.prof_reset()
    assert names == ["rng_uniform", "scale"] or (len(names) == 2 and names[0] == "rng_uniform"), names
    _lib.check(lib.bg_program_destroy(h), "destroy")
    del y


def test_graph_replay_of_a_whole_step_equals_eager(tmp_path, monkeypatch):
    monkeypatch.setenv("BGAN_STEP_GRAPH", "1")
    arch, B = "mnist", 8
    a = _make(arch, B, 31, tmp_path, False, 1.5)
    b = _make(arch, B, 31, tmp_path, True, 1.5)
    g = torch.Generator().manual_seed(4)
    for i in range(6):
        reals = (torch.rand(B, 28, 28, 1, generator=g) * 2 - 1).cuda()
        for m in (a, b):
            m.std.assign(1.5 - 0.01 * i)
        assert a.train_on_batch(reals) == b.train_on_batch(reals)
        _same_state(a, b)
    assert b._programs.stats["replayed"] >= 6


def test_alternating_batch_shapes_keep_their_programs(tmp_path):
    """the partial last batch of an epoch has another shape; the staging buffer used to be re-allocated at every switch,
    which stranded the recorded programs of BOTH shapes (their key holds the buffer's address).  One staging buffer per shape:
    after the first epoch nothing is recorded again, and the steps still equal the eager path's bit for bit."""
    eager = _make("tiny", 4, 5, tmp_path, False, 0.9)
    prog = _make("tiny", 4, 5, tmp_path, True, 0.9)
    g = torch.Generator().manual_seed(11)
    recorded = []
    for epoch in range(4):
        for B in (4, 4, 4, 4, 3, 3, 3):                 # full batches, then "partial" ones (three each so that both shapes get to replay)
            reals = (torch.rand(B, *SHAPES["tiny"], generator=g) * 2 - 1).cuda()
            assert eager.train_on_batch(reals) == prog.train_on_batch(reals.clone())
        recorded.append(prog._programs.stats["recorded"])
    _same_state(eager, prog)
    assert recorded[0] == 4, recorded                     # D and G programs of two shapes
    assert recorded[1:] == [recorded[0]] * 3, recorded    # ... and never again
    assert prog._programs.stats["rerecorded"] == 0 and len(prog._reals_stage) == 2
