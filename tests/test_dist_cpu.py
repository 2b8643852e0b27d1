"""World-size-2 `gloo` tests of the data-parallel path (CPU): the all-reduce plumbing of blurred_gan_amd.dist and
the DP semantics the step uses: SUM of the shards' critic gradients == the single-process gradient
at the global batch (Q1 vector-loss factor and GP mean use the GLOBAL batch)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from blurred_gan_amd import dist
    from oracle import step as S
    assert dist.init_from_env(backend="gloo") == world and dist.rank() == rank and dist.world_size() == world
    arch, Bg = "tiny", 6
    rng = np.random.default_rng(11)                       # identical on every rank: same weights, same global batch
    st = S.new_state(arch, rng, np.float64, std=0.8)
    reals = rng.uniform(-1, 1, size=(Bg, 8, 8, 3))
    rnd = S.draw_randomness(arch, Bg, rng, np.float64)
    sh = lambda a: dist.shard(a)                          # this rank's equal slice of the global batch
    rnd_local = {k: ([sh(m) for m in v] if isinstance(v, list) else sh(v)) for k, v in rnd.items()}
    hp = dict(S.DEFAULT_HP, global_batch_size=Bg, dp_world=world)
    dg, met, _ = S.discriminator_grads(st, sh(reals), rnd_local, hp)
    flat = torch.from_numpy(np.concatenate([g[k].ravel() for g in dg for k in ("kernel", "bias") if k in g]))
    dist.all_reduce_sum_(flat)                            # the collective the product issues per network
    gg, _, gm = S.generator_grads(st, rnd_local, hp, Bg // world)
    gflat = torch.from_numpy(np.concatenate([g[k].ravel() for g in gg for k in ("kernel", "gamma", "beta") if k in g]))
    local = gflat.clone()
    dist.all_reduce_sum_(gflat)
    parts = [torch.zeros_like(local) for _ in range(world)]
    torch.distributed.all_gather(parts, local)
    assert torch.allclose(gflat, sum(parts), rtol=0, atol=1e-12)        # SUM semantics
    # bucketed reducer: slices reported out of order, with gaps and an overlap, buckets smaller than some slices
    g = torch.Generator().manual_seed(100 + rank)
    buf = torch.rand(1000, generator=g, dtype=torch.float64)
    mine = buf.clone()
    red = dist.GradReducer(buf, 990, bucket_bytes=4 * 64)
    for lo, hi in [(900, 990), (800, 900), (300, 420), (400, 500), (0, 16)]:
        red.ready(lo, hi)
    red.finish()
    parts = [torch.zeros_like(mine) for _ in range(world)]
    torch.distributed.all_gather(parts, mine)
    assert torch.allclose(buf[:990], sum(parts)[:990], rtol=0, atol=1e-12), "every element reduced exactly once"
    assert torch.equal(buf[990:], mine[990:]), "elements past n untouched"
    assert red.n_collectives >= 4
    dist.barrier()
    if rank == 0:
        np.save(os.path.join(out_dir, "d_allreduced.npy"), flat.numpy())
    torch.distributed.destroy_process_group()


def test_dp_sum_of_shard_gradients_equals_global_batch_gradient(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    from oracle import step as S
    arch, Bg = "tiny", 6
    rng = np.random.default_rng(11)
    st = S.new_state(arch, rng, np.float64, std=0.8)
    reals = rng.uniform(-1, 1, size=(Bg, 8, 8, 3))
    rnd = S.draw_randomness(arch, Bg, rng, np.float64)
    dg, _, _ = S.discriminator_grads(st, reals, rnd, dict(S.DEFAULT_HP, global_batch_size=Bg))
    ref = np.concatenate([g[k].ravel() for g in dg for k in ("kernel", "bias") if k in g])
    got = np.load(tmp_path / "d_allreduced.npy")
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-11)


def test_single_process_dist_helpers_are_noops():
    from blurred_gan_amd import dist
    assert dist.world_size() == 1 and dist.rank() == 0
    t = torch.arange(4.0)
    assert torch.equal(dist.all_reduce_sum_(t.clone()), t)
    assert torch.equal(dist.shard(torch.arange(8), r=1, n=4), torch.tensor([2, 3]))
    red = dist.GradReducer(t.clone(), 4)
    red.ready(0, 2)
    red.finish()
    assert red.n_collectives == 0 and dist.max_over_ranks(1.5) == 1.5


def test_two_ranks_on_one_physical_device_are_refused_not_hung():
    """a box-wide device mask and a per-rank mask look the same to local_rank(); what tells them apart is the
    identity of the card each rank ended up on, exchanged through the rendezvous store before the first collective."""
    from blurred_gan_amd import dist
    dist._distinct_devices(["hostA|uuid-1|0:1:0", "hostA|uuid-2|0:2:0", "hostB|uuid-1|0:1:0"])       # same card id on another host: fine
    dist._distinct_devices(["", "", "hostA|uuid-1|0:1:0", ""])                # a runtime that reports neither uuid nor PCI ids: unknown, never "shared"
    with pytest.raises(RuntimeError, match="ranks 0 and 2 both run on device"):
        dist._distinct_devices(["hostA|uuid-1|0:1:0", "hostA|uuid-2|0:2:0", "hostA|uuid-1|0:1:0"])


def _same_device_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      BGAN_DIST_CHECK_DEVICES="1")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from blurred_gan_amd import dist
    dist._device_identity = lambda: "thishost|the-one-card|0:5:0"        # an identical mask for both local ranks
    try:
        dist.init_from_env(backend="gloo")
        msg = "no error"
    except RuntimeError as e:
        msg = str(e)
    open(os.path.join(out_dir, f"msg_{rank}.txt"), "w").write(msg)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def test_identical_device_mask_for_two_local_ranks_raises_on_every_rank(tmp_path):
    mp.spawn(_same_device_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert "both run on device thishost|the-one-card" in open(tmp_path / f"msg_{r}.txt").read()


def test_step_program_api_without_a_gpu():
    """include/bgan.h bg_program_*: argument checking and the recorder's host-action bookkeeping need no device."""
    import ctypes as C
    from blurred_gan_amd import _lib, program
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.bg_program_create(None, 4) == -6 and lib.bg_program_create(C.byref(h), -1) == -1
    assert lib.bg_program_create(C.byref(h), 4) == 0
    assert lib.bg_program_bind_next(_lib.BIND_ADAM_LR, 0) == -3               # nothing is being recorded
    assert lib.bg_program_record_begin(h) == 0
    assert lib.bg_program_record_begin(h) == -3                                # one recording per thread
    assert lib.bg_program_replay(h, 0, -1, None) == -3                         # still recording
    assert lib.bg_program_bind_next(99, 0) == -1 and lib.bg_program_bind_next(_lib.BIND_ADAM_LR, 4) == -1
    assert lib.bg_program_bind_next(_lib.BIND_RNG_OFFSET, 3) == 0
    assert lib.bg_program_record_end(h) == -3                                  # a binding no launch consumed
    assert b"no launch consumed" in lib.bg_last_error()
    assert lib.bg_program_record_begin(h) == 0 and lib.bg_program_record_end(h) == 0
    assert lib.bg_program_size(h) == 0 and lib.bg_program_launches(h) == 0 and lib.bg_program_binds(h) == 0
    assert lib.bg_program_replay(h, 0, -1, None) == 0 and lib.bg_dstep(h, None) == 0 and lib.bg_gstep(h, None) == 0
    assert lib.bg_program_replay(h, 1, 0, None) == -1 and lib.bg_program_replay(None, 0, -1, None) == -6
    assert lib.bg_program_destroy(h) == 0 and lib.bg_program_destroy(None) == 0
    # host actions and per-replay updates run in recording order around the (here empty) stretches of launches
    log = []
    rec = program.Recorder()
    with rec:
        rec.host_action(lambda: log.append("a"))
        rec.updates.append(lambda: log.append("u"))
        rec.host_action(lambda: log.append("b"))
        rec.result = "r"
    assert program.active() is None and rec.n_launches == 0
    assert rec.replay(None) == "r" and log == ["u", "a", "b"]
    progs = program.StepPrograms(capacity=2)
    calls = []
    for _ in range(3):
        progs.run("k", lambda: calls.append(1) or "x", None)
    assert calls == [1, 1] and progs.stats == {"eager": 1, "recorded": 1, "replayed": 1, "evicted": 0, "rerecorded": 0} and progs.last_was_replay
    progs.run("k2", lambda: None, None); progs.run("k3", lambda: None, None)
    assert list(progs.entries) == ["k2", "k3"]                                 # least recently used program dropped


def test_recorder_state_is_per_thread_like_the_library():
    """program.py's recording state was process-global while csrc/program.hip's is thread_local: a library call from
    a second thread during a recording step landed in the wrong recorder.  Two threads now record two programs at once, and a
    thread that is not recording sees no active recorder."""
    import threading
    from blurred_gan_amd import program
    seen, errs = {}, []
    gate = threading.Barrier(2, timeout=30)

    def worker(name):
        try:
            rec = program.Recorder()
            with rec:
                gate.wait()                               # both threads are inside their recording at the same time
                seen[name] = program.active() is rec and program._active is rec
                gate.wait()
            seen[name] = seen[name] and program.active() is None
            rec.close()
        except Exception as e:                            # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=worker, args=(n,)) for n in ("a", "b")]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
    assert not errs, errs
    assert seen == {"a": True, "b": True}
    assert program.active() is None                       # the main thread never recorded


def test_step_programs_warn_when_an_evicted_key_comes_back():
    """StepPrograms is an LRU of `capacity` programs keyed (among others) on the batch's device address; a loader
    cycling through more buffers than that re-records for ever.  That now shows: a warning and a 'rerecorded' count."""
    import warnings
    from blurred_gan_amd import program
    sp = program.StepPrograms(capacity=2)
    calls = []
    for rnd in range(3):
        for key in ("k0", "k1", "k2"):                    # three keys through a cache of two
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                sp.run(key, lambda: calls.append(key), None)
                if rnd == 0:
                    assert not w
    # round 0: three eager runs (k0 evicted as a placeholder, never recorded); nothing was RECORDED and evicted yet
    assert sp.stats["evicted"] == 0 and sp.stats["rerecorded"] == 0
    # now with real recordings: capacity 1, two keys alternating -> every return to a key finds its program evicted
    sp = program.StepPrograms(capacity=1)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        for _ in range(3):
            for key in ("a", "b"):
                sp.run(key, lambda: None, None)           # eager
                sp.run(key, lambda: None, None)           # recorded (an empty program)
    assert sp.stats["evicted"] >= 4 and sp.stats["rerecorded"] >= 4, sp.stats
    assert sum("re-recorded continuously" in str(x.message) for x in w) == 1      # warned once, not per step
    sp.clear()


def test_step_key_normalises_numpy_hyperparameters():
    """np.float32 hyper-parameters fell through the isinstance filter of the step-program key."""
    import numpy as np
    from blurred_gan_amd import wgan
    assert wgan._plain(np.float32(0.5)) == 0.5 and type(wgan._plain(np.float32(0.5))) is float
    assert wgan._plain(np.int64(3)) == 3 and type(wgan._plain(np.int64(3))) is int
    assert wgan._plain(True) is True and wgan._plain("adam") == "adam"
    assert len(wgan._env_switches()) == len(wgan._KEY_ENV)


def _evidence_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from blurred_gan_amd import dist
    dist.init_from_env(backend="gloo")
    ev = dist.evidence()
    json.dump(ev, open(os.path.join(out_dir, f"ev_{rank}.json"), "w"))
    dist.shutdown()


def test_dp_evidence_counts_the_peers_through_the_communicator(tmp_path):
    """bench.py's `dp` object with two gloo ranks on the CPU: the peer count comes from an all-reduce of ones
    through the group, not from WORLD_SIZE; every rank reports the same picture; no RCCL claim on a gloo group."""
    import json
    mp.spawn(_evidence_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    ev = [json.load(open(tmp_path / f"ev_{r}.json")) for r in range(2)]
    assert ev[0] == ev[1]
    assert ev[0]["backend"] == "gloo" and ev[0]["collective_nranks"] == 2 and ev[0]["world_size_env"] == 2
    assert len(ev[0]["devices"]) == 2 and "rccl_nranks" not in ev[0] and ev[0]["collective_route"] == "torch"
    from blurred_gan_amd import dist
    assert dist.evidence() == {}                              # no group joined in this process
