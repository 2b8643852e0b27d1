"""GPU check of the data-parallel path with 2 ranks sharing the one GPU of the test box (gloo rendezvous, gradients
staged through the host): SUM of the ranks' critic gradients from the HIP step == the single-process HIP gradients
at the global batch; both replicas end the step with identical weights."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(arch, B, gbs, seed=3):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import blurred_gan_amd as bg
    from blurred_gan_amd import models
    from oracle import step as S
    from helpers import load_oracle_weights
    rng = np.random.default_rng(seed)
    st = S.new_state(arch, rng, np.float64, std=0.8)
    gen, disc = models.DCGANGenerator(arch=arch), models.DCGANDiscriminator(arch=arch)
    hp = bg.BlurredWGANGP.HyperParameters(initial_blur_std=0.8, global_batch_size=gbs, batch_size=B)
    gan = bg.BlurredWGANGP(gen, disc, hp, bg.TrainingConfig(log_dir="/tmp/bg_dp_logs"))
    load_oracle_weights(gen, st["g"])
    load_oracle_weights(disc, st["d"])
    reals = rng.uniform(-1, 1, size=(gbs,) + tuple(models.IMAGE_SHAPE[arch])).astype(np.float32)
    rnd = S.draw_randomness(arch, gbs, rng, np.float64)
    return gan, reals, rnd


def _worker(rank, world, port, out_dir, arch="tiny", B_local=3):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from blurred_gan_amd import dist
    torch.cuda.set_device(0)
    torch.distributed.init_process_group(backend="gloo")
    gan, reals, rnd = _build(arch, B_local, B_local * world)
    sh = lambda a: dist.shard(torch.from_numpy(np.asarray(a))).numpy()
    rnd_local = {k: ([sh(m) for m in v] if isinstance(v, list) else sh(v)) for k, v in rnd.items()}
    gan.train_on_batch(sh(reals), randomness=rnd_local)
    st = gan.discriminator.store
    np.save(os.path.join(out_dir, f"d_grad_{rank}.npy"), st.grad[:st.n_train].cpu().numpy())
    np.save(os.path.join(out_dir, f"d_theta_{rank}.npy"), st.theta[:st.n_train].cpu().numpy())
    sg = gan.generator.store
    np.save(os.path.join(out_dir, f"g_grad_{rank}.npy"), sg.grad[:sg.n_train].cpu().numpy())
    np.save(os.path.join(out_dir, f"g_theta_{rank}.npy"), sg.theta[:sg.n_train].cpu().numpy())
    np.save(os.path.join(out_dir, f"g_state_{rank}.npy"), sg.state[:sg.n_state].cpu().numpy())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("arch,B_local", [("tiny", 3), ("mnist", 4)])          # the 8x8 test stack and the real MNIST stack at global batch 8
def test_two_ranks_match_single_process_global_batch(tmp_path, arch, B_local):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), arch, B_local), nprocs=world, join=True)
    gan, reals, rnd = _build(arch, B_local * world, B_local * world)
    gan.train_on_batch(reals, randomness=rnd)
    st = gan.discriminator.store
    ref_g, ref_t = st.grad[:st.n_train].cpu().numpy(), st.theta[:st.n_train].cpu().numpy()
    g0, g1 = np.load(tmp_path / "d_grad_0.npy"), np.load(tmp_path / "d_grad_1.npy")
    np.testing.assert_array_equal(g0, g1)                                   # all-reduced: identical on both ranks
    np.testing.assert_allclose(g0, ref_g, rtol=2e-3, atol=2e-4 * np.abs(ref_g).max())
    t0, t1 = np.load(tmp_path / "d_theta_0.npy"), np.load(tmp_path / "d_theta_1.npy")
    np.testing.assert_array_equal(t0, t1)                                   # replicas stay in lock step
    np.testing.assert_allclose(t0, ref_t, rtol=1e-3, atol=2e-4)
    # generator: SyncBN makes the 2-replica step equal to the single-device step at the global batch (SURVEY.md 8e iii)
    sg = gan.generator.store
    rg = sg.grad[:sg.n_train].cpu().numpy()
    gg0, gg1 = np.load(tmp_path / "g_grad_0.npy"), np.load(tmp_path / "g_grad_1.npy")
    np.testing.assert_array_equal(gg0, gg1)
    np.testing.assert_allclose(gg0, rg, rtol=5e-3, atol=5e-4 * np.abs(rg).max())
    np.testing.assert_allclose(np.load(tmp_path / "g_theta_0.npy"), sg.theta[:sg.n_train].cpu().numpy(), rtol=1e-3, atol=3e-4)
    s0, s1 = np.load(tmp_path / "g_state_0.npy"), np.load(tmp_path / "g_state_1.npy")
    np.testing.assert_array_equal(s0, s1)
    np.testing.assert_allclose(s0, sg.state[:sg.n_state].cpu().numpy(), rtol=1e-4, atol=1e-6)      # BN moving statistics


def test_c3_global_batch_2048_sharded_over_four_ranks(tmp_path):
    """BASELINE.json configs[2] (C3) as far as one card goes: the 64x64 stack at the GLOBAL batch of 2048, sharded over 4 ranks
    of 512 (gloo rendezvous, gradients staged through the host; the box allows 6 processes on its card, the real run is 8 x 256
    over RCCL) against the single-process step at batch 2048: the all-reduced critic and generator gradients (SyncBN statistics
    exchanged in the forward AND the backward of the G-step, the [B]-vector quirk scaled by the global batch, the penalty
    averaged over it) equal the global-batch gradients -- compared as whole flat vectors by relative L2 error (both sides are
    the HIP path on their own LeakyReLU branches: single elements differ by branch flips, see tests/test_step_gpu.py) -- and
    the replicas end the step bit-identical."""
    from helpers import rel_l2, cosine
    world, arch, B_local = 4, "celeba64", 512
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), arch, B_local), nprocs=world, join=True)
    gan, reals, rnd = _build(arch, B_local * world, B_local * world)
    gan.train_on_batch(reals, randomness=rnd)
    for tag, store in (("d", gan.discriminator.store), ("g", gan.generator.store)):
        ref = store.grad[:store.n_train].cpu().numpy()
        got = [np.load(tmp_path / f"{tag}_grad_{r}.npy") for r in range(world)]
        for r in range(1, world):
            np.testing.assert_array_equal(got[0], got[r])                   # all-reduced: identical on every rank
        l2, c = rel_l2(got[0], ref), 1.0 - cosine(got[0], ref)
        print(f"C3 global batch 2048 over {world} ranks, {tag} gradients: rel-L2 {l2:.1e}, 1-cos {c:.0e}")
        assert l2 <= (1e-3 if tag == "g" else 1e-4) and c <= 1e-6, (tag, l2, c)     # measured 2.6e-5 / 4.5e-7
        thetas = [np.load(tmp_path / f"{tag}_theta_{r}.npy") for r in range(world)]
        for r in range(1, world):
            np.testing.assert_array_equal(thetas[0], thetas[r])             # replicas stay in lock step
    s = [np.load(tmp_path / f"g_state_{r}.npy") for r in range(world)]
    for r in range(1, world):
        np.testing.assert_array_equal(s[0], s[r])
    sg = gan.generator.store
    np.testing.assert_allclose(s[0], sg.state[:sg.n_state].cpu().numpy(), rtol=2e-4, atol=2e-6)      # BN moving statistics of the global batch


def test_abi_communicator_single_rank():
    """include/bgan.h's own RCCL communicator (bg_comm_*): with one rank the SUM all-reduce is the identity, runs on the
    stream it is given and leaves the buffer untouched; the handle is created and destroyed without touching torch.distributed.
    (Two RCCL ranks cannot share the single card of the test box; the N>1 wiring is covered by the gloo tests.)"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from blurred_gan_amd import dist
    torch.cuda.set_device(0)
    comm = dist.AbiComm()
    x = torch.randn(1 << 20, device="cuda")
    ref = x.clone()
    y = x * 2.0                      # enqueue work the collective must be ordered after
    w = comm.all_reduce_async(y)
    w.wait()
    z = y + 1.0
    torch.cuda.synchronize()
    assert torch.equal(y, ref * 2.0) and torch.equal(z, ref * 2.0 + 1.0)
    comm.close()


def test_abi_communicator_argument_errors():
    import ctypes as C
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from blurred_gan_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.bg_comm_init(C.byref(h), 2, 2, b"\0" * 128) == -1          # rank out of range
    assert lib.bg_comm_init(None, 0, 1, b"\0" * 128) == -6
    assert lib.bg_allreduce_sum_f32(None, None, 4, None) == -6
    assert lib.bg_comm_destroy(None) == 0


def _rccl_worker(rank, port, out_dir):
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      BGAN_DP_FORCE_COLLECTIVES="1")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from blurred_gan_amd import dist
    torch.cuda.set_device(0)
    torch.distributed.init_process_group(backend="nccl")             # RCCL, one rank
    assert dist.collectives_active()
    gan, reals, rnd = _build("tiny", 6, 6)
    for _ in range(2):
        gan.train_on_batch(reals, randomness=rnd)
    torch.cuda.synchronize()
    for tag, net in (("d", gan.discriminator), ("g", gan.generator)):
        st = net.store
        np.save(os.path.join(out_dir, f"rccl_{tag}_theta.npy"), st.theta[:st.n_train].cpu().numpy())
        np.save(os.path.join(out_dir, f"rccl_{tag}_grad.npy"), st.grad[:st.n_train].cpu().numpy())
    assert dist.max_over_ranks(3.5) == 3.5
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_rccl_path_single_rank_matches_plain_step(tmp_path):
    """The production collective path on real RCCL (backend "nccl"), with the one rank the test box has: bucketed asynchronous
    gradient all-reduces overlapped with the backward, SyncBN statistics exchange, wait-before-Adam.  Every collective is the
    identity here, so two training steps must land on the weights of the plain single-process steps."""
    mp.spawn(_rccl_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    gan, reals, rnd = _build("tiny", 6, 6)
    for _ in range(2):
        gan.train_on_batch(reals, randomness=rnd)
    for tag, net in (("d", gan.discriminator), ("g", gan.generator)):
        st = net.store
        ref_t, ref_g = st.theta[:st.n_train].cpu().numpy(), st.grad[:st.n_train].cpu().numpy()
        got_t, got_g = np.load(tmp_path / f"rccl_{tag}_theta.npy"), np.load(tmp_path / f"rccl_{tag}_grad.npy")
        np.testing.assert_allclose(got_g, ref_g, rtol=2e-3, atol=2e-4 * np.abs(ref_g).max())
        np.testing.assert_allclose(got_t, ref_t, rtol=1e-3, atol=2e-4)


def _demo_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      BGAN_DIST_BACKEND="gloo", BGAN_DIST_SHARE_DEVICES="1")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import demo_mnist
    gan = demo_mnist.main(["--batch_size", "4", "--epochs", "1", "--max_batches", "3", "--results_dir", os.path.join(out_dir, "results")])
    assert int(gan.n_batches) == 3 and gan.hparams.global_batch_size == 8
    for tag, net in (("d", gan.discriminator), ("g", gan.generator)):
        np.save(os.path.join(out_dir, f"demo_{tag}_theta_{rank}.npy"), net.store.theta.cpu().numpy())
    torch.distributed.destroy_process_group()


def test_demo_under_two_ranks_one_run_directory_and_lockstep_replicas(tmp_path):
    """The demo's fit() as two data-parallel ranks (gloo rendezvous, both on the test box's one card): the global batch is
    written into the hyper-parameters, ONE run directory exists (made by rank 0, its name broadcast), only rank 0 wrote files,
    and the replicas end with identical weights."""
    import json
    mp.spawn(_demo_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    runs = os.listdir(tmp_path / "results")
    assert runs == ["01-mnist"], runs
    hp = json.load(open(tmp_path / "results" / "01-mnist" / "hyper_parameters.json"))
    assert hp["global_batch_size"] == 8 and hp["batch_size"] == 4
    for tag in ("d", "g"):
        np.testing.assert_array_equal(np.load(tmp_path / f"demo_{tag}_theta_0.npy"), np.load(tmp_path / f"demo_{tag}_theta_1.npy"))


def test_local_rank_beyond_visible_devices_is_an_error(monkeypatch):
    """More local ranks than GPUs must fail loudly (two RCCL ranks on one card hang inside the first collective); sharing a card
    is an explicit rehearsal switch."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from blurred_gan_amd import dist
    n = torch.cuda.device_count()
    monkeypatch.setenv("LOCAL_RANK", str(n))
    monkeypatch.delenv("BGAN_DIST_SHARE_DEVICES", raising=False)
    monkeypatch.delenv("BGAN_DIST_BACKEND", raising=False)
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):       # the box itself may mask its one card
        monkeypatch.delenv(v, raising=False)
    with pytest.raises(RuntimeError, match="LOCAL_RANK"):
        dist.local_rank()
    monkeypatch.setenv("BGAN_DIST_SHARE_DEVICES", "1")           # sharing a card under RCCL hangs: the switch alone is refused
    with pytest.raises(RuntimeError, match="LOCAL_RANK"):
        dist.local_rank()
    monkeypatch.setenv("BGAN_DIST_BACKEND", "gloo")              # the rehearsal mode: gloo, ranks round-robin over the cards
    assert dist.local_rank() == 0
    monkeypatch.delenv("BGAN_DIST_BACKEND", raising=False)
    monkeypatch.delenv("BGAN_DIST_SHARE_DEVICES", raising=False)
    if n == 1:                                                   # one masked device per rank (SLURM --gpus-per-task=1): device 0
        monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "3")
        assert dist.local_rank() == 0
        monkeypatch.delenv("ROCR_VISIBLE_DEVICES")
    monkeypatch.setenv("LOCAL_RANK", "0")
    assert dist.local_rank() == 0


def test_bench_two_ranks_strong_scaling_smoke():
    """bench.py --gpus 2 --strong (the launcher path of the contract: bench.py starts torch.distributed.run itself), two ranks
    sharing the box's one card over gloo (the rehearsal mode): ONE JSON line, n_gpus 2, strong scaling, the global batch split."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BGAN_DIST_BACKEND="gloo", BGAN_DIST_SHARE_DEVICES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--strong", "--arch", "mnist", "--batch", "16",
                        "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-profile"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["global_batch"] == 16 and out["value"] > 0
    assert out["config"]["parallelism"] == "dp2" and "8/GPU" in out["config"]["workload"]
    # the collective layer's own account of the group: two peers counted by an all-reduce through the step's
    # communicator; both ranks rehearse on the box's one card, and the line says so
    dp = out["dp"]
    assert dp["backend"] == "gloo" and dp["collective_nranks"] == 2 and dp["world_size_env"] == 2 and len(dp["devices"]) == 2
    assert dp["distinct_devices"] == 1 and dp["devices"][0] == dp["devices"][1] and dp["devices"][0]
    assert "rccl_nranks" not in dp                         # only an RCCL group may claim RCCL peers


def test_bench_one_rank_rccl_line_carries_the_communicators_own_count():
    """bench.py --gpus 1 with every collective forced through RCCL (BGAN_DP_FORCE_COLLECTIVES=1, the N = 1 point of the scaling
    run that can be cross-checked with the plain bench line): backend nccl, rccl_nranks 1 from an all-reduce, the card's uuid;
    through the C ABI's communicator also ncclCommCount / ncclCommUserRank (bg_comm_query)."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for route in ("torch", "abi"):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, BGAN_DP_FORCE_COLLECTIVES="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", BGAN_DP_COLLECTIVE=route)
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--arch", "mnist", "--batch", "16",
                            "--steps", "3", "--warmup", "3", "--no-cpu-baseline", "--no-profile"], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
        dp = out["dp"]
        assert dp["backend"].startswith("nccl") and dp["rccl_nranks"] == 1 and dp["collective_route"] == route and dp["devices"][0]
        if route == "abi":
            assert dp["abi_comm"] == {"nranks": 1, "rank": 0}


def _replay_worker(rank, world, port, out_dir, backend, replay, tag):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if world == 1:
        os.environ["BGAN_DP_FORCE_COLLECTIVES"] = "1"
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import blurred_gan_amd as bg
    from blurred_gan_amd import dist, models
    torch.cuda.set_device(0)
    torch.distributed.init_process_group(backend=backend)
    assert dist.collectives_active()
    arch, B = "mnist", 4
    bg.set_seed(17)
    gen, disc = models.DCGANGenerator(arch=arch), models.DCGANDiscriminator(arch=arch)
    hp = bg.BlurredWGANGP.HyperParameters(initial_blur_std=1.30, global_batch_size=B * world, batch_size=B)
    gan = bg.BlurredWGANGP(gen, disc, hp, bg.TrainingConfig(log_dir="/tmp/bg_dp_logs"), step_replay=replay)
    g = torch.Generator().manual_seed(40 + rank)
    mets = []
    for i in range(6):
        gan.std.assign(1.30 - 0.01 * i)                      # 9 taps throughout: one D and one G program
        mets.append(gan.train_on_batch((torch.rand(B, 28, 28, 1, generator=g) * 2 - 1).cuda()))
    torch.cuda.synchronize()
    st = gan._programs.stats
    assert (st["replayed"] >= 6) if replay else (st["replayed"] == 0), st
    if replay:       # the collectives of the recorded steps are host actions of their programs
        assert all(len(e.actions) >= 2 for e in gan._programs.entries.values() if hasattr(e, "actions"))
    for name, net in (("d", gan.discriminator), ("g", gan.generator)):
        s = net.store
        np.save(os.path.join(out_dir, f"{tag}_{name}_theta_{rank}.npy"), s.theta.cpu().numpy())
        np.save(os.path.join(out_dir, f"{tag}_{name}_state_{rank}.npy"), s.state.cpu().numpy())
        np.save(os.path.join(out_dir, f"{tag}_{name}_v_{rank}.npy"), s.v.cpu().numpy())
    np.save(os.path.join(out_dir, f"{tag}_metrics_{rank}.npy"), np.asarray(mets, np.float64))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("backend,world", [("gloo", 2), ("nccl", 1)])
def test_replayed_steps_issue_the_same_collectives_as_eager_steps(tmp_path, backend, world):
    """Step programs under data parallelism (include/bgan.h: collectives are the host's, the replay is split where they were
    issued while recording): six steps of the MNIST stack with the build's own RNG and a changing sigma, as two gloo ranks on
    the box's one card and as a ONE-rank RCCL group with every collective forced (bucketed asynchronous gradient all-reduces,
    SyncBN exchanges in the forward and, overlapped, in the backward, wait-before-Adam) -- replay on against replay off:
    weights, BatchNorm statistics, Adam's second moments and every step's metrics bit-identical, replicas in lock step."""
    for replay, tag in ((False, "eager"), (True, "replay")):
        mp.spawn(_replay_worker, args=(world, _free_port(), str(tmp_path), backend, replay, tag), nprocs=world, join=True)
    for r in range(world):
        for name in ("d", "g"):
            for what in ("theta", "state", "v"):
                a, b = np.load(tmp_path / f"eager_{name}_{what}_{r}.npy"), np.load(tmp_path / f"replay_{name}_{what}_{r}.npy")
                np.testing.assert_array_equal(a, b)
                if r:
                    np.testing.assert_array_equal(b, np.load(tmp_path / f"replay_{name}_{what}_0.npy"))
        np.testing.assert_array_equal(np.load(tmp_path / f"eager_metrics_{r}.npy"), np.load(tmp_path / f"replay_metrics_{r}.npy"))
