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
