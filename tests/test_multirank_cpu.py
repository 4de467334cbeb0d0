"""World-size-2 `gloo` test of the point-sharded multi-GPU path on CPU.

The HIP engine cannot run here, so the compute on each rank is the oracle (test infrastructure); what is under test
is the product's host logic: WindowProblem.shard() / shard_bounds() and the exchange protocol -- the per-rank partial
sums of the reduced camera system ingredients [U | g_c | cost] are all-reduced (SUM) and must reproduce the
single-rank system to 1e-12, while every point block stays rank-local."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from oracle import oracle
    from photobundle_amd import synthetic
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = synthetic.make_window(n_frames=4, n_points=101, radius=2, size=(120, 160), K=(200.0, 200.0, 80.0, 60.0),
                              visibility="causal", huber=0.05, seed_offset=7)
    sh = p.shard(rank, world)
    lin = oracle.linearize(sh)
    packed = np.concatenate([lin["U"].reshape(-1), lin["grad_cams"].reshape(-1), [lin["cost"]],
                             [float(sh.n_obs)], [float(sh.n_points)]])
    t = torch.from_numpy(packed.copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    gmax = torch.tensor([np.abs(lin["grad_pts"]).max()], dtype=torch.float64)
    dist.all_reduce(gmax, op=dist.ReduceOp.MAX)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), reduced=t.numpy(), gmax=gmax.numpy(),
             V=lin["V"], range=np.array(sh.meta["point_range"]))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_partial_sums_reproduce_the_full_system(tmp_path):
    import torch.multiprocessing as mp
    from oracle import oracle
    from photobundle_amd import synthetic
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    p = synthetic.make_window(n_frames=4, n_points=101, radius=2, size=(120, 160), K=(200.0, 200.0, 80.0, 60.0),
                              visibility="causal", huber=0.05, seed_offset=7)
    full = oracle.linearize(p)
    ref = np.concatenate([full["U"].reshape(-1), full["grad_cams"].reshape(-1), [full["cost"]],
                          [float(p.n_obs)], [float(p.n_points)]])
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["reduced"], r1["reduced"])            # every rank holds the same reduced buffer
    scale = np.abs(ref).max()
    assert np.abs(r0["reduced"] - ref).max() <= 1e-12 * scale
    assert np.isclose(r0["gmax"][0], np.abs(full["grad_pts"]).max(), rtol=0, atol=0)
    # point blocks never leave their rank: the shards tile the point range exactly
    (a0, b0), (a1, b1) = r0["range"], r1["range"]
    assert a0 == 0 and b0 == a1 and b1 == p.n_points
    assert np.array_equal(np.concatenate([r0["V"], r1["V"]]), full["V"])


def test_shard_bounds_balance_and_cover():
    from photobundle_amd.problem import shard_bounds
    rng = np.random.default_rng(3)
    counts = rng.integers(1, 9, 1000)
    obs_point = np.repeat(np.arange(1000), counts)
    for world in (1, 2, 3, 4, 8):
        cuts = [shard_bounds(obs_point, 1000, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == 1000
        for (a, b), (c, d) in zip(cuts[:-1], cuts[1:]):
            assert b == c
        per = [np.sum((obs_point >= a) & (obs_point < b)) for a, b in cuts]
        assert max(per) - min(per) <= 16, per


def test_shard_of_empty_tail():
    from photobundle_amd.problem import shard_bounds
    obs_point = np.array([0, 0, 1], dtype=np.int32)
    cuts = [shard_bounds(obs_point, 2, r, 4) for r in range(4)]
    assert cuts[0][0] == 0 and cuts[-1][1] == 2
    assert sum(b - a for a, b in cuts) == 2
