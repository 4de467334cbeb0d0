"""-m gpu: the engine's multi-rank path on ONE GPU.  Two processes share the device, each owns a point shard; the
all-reduces go through the host-staged callback transport over gloo, so the packing / reduction protocol of the
product code is exercised end to end (the RCCL transport differs only in who moves the bytes)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make():
    from photobundle_amd import synthetic
    return synthetic.make_window(n_frames=4, n_points=300, radius=2, size=(120, 160), K=(200.0, 200.0, 80.0, 60.0),
                                 visibility="causal", seed_offset=4)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from photobundle_amd.engine import default_solver_options
    from gpu_util import make_engine
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = _make()
    sh = p.shard(rank, world)
    e = make_engine(sh, keep_reduced_system=True)

    def allreduce(a, op):
        t = torch.from_numpy(a)
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX)

    e.comm_init_callback(allreduce, rank, world)
    res = e.solve(default_solver_options(max_num_iterations=8))
    S, rhs = e.reduced_system()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), cams=res["cams"], xyz=res["xyz"],
             costs=np.array([i["cost"] for i in res["iterations"]]), ok=np.array([i["step_is_successful"] for i in res["iterations"]]),
             S=S, rhs=rhs, nres=res["num_residuals"], range=np.array(sh.meta["point_range"]))
    e.close()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_match_one_rank(tmp_path):
    import torch.multiprocessing as mp
    from photobundle_amd.engine import default_solver_options
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gpu_util import make_engine
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.get_context("spawn")
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    p = _make()
    with make_engine(p) as e:
        ref = e.solve(default_solver_options(max_num_iterations=8))
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    # replicated state is bit-identical across ranks (same reduced system -> same camera step on every rank)
    assert np.array_equal(r0["cams"], r1["cams"]) and np.array_equal(r0["S"], r1["S"])
    assert np.array_equal(r0["ok"], r1["ok"]) and np.array_equal(r0["costs"], r1["costs"])
    # and matches the single-rank solve up to summation order
    ref_costs = np.array([i["cost"] for i in ref["iterations"]])
    assert len(ref_costs) == len(r0["costs"])
    assert np.allclose(r0["costs"], ref_costs, rtol=1e-9)
    assert np.array_equal(r0["ok"], np.array([i["step_is_successful"] for i in ref["iterations"]]))
    assert np.abs(r0["cams"] - ref["cams"]).max() <= 1e-8
    xyz = np.concatenate([r0["xyz"], r1["xyz"]])
    assert np.abs(xyz - ref["xyz"]).max() <= 1e-6
    assert int(r0["nres"]) == ref["num_residuals"]


def test_rccl_transport_initialises_single_rank():
    """ncclCommInitRank through the dlopen'ed librccl on the real device (world = 1: no collective is issued)."""
    from photobundle_amd.engine import Engine, default_solver_options
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gpu_util import make_engine
    p = _make()
    with make_engine(p) as e:
        ref = e.solve(default_solver_options(max_num_iterations=4))
    with make_engine(p) as e:
        e.comm_init_rccl(Engine.comm_unique_id(), 0, 1)
        res = e.solve(default_solver_options(max_num_iterations=4))
    assert np.array_equal(res["cams"], ref["cams"]) and res["final_cost"] == ref["final_cost"]


def test_rccl_collectives_on_the_device_stream(tmp_path):
    """PBA_FORCE_MULTI=1 runs the multi-rank code path of both drivers (packed reduced system + one scalar exchange,
    k_decide / k_publish after the collectives) with real ncclAllReduce calls on the engine's stream, world = 1."""
    import subprocess
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from photobundle_amd.engine import Engine, default_solver_options\n"
        "from gpu_util import make_engine\n"
        "import test_gpu_multirank as t\n"
        "p = t._make()\n"
        "import os\n"
        "e = make_engine(p); e.comm_init_rccl(Engine.comm_unique_id(), 0, 1)\n"
        "if os.environ.get('PBA_TEST_PEER') == '1': assert e.comm_enable_peer_exchange() == 'rccl+peer'\n"
        "res = e.solve(default_solver_options(max_num_iterations=8))\n"
        "e.load(p); conv = e.solve(default_solver_options(max_num_iterations=200, function_tolerance=1e-4)); e.close()\n"
        "np.savez(%r + '/out_' + sys.argv[1] + '.npz', cams=res['cams'], xyz=res['xyz'], costs=np.array([i['cost'] for i in res['iterations']]),"
        " g=np.array([i['gradient_max_norm'] for i in res['iterations']]), conv_costs=np.array([i['cost'] for i in conv['iterations']]),"
        " conv_type=conv['termination_type'])\n"
    ) % (ROOT, os.path.join(ROOT, "tests"), str(tmp_path))
    outs = {}
    for tag, env in [("plain", {}), ("multi_async", {"PBA_FORCE_MULTI": "1"}), ("multi_sync", {"PBA_FORCE_MULTI": "1", "PBA_ASYNC": "0"}),
                     ("peer_async", {"PBA_FORCE_MULTI": "1", "PBA_TEST_PEER": "1"}), ("peer_sync", {"PBA_FORCE_MULTI": "1", "PBA_TEST_PEER": "1", "PBA_ASYNC": "0"})]:
        r = subprocess.run([sys.executable, "-c", code, tag], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[tag] = np.load(tmp_path / ("out_%s.npz" % tag))
    # same driver, same kernels, collectives are identities at world = 1: bit-identical
    assert np.array_equal(outs["multi_async"]["costs"], outs["plain"]["costs"])
    assert np.array_equal(outs["multi_async"]["cams"], outs["plain"]["cams"]) and np.array_equal(outs["multi_async"]["xyz"], outs["plain"]["xyz"])
    # the peer exchange at world = 1 (this rank's mailbox is its only peer): k_peer_allreduce instead of ncclAllReduce, same bits
    assert np.array_equal(outs["peer_async"]["costs"], outs["plain"]["costs"]) and np.array_equal(outs["peer_async"]["cams"], outs["plain"]["cams"])
    assert np.array_equal(outs["peer_async"]["conv_costs"], outs["plain"]["conv_costs"]) and int(outs["peer_async"]["conv_type"]) == 0
    assert np.array_equal(outs["peer_sync"]["costs"], outs["multi_sync"]["costs"]) and np.array_equal(outs["peer_sync"]["cams"], outs["multi_sync"]["cams"])
    # the synchronous driver linearises the first point on a different workgroup grid (different summation order)
    assert np.allclose(outs["multi_sync"]["costs"], outs["plain"]["costs"], rtol=1e-11)
    assert np.abs(outs["multi_sync"]["cams"] - outs["plain"]["cams"]).max() <= 1e-9
    for tag in ("multi_async", "multi_sync"):
        assert np.allclose(outs[tag]["g"], outs["plain"]["g"], rtol=1e-9)
    # tolerance-terminated solve: the multi-rank driver stops a fixed number of steps after the terminating one (every
    # rank must enqueue the same number of collectives), with the same trace as the single-rank driver
    assert int(outs["plain"]["conv_type"]) == 0 and len(outs["plain"]["conv_costs"]) < 200
    assert np.array_equal(outs["multi_async"]["conv_costs"], outs["plain"]["conv_costs"])
    assert int(outs["multi_async"]["conv_type"]) == 0


# ---- two DISTINCT devices over RCCL with the asynchronous driver (runs wherever >= 2 GPUs are visible) -------------
def _worker_rccl(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["PBA_WAIT_TIMEOUT_S"] = "60"          # a mismatch in enqueued collectives becomes PBA_ERR_COMM, not a hang
    import torch
    import torch.distributed as dist
    from photobundle_amd.engine import Engine, default_solver_options
    from gpu_util import make_engine
    torch.cuda.set_device(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    uid = [Engine.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    p = _make()
    sh = p.shard(rank, world)
    e = make_engine(sh, device=rank, keep_reduced_system=True)
    e.comm_init_rccl(uid[0], rank, world)
    out = {}
    # (1) fixed iteration count, (2) tolerance-terminated (every rank must stop after the same number of enqueued steps)
    res = e.solve(default_solver_options(max_num_iterations=8))
    out.update(cams=res["cams"], xyz=res["xyz"], costs=np.array([i["cost"] for i in res["iterations"]]),
               ok=np.array([i["step_is_successful"] for i in res["iterations"]]), nres=res["num_residuals"])
    e.load(sh)
    conv = e.solve(default_solver_options(max_num_iterations=200, function_tolerance=1e-4))
    out.update(conv_costs=np.array([i["cost"] for i in conv["iterations"]]), conv_type=conv["termination_type"],
               conv_cams=conv["cams"])
    np.savez(os.path.join(out_dir, "rccl_rank%d.npz" % rank), **out)
    e.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_devices_over_rccl_async_driver(tmp_path):
    """Two processes, two distinct GPUs, device-direct ncclAllReduce on the engines' streams, asynchronous driver (the
    default at world > 1): fixed-length and tolerance-terminated solves against the single-rank run."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    import torch.multiprocessing as mp
    from photobundle_amd.engine import default_solver_options
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gpu_util import make_engine
    world = 2
    port = 29500 + ((os.getpid() + 31) % 2000)
    mp.spawn(_worker_rccl, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    p = _make()
    with make_engine(p) as e:
        ref = e.solve(default_solver_options(max_num_iterations=8))
        e.load(p)
        ref_conv = e.solve(default_solver_options(max_num_iterations=200, function_tolerance=1e-4))
    r0 = np.load(tmp_path / "rccl_rank0.npz")
    r1 = np.load(tmp_path / "rccl_rank1.npz")
    assert np.array_equal(r0["cams"], r1["cams"]) and np.array_equal(r0["costs"], r1["costs"]) and np.array_equal(r0["ok"], r1["ok"])
    ref_costs = np.array([i["cost"] for i in ref["iterations"]])
    assert len(ref_costs) == len(r0["costs"]) and np.allclose(r0["costs"], ref_costs, rtol=1e-9)
    assert np.abs(r0["cams"] - ref["cams"]).max() <= 1e-8
    assert np.abs(np.concatenate([r0["xyz"], r1["xyz"]]) - ref["xyz"]).max() <= 1e-6
    assert int(r0["nres"]) == ref["num_residuals"]
    # tolerance-terminated: same trace on both ranks and as the single-rank driver
    assert np.array_equal(r0["conv_costs"], r1["conv_costs"]) and int(r0["conv_type"]) == int(r1["conv_type"]) == 0
    rc = np.array([i["cost"] for i in ref_conv["iterations"]])
    # (the shards sum in a different order than one rank does: a tolerance test within 1e-12 of its threshold may fire
    # one iteration apart)
    n = min(len(rc), len(r0["conv_costs"]))
    assert abs(len(rc) - len(r0["conv_costs"])) <= 1 and np.allclose(r0["conv_costs"][:n], rc[:n], rtol=1e-8)
    if len(rc) == len(r0["conv_costs"]):
        assert np.abs(r0["conv_cams"] - ref_conv["cams"]).max() <= 1e-6


# ---- device-side peer exchange (hipIpc* mailboxes) with the asynchronous driver: two processes on ONE device ------------
def _worker_peer(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["PBA_WAIT_TIMEOUT_S"] = "30"          # a stuck exchange becomes PBA_ERR_COMM, not a hang
    import torch
    import torch.distributed as dist
    from photobundle_amd.engine import default_solver_options
    from gpu_util import make_engine
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = _make()
    sh = p.shard(rank, world)
    e = make_engine(sh, keep_reduced_system=True)

    def allreduce(a, op):
        t = torch.from_numpy(a)
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX)

    e.comm_init_callback(allreduce, rank, world)
    transport = e.comm_enable_peer_exchange()
    out = dict(transport=np.array(transport))
    if transport == "callback+peer":
        res = e.solve(default_solver_options(max_num_iterations=8))
        out.update(cams=res["cams"], xyz=res["xyz"], costs=np.array([i["cost"] for i in res["iterations"]]),
                   ok=np.array([i["step_is_successful"] for i in res["iterations"]]), nres=res["num_residuals"])
        S, rhs = e.reduced_system()
        out.update(S=S, rhs=rhs)
        e.load(sh)
        conv = e.solve(default_solver_options(max_num_iterations=200, function_tolerance=1e-4))
        out.update(conv_costs=np.array([i["cost"] for i in conv["iterations"]]), conv_type=conv["termination_type"], conv_cams=conv["cams"])
    np.savez(os.path.join(out_dir, "peer_rank%d.npz" % rank), **out)
    dist.barrier()           # nobody frees its mailbox while a peer may still read it
    e.close()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_peer_exchange_async_driver(tmp_path):
    """Two processes share the device; the gloo callback transport only bootstraps (IPC handles, set-up reductions), every
    per-step exchange is a flag-and-slot read of the peer's mailbox inside k_peer_allreduce on the engine's stream, so
    the ASYNCHRONOUS driver runs (the host-staged transport cannot).  Same assertions as the host-staged two-rank test,
    plus a tolerance-terminated solve (every rank must stop after the same number of enqueued steps)."""
    import torch.multiprocessing as mp
    from photobundle_amd.engine import default_solver_options
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gpu_util import make_engine
    world = 2
    port = 29500 + ((os.getpid() + 77) % 2000)
    mp.spawn(_worker_peer, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "peer_rank0.npz")
    r1 = np.load(tmp_path / "peer_rank1.npz")
    assert str(r0["transport"]) == str(r1["transport"])
    if str(r0["transport"]) != "callback+peer":
        pytest.skip("IPC mapping of fine-grained device memory is not available here: the engine stayed on %s" % r0["transport"])
    p = _make()
    with make_engine(p) as e:
        ref = e.solve(default_solver_options(max_num_iterations=8))
        e.load(p)
        ref_conv = e.solve(default_solver_options(max_num_iterations=200, function_tolerance=1e-4))
    assert np.array_equal(r0["cams"], r1["cams"]) and np.array_equal(r0["S"], r1["S"])           # rank-ordered sums: bit-identical replicas
    assert np.array_equal(r0["ok"], r1["ok"]) and np.array_equal(r0["costs"], r1["costs"])
    ref_costs = np.array([i["cost"] for i in ref["iterations"]])
    assert len(ref_costs) == len(r0["costs"]) and np.allclose(r0["costs"], ref_costs, rtol=1e-9)
    assert np.array_equal(r0["ok"], np.array([i["step_is_successful"] for i in ref["iterations"]]))
    assert np.abs(r0["cams"] - ref["cams"]).max() <= 1e-8
    assert np.abs(np.concatenate([r0["xyz"], r1["xyz"]]) - ref["xyz"]).max() <= 1e-6
    assert int(r0["nres"]) == ref["num_residuals"]
    assert np.array_equal(r0["conv_costs"], r1["conv_costs"]) and int(r0["conv_type"]) == int(r1["conv_type"]) == 0
    rc = np.array([i["cost"] for i in ref_conv["iterations"]])
    n = min(len(rc), len(r0["conv_costs"]))
    assert abs(len(rc) - len(r0["conv_costs"])) <= 1 and np.allclose(r0["conv_costs"][:n], rc[:n], rtol=1e-8)


# ---- EIGHT ranks, sixteen frames (the sharding of BASELINE configs[3]) on ONE device ------------------------------------------
def _make16():
    from photobundle_amd import synthetic
    return synthetic.make_window(n_frames=16, n_points=960, radius=2, size=(120, 200), K=(250.0, 250.0, 100.0, 60.0),
                                 visibility="causal", seed_offset=6)


def _worker_peer8(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["PBA_WAIT_TIMEOUT_S"] = "60"
    import torch
    import torch.distributed as dist
    from photobundle_amd.engine import default_solver_options
    from gpu_util import make_engine
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = _make16()
    sh = p.shard(rank, world)
    e = make_engine(sh, keep_reduced_system=True)

    def allreduce(a, op):
        t = torch.from_numpy(a)
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX)

    e.comm_init_callback(allreduce, rank, world)
    transport = e.comm_enable_peer_exchange()
    res = e.solve(default_solver_options(max_num_iterations=5))
    S, rhs = e.reduced_system()
    np.savez(os.path.join(out_dir, "p8_rank%d.npz" % rank), transport=np.array(transport), cams=res["cams"], xyz=res["xyz"],
             costs=np.array([i["cost"] for i in res["iterations"]]), ok=np.array([i["step_is_successful"] for i in res["iterations"]]),
             S=S, rhs=rhs, nres=res["num_residuals"], n_points=sh.n_points)
    dist.barrier()           # nobody frees its mailbox while a peer may still read it
    e.close()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_eight_ranks_sixteen_frames_on_one_device(tmp_path):
    """The partitioning of BASELINE configs[3] -- a 16-frame window point-sharded over EIGHT ranks, the 90 x 90 reduced system summed
    in rank order and solved redundantly on every rank (k_reduce_final -> mailbox -> k_solve_blocked<512> with its peer wait; host-
    staged all-reduce where IPC mapping is unavailable) -- with real kernels and the asynchronous driver, all eight processes on the
    one GPU of the box.  What is NOT covered here is the xGMI hop between devices (no multi-GPU box): everything else of the 8-way
    path is.  Replicas must be bit-identical, the solve must match one rank up to the summation order of the shards."""
    import torch.multiprocessing as mp
    from photobundle_amd.engine import default_solver_options
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gpu_util import make_engine
    world = 8
    port = 29500 + ((os.getpid() + 131) % 2000)
    mp.spawn(_worker_peer8, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / ("p8_rank%d.npz" % k)) for k in range(world)]
    assert len({str(x["transport"]) for x in r}) == 1
    print("transport of the eight ranks:", str(r[0]["transport"]))
    p = _make16()
    assert sum(int(x["n_points"]) for x in r) == p.n_points and all(int(x["n_points"]) > 0 for x in r)
    with make_engine(p) as e:
        ref = e.solve(default_solver_options(max_num_iterations=5))
    assert r[0]["S"].shape == (90, 90)
    for x in r[1:]:
        assert np.array_equal(x["cams"], r[0]["cams"]) and np.array_equal(x["S"], r[0]["S"]) and np.array_equal(x["rhs"], r[0]["rhs"])
        assert np.array_equal(x["costs"], r[0]["costs"]) and np.array_equal(x["ok"], r[0]["ok"])
    ref_costs = np.array([i["cost"] for i in ref["iterations"]])
    # (five iterations: eight shards sum in another order than one rank, and on this poorly initialised 16-frame window the first
    # float-rounded sample position flips at the sixth -- 1.6e-10 of the cost there, 2e-9 one iteration later; first run of the test)
    print("costs, eight ranks vs one:", np.abs(r[0]["costs"] / ref_costs - 1.0))
    # r6: the bar follows the amplification it documents -- 1e-9 while the float-rounded sample positions of the two summation orders
    # agree (iterations 0-4), 1e-8 on the iteration behind the first flip.  (The round-6 build, whose camera steps differ from round 5's in
    # the 13th digit -- recompiled kernels, other FMA contraction; identical costs on single-rank windows: profiles/r06/bits_r5_vs_r6.txt --
    # flips at the fifth iteration, 2.7e-10, and reads 1.26e-9 at the sixth; round 5 read 1.6e-10 there and 2e-9 one later.)
    assert len(ref_costs) == len(r[0]["costs"]) and np.allclose(r[0]["costs"][:5], ref_costs[:5], rtol=1e-9)
    assert np.allclose(r[0]["costs"][5:], ref_costs[5:], rtol=1e-8)
    assert np.array_equal(r[0]["ok"], np.array([i["step_is_successful"] for i in ref["iterations"]]))
    assert np.abs(r[0]["cams"] - ref["cams"]).max() <= 1e-7
    assert np.abs(np.concatenate([x["xyz"] for x in r]) - ref["xyz"]).max() <= 1e-5
    assert int(r[0]["nres"]) == ref["num_residuals"]
