"""World-size-2 `gloo` test of the product's collective transport (photobundle_amd/csrc/pba_comm.cpp, compiled into
tests/native/libcomm_probe.so unchanged) on CPU: the host-staged callback path -- the transport the engine uses when
RCCL is unavailable, and for every host-side scalar exchange -- with real data crossing real processes, no oracle."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, "tests", "native")
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int64, C.c_int32, C.c_void_p)


def _lib():
    path = os.path.join(NATIVE, "libcomm_probe.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-s", "-C", NATIVE])
    L = C.CDLL(path)
    L.probe_comm_create.restype = C.c_void_p
    L.probe_comm_error.restype = C.c_char_p
    for f in ("probe_comm_destroy", "probe_comm_init_callback", "probe_comm_allreduce_host", "probe_comm_multi",
              "probe_comm_world", "probe_comm_error"):
        getattr(L, f).argtypes = {"probe_comm_destroy": [C.c_void_p],
                                  "probe_comm_init_callback": [C.c_void_p, ALLREDUCE_FN, C.c_void_p, C.c_int, C.c_int],
                                  "probe_comm_allreduce_host": [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int],
                                  "probe_comm_multi": [C.c_void_p], "probe_comm_world": [C.c_void_p],
                                  "probe_comm_error": [C.c_void_p]}[f]
    return L


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = _lib()
    calls = []

    def tramp(ptr, n, op, ctx):
        a = np.ctypeslib.as_array(ptr, shape=(n,))
        calls.append((int(n), int(op)))
        if a[0] == -12345.0:          # the test's "transport failure" marker
            return 7
        t = torch.from_numpy(a)
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX)
        return 0
    cb = ALLREDUCE_FN(tramp)
    c = L.probe_comm_create()
    assert L.probe_comm_multi(c) == 0
    assert L.probe_comm_init_callback(c, cb, None, rank, world) == 0
    assert L.probe_comm_multi(c) == 1 and L.probe_comm_world(c) == world
    # SUM of the step scalars and rank-slotted MAX group, as the LM driver exchanges them (pba_lm.cpp / k_xchg_pack)
    rng = np.random.default_rng(100 + rank)
    v = rng.standard_normal(4 + 4 * world)
    mine = v.copy()
    buf = v.copy()
    assert L.probe_comm_allreduce_host(c, buf.ctypes.data_as(C.POINTER(C.c_double)), buf.size, 0) == 0
    mx = mine.copy()
    assert L.probe_comm_allreduce_host(c, mx.ctypes.data_as(C.POINTER(C.c_double)), mx.size, 1) == 0
    # limits and failures are reported, not swallowed
    big = np.zeros(65)
    rc_big = L.probe_comm_allreduce_host(c, big.ctypes.data_as(C.POINTER(C.c_double)), 65, 0)
    err_big = L.probe_comm_error(c).decode()
    bad = np.full(3, -12345.0)
    rc_bad = L.probe_comm_allreduce_host(c, bad.ctypes.data_as(C.POINTER(C.c_double)), 3, 0)
    err_bad = L.probe_comm_error(c).decode()
    dist.barrier()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), mine=mine, summed=buf, maxed=mx, rc=np.array([rc_big, rc_bad]),
             calls=np.array(calls))
    assert "n > 64" in err_big and "callback failed" in err_bad, (err_big, err_bad)
    L.probe_comm_destroy(c)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_callback_transport_across_two_processes(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    port = 29500 + ((os.getpid() + 777) % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / ("rank%d.npz" % k)) for k in range(world)]
    total = r[0]["mine"] + r[1]["mine"]
    for k in range(world):
        assert np.array_equal(r[k]["summed"], total)                         # bit-identical on every rank
        assert np.array_equal(r[k]["maxed"], np.maximum(r[0]["mine"], r[1]["mine"]))
        assert r[k]["rc"][0] != 0 and r[k]["rc"][1] != 0
        # the oversize request never reached the callback; the other three did, with the right (n, op)
        assert [tuple(c) for c in r[k]["calls"]] == [(12, 0), (12, 1), (3, 0)]


def test_single_rank_is_a_no_op():
    L = _lib()
    c = L.probe_comm_create()
    hit = []
    cb = ALLREDUCE_FN(lambda p, n, op, ctx: hit.append(1) or 0)
    assert L.probe_comm_init_callback(c, cb, None, 0, 1) == 0
    v = np.arange(5, dtype=np.float64)
    assert L.probe_comm_allreduce_host(c, v.ctypes.data_as(C.POINTER(C.c_double)), 5, 0) == 0
    assert not hit and np.array_equal(v, np.arange(5.0))                     # world = 1: nothing to exchange
    L.probe_comm_destroy(c)
