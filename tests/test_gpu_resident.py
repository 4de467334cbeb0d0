"""-m gpu: the RESIDENT solve (photobundle_amd/csrc/pba_resident.h: the whole pba_solve as ONE cooperative launch, every workgroup
keeping its tiles' state in registers across the LM iterations) against the pipelined three-kernel path it replaces on windows that fit
one resident round of workgroups.  Both call the same device functions on the same tiles and reduce the same per-tile partials in the
same fixed order, so the bar is BIT-IDENTITY: iteration log, final cameras and points, Jacobian-pass records -- at the reference's own
operating point (config/kitti_stereo.cfg: window 5, 3x3 patches, reference src/photobundle.cc:764-876) and around it.  The pipelined
path itself is pinned to the oracle by the rest of the suite (test_gpu_parity.py, test_gpu_configs0.py, ...); one case here is also
compared with the oracle directly."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [
    # (make_window keywords, solver keywords)
    (dict(n_frames=5, n_points=2000, radius=1, size=(188, 621), K=(359.4, 359.4, 303.6, 92.6)), dict(max_num_iterations=12)),      # configs[0]-like
    (dict(n_frames=4, n_points=300, radius=2, size=(120, 160), K=(200.0, 200.0, 80.0, 60.0)), dict(max_num_iterations=8)),
    (dict(n_frames=4, n_points=300, radius=2, size=(120, 160), K=(200.0, 200.0, 80.0, 60.0)), dict(max_num_iterations=0)),          # iteration zero only
    (dict(n_frames=4, n_points=300, radius=2, size=(120, 160), K=(200.0, 200.0, 80.0, 60.0)), dict(max_num_iterations=1)),
    (dict(n_frames=8, n_points=3000, radius=2, size=(188, 621), K=(359.4, 359.4, 303.6, 92.6), visibility="causal"), dict(max_num_iterations=10)),
    (dict(n_frames=6, n_points=1500, radius=2, size=(120, 200), K=(250.0, 250.0, 100.0, 60.0), huber=0.05, gaussian=True), dict(max_num_iterations=10)),
    (dict(n_frames=5, n_points=700, radius=1, size=(120, 200), K=(250.0, 250.0, 100.0, 60.0), gaussian=True), dict(max_num_iterations=30)),     # runs into the tolerances
    (dict(n_frames=9, n_points=900, radius=2, size=(120, 200), K=(250.0, 250.0, 100.0, 60.0), rot_deg=2.0, trans=0.4), dict(max_num_iterations=12)),   # rejections
    (dict(n_frames=3, n_points=40, radius=2, size=(120, 160), K=(200.0, 200.0, 80.0, 60.0)), dict(max_num_iterations=6)),           # one workgroup
]

CODE = textwrap.dedent("""
    import json, sys
    sys.path.insert(0, %r)
    import numpy as np
    from photobundle_amd import synthetic
    from photobundle_amd.engine import Engine, default_solver_options
    cases = json.loads(sys.argv[1])
    out = []
    for wkw, skw in cases:
        wkw = dict(wkw); wkw["size"] = tuple(wkw["size"]); wkw["K"] = tuple(wkw["K"])
        p = synthetic.make_window(**wkw)
        rows, cols = wkw["size"]
        with Engine(rows, cols, p.K, p.radius, p.n_frames, huber=p.huber) as e:
            e.load(p)
            runs = []
            for rep in range(2):          # the second solve starts where the first one ended (state handed over through global memory)
                r = e.solve(default_solver_options(**skw))
                rec = e.obs_records()
                runs.append(dict(driver=e.solve_driver(), message=r["message"], n_jac=r["num_jacobian_passes"],
                                 log=[[i["cost"].hex(), i["cost_change"].hex(), i["gradient_max_norm"].hex(), i["gradient_norm"].hex(), i["step_norm"].hex(),
                                       i["relative_decrease"].hex(), i["trust_region_radius"].hex(), i["step_is_successful"], i["step_is_valid"], i["iteration"]]
                                      for i in r["iterations"]],
                                 cams=r["cams"].tobytes().hex(), xyz=r["xyz"].tobytes().hex(), rec=rec.tobytes().hex(),
                                 final_cost=r["final_cost"].hex(), initial_cost=r["initial_cost"].hex()))
            # ... and the primitive passes still work on the state a solve leaves behind
            e.linearize()
            st = e.step(1e4, init_scale=True)
            runs.append(dict(step={k: (v.hex() if isinstance(v, float) else v) for k, v in st.items()}))
        out.append(runs)
    print("RESULT" + json.dumps(out))
""" % ROOT)


def _run(resident):
    env = dict(os.environ, PBA_RESIDENT="1" if resident else "0")
    r = subprocess.run([sys.executable, "-c", CODE, json.dumps(CASES)], capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT")][0][6:])


@pytest.fixture(scope="module")
def both():
    return _run(True), _run(False)


def test_the_resident_driver_is_the_one_that_runs(both):
    res, pip = both
    for (wkw, skw), a, b in zip(CASES, res, pip):
        assert a[0]["driver"] == "resident" and a[1]["driver"] == "resident", (wkw, a[0]["driver"])
        assert b[0]["driver"] == "pipelined", (wkw, b[0]["driver"])


@pytest.mark.parametrize("k", range(len(CASES)))
def test_resident_solve_is_bit_identical_to_the_pipelined_one(both, k):
    res, pip = both
    a, b = res[k], pip[k]
    for rep in range(2):
        assert len(a[rep]["log"]) == len(b[rep]["log"]), (CASES[k], len(a[rep]["log"]), len(b[rep]["log"]))
        for it, (x, y) in enumerate(zip(a[rep]["log"], b[rep]["log"])):
            assert x == y, (CASES[k], rep, it, x, y)
        for f in ("message", "final_cost", "initial_cost", "cams", "xyz", "rec"):
            assert a[rep][f] == b[rep][f], (CASES[k], rep, f)
    assert a[2]["step"] == b[2]["step"], (CASES[k], a[2]["step"], b[2]["step"])
    n_steps = len(a[0]["log"]) - 1
    print("case %d: %d iterations, %d accepted, bit-identical (log, cameras, points, records); %s" %
          (k, n_steps, sum(1 for i in a[0]["log"][1:] if i[7]), a[0]["message"]))


def test_resident_solve_against_the_oracle():
    from oracle import oracle
    from photobundle_amd import synthetic
    from photobundle_amd.engine import Engine, default_solver_options
    wkw, skw = CASES[0]
    p = synthetic.make_window(**wkw)
    ref = oracle.solve(p, oracle.default_options(**skw))
    rows, cols = wkw["size"]
    with Engine(rows, cols, p.K, p.radius, p.n_frames, huber=p.huber) as e:
        e.load(p)
        res = e.solve(default_solver_options(**skw))
        assert e.solve_driver() == "resident"
    assert len(res["iterations"]) == len(ref["iterations"])
    for a, b in zip(ref["iterations"], res["iterations"]):
        assert a["step_is_successful"] == b["step_is_successful"]
        assert abs(a["cost"] - b["cost"]) <= 1e-9 * abs(a["cost"]), (a["cost"], b["cost"])
    assert float(np.abs(res["cams"] - ref["cams"]).max()) <= 1e-5


def test_a_lost_hand_over_ends_in_an_error_not_in_a_hang():
    """Every wait of the resident kernel is bounded: with a flag deliberately withheld (fault injection, PBA_RES_STOP=100: in the second step
    the last workgroup keeps its arrival flag to itself) the dependent waits time out on the device, the abort word makes every workgroup
    leave, the host finds a finished stream without a published result and returns an error; the engine is unusable afterwards (fails
    fast) -- nothing hangs."""
    code = textwrap.dedent("""
        import sys, time
        sys.path.insert(0, %r)
        from photobundle_amd import synthetic
        from photobundle_amd.engine import Engine, EngineError, default_solver_options
        p = synthetic.make_window(n_frames=5, n_points=2000, radius=1, size=(188, 621), K=(359.4, 359.4, 303.6, 92.6))
        e = Engine(188, 621, p.K, p.radius, p.n_frames, huber=p.huber)
        e.load(p)
        t0 = time.time()
        try:
            e.solve(default_solver_options(max_num_iterations=6))
            print("RESULT no error")
        except EngineError as exc:
            print("RESULT error after %%.1f s: %%s" %% (time.time() - t0, exc))
        t0 = time.time()
        try:
            e.solve(default_solver_options(max_num_iterations=6))
            print("SECOND no error")
        except EngineError as exc:
            print("SECOND error after %%.3f s: %%s" %% (time.time() - t0, exc))
        e.close()
    """ % ROOT)
    env = dict(os.environ, PBA_RESIDENT="1", PBA_RES_STOP="100", PBA_WAIT_TIMEOUT_S="4")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    first = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0]
    second = [l for l in r.stdout.splitlines() if l.startswith("SECOND")][0]
    print(first)
    print(second)
    assert first.startswith("RESULT error after") and "without publishing" in first, first
    assert float(first.split()[3]) < 30.0
    assert second.startswith("SECOND error after") and "unusable" in second and float(second.split()[3]) < 1.0, second


def test_a_refused_cooperative_launch_falls_back_to_the_pipelined_driver():
    """The runtime may refuse a cooperative launch before anything runs (co-residency granted to the process: CU masks, partition modes).
    PBA_RES_REFUSE=1 injects that refusal: the SAME pba_solve call then runs on the pipelined driver (reported as such), with the bits the
    resident solve produces; the engine stays there for later solves."""
    from photobundle_amd import synthetic
    from photobundle_amd.engine import Engine, default_solver_options
    wkw, skw = CASES[1]
    p = synthetic.make_window(**wkw)
    rows, cols = wkw["size"]
    out = []
    for refuse in (False, True):
        with Engine(rows, cols, p.K, p.radius, p.n_frames, huber=p.huber) as e:
            e.load(p)
            if refuse:
                os.environ["PBA_RES_REFUSE"] = "1"
            try:
                r = e.solve(default_solver_options(**skw))
            finally:
                os.environ.pop("PBA_RES_REFUSE", None)
            drivers = [e.solve_driver()]
            r2 = e.solve(default_solver_options(**skw))     # (no injection any more: a refused engine does not try again)
            drivers.append(e.solve_driver())
            out.append((drivers, r, r2))
    assert out[0][0] == ["resident", "resident"] and out[1][0] == ["pipelined", "pipelined"], (out[0][0], out[1][0])
    for k in (1, 2):
        a, b = out[0][k], out[1][k]
        assert [i["cost"] for i in a["iterations"]] == [i["cost"] for i in b["iterations"]]
        assert a["cams"].tobytes() == b["cams"].tobytes() and a["xyz"].tobytes() == b["xyz"].tobytes()


def test_the_epochs_restart_long_before_they_wrap():
    """The hand-over words carry 32-bit epochs that grow from launch to launch; before a launch would count beyond 0xF0000000 the flag block is
    zeroed and the count restarts at 1.  PBA_RES_EPOCH0 starts an engine 30 epochs short of that: the first solve runs on the high epochs, the
    second one restarts them, the third is an ordinary one -- all three with the bits of an engine that starts at epoch 1."""
    from photobundle_amd import synthetic
    from photobundle_amd.engine import Engine, default_solver_options
    wkw, skw = CASES[1]
    p = synthetic.make_window(**wkw)
    rows, cols = wkw["size"]
    out = []
    for epoch0 in (None, 0xF0000000 - 30):
        if epoch0 is not None:
            os.environ["PBA_RES_EPOCH0"] = str(epoch0)
        try:
            with Engine(rows, cols, p.K, p.radius, p.n_frames, huber=p.huber) as e:
                runs = []
                for rep in range(3):
                    e.load(p)
                    r = e.solve(default_solver_options(**skw))
                    assert e.solve_driver() == "resident"
                    runs.append(([i["cost"] for i in r["iterations"]], r["cams"].tobytes(), r["xyz"].tobytes()))
                out.append(runs)
        finally:
            os.environ.pop("PBA_RES_EPOCH0", None)
    assert out[0] == out[1]
    assert out[0][0] == out[0][1] == out[0][2]
