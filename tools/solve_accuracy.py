"""Dev tool (GPU box): the blocked L D L^T reduced solve against the independent plain-Cholesky k_solve_generic (PBA_SOLVE=1) and against
numpy on the system the engine itself read back, on windows of the random sweep (tests/test_gpu_random_shapes.py).
usage: solve_accuracy.py case [case ...]      (runs itself once per solver in a child process)"""
import os, subprocess, sys, tempfile
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np

if os.environ.get("PBA_ACC_CHILD"):
    os.environ.setdefault("PBA_RANDOM_CASES", "200")
    import test_gpu_random_shapes as T
    from gpu_util import make_engine
    out = {}
    for ci in [int(a) for a in sys.argv[1:]]:
        p = T._make(T.CASES[ci])
        with make_engine(p) as e:
            e.linearize()
            for it in range(3):
                c0 = e.get_state()[0].copy()
                info = e.step(1e4 * 3.0 ** it, init_scale=(it == 0))
                S, rhs = e.reduced_system()
                e.accept(); e.linearize()
                c1 = e.get_state()[0].copy()
                out["d_%d_%d" % (ci, it)] = c1 - c0
                out["S_%d_%d" % (ci, it)] = S; out["r_%d_%d" % (ci, it)] = rhs
                out["m_%d_%d" % (ci, it)] = np.array([info["model_cost_change"], info["step_norm"]])
    np.savez(os.environ["PBA_ACC_CHILD"], **out)
    sys.exit(0)

tmp = tempfile.mkdtemp()
res = {}
for solver in ("0", "1"):
    f = os.path.join(tmp, "s%s.npz" % solver)
    subprocess.check_call([sys.executable, __file__] + sys.argv[1:], env=dict(os.environ, PBA_ACC_CHILD=f, PBA_SOLVE=solver, PBA_ASYNC="0"))
    res[solver] = np.load(f)
for k in sorted(k for k in res["0"].files if k.startswith("d_")):
    a, b = res["0"][k], res["1"][k]
    S, r = res["0"]["S" + k[1:]], res["0"]["r" + k[1:]]
    same_sys = np.array_equal(S, res["1"]["S" + k[1:]]) and np.array_equal(r, res["1"]["r" + k[1:]])
    y = np.linalg.solve(S, r)
    resid = np.abs(S @ y - r).max() / np.abs(r).max()
    print("%s: |blocked - generic| / |step| = %.2e (same system read back: %s); n = %d, cond(S) = %.1e, numpy residual %.1e; model cost change blocked %.15e generic %.15e"
          % (k, np.abs(a - b).max() / np.abs(a).max(), same_sys, len(r), np.linalg.cond(S), resid, res["0"]["m" + k[1:]][0], res["1"]["m" + k[1:]][0]))
