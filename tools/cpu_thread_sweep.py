"""Dev tool: the oracle's LM iterations/s on bench.py's bounded CPU sample as a function of the OpenMP thread count
(python tools/cpu_thread_sweep.py [threads ...]); prints the host's core count, affinity mask size and cgroup quota beside it."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from photobundle_amd import synthetic
from oracle import oracle

def main():
    ths = [int(a) for a in sys.argv[1:]] or [4, 8, 16, 32, 64, 128, os.cpu_count() or 1]
    print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            print(f, open(f).read().strip())
        except OSError:
            pass
    prob = synthetic.make_window(n_frames=8, n_points=50000, radius=2, visibility="dense")
    for t in ths:
        o = oracle.default_options(max_num_iterations=5, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0,
                                   num_threads=t, use_autodiff=1)
        t0 = time.perf_counter()
        r = oracle.solve(prob, o)
        dt = time.perf_counter() - t0
        it = len(r["iterations"]) - 1
        print("threads %4d: %d iterations in %.2f s = %.2f LM iterations/s" % (t, it, dt, it / dt), flush=True)

main()
