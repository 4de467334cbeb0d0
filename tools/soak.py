"""Dev tool: soak run of the LM drivers (python tools/soak.py [seconds] [steps] [small]).  One engine, BASELINE configs[1] -- or, with a third
argument, the reference's operating point (5 frames x 5 000 points x 3x3: the RESIDENT driver, one cooperative launch per solve); the
same window is solved over and over for the given wall time; every solve must return the bits of the first one (final cost,
cameras, iteration log), device memory in use must not grow, and the per-solve time distribution is printed."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from photobundle_amd import synthetic
from photobundle_amd.engine import Engine, default_solver_options


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    small = len(sys.argv) > 3
    prob = (synthetic.make_window(n_frames=5, n_points=5000, radius=1, visibility="dense") if small else
            synthetic.make_window(n_frames=8, n_points=50000, radius=2, visibility="dense"))
    eng = Engine(prob.planes.shape[2], prob.planes.shape[3], prob.K, prob.radius, max_frames=prob.cams.shape[0], huber=prob.huber)
    eng.load(prob)
    o = default_solver_options(max_num_iterations=steps, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    first, times, n = None, [], 0
    free0 = None
    t_end = time.time() + budget
    t_mark, marks = time.time() + budget / 10.0, []
    while time.time() < t_end:
        eng.set_problem(prob.xyz, prob.desc, prob.obs_point, prob.obs_slot, prob.weights)
        eng.set_cameras(prob.cams, prob.fixed_slot)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = eng.solve(o)
        times.append(time.perf_counter() - t0)
        key = (res["final_cost"], res["cams"].tobytes(), tuple(i["cost"] for i in res["iterations"]), res["xyz"][::997].tobytes())
        if first is None:
            first = key
            free0 = torch.cuda.mem_get_info()[0]
        elif key != first:
            print("solve %d differs from the first: final cost %r vs %r" % (n, res["final_cost"], first[0]))
            sys.exit(1)
        n += 1
        if time.time() >= t_mark:      # growth of the device memory in use along the run (a leak grows with the solves, a one-off pool growth does not)
            marks.append((n, free0 - torch.cuda.mem_get_info()[0]))
            t_mark += budget / 10.0
    free1 = torch.cuda.mem_get_info()[0]
    t = np.sort(np.array(times[1:])) * 1e3
    print("driver of the last solve: %s" % eng.solve_driver())
    print("%d solves of %d LM iterations in %.0f s, all bit-identical (final cost %.9e); ms per solve incl. read-back: min %.3f  p50 %.3f  "
          "p99 %.3f  max %.3f; device memory in use grew by %d bytes" % (n, steps, budget, first[0], t[0], t[len(t) // 2], t[int(0.99 * len(t))], t[-1], free0 - free1))
    print("memory growth along the run (solves, bytes): %s" % marks)
    eng.close()
    # a leak grows with the solves; the runtime's own pools grow once or twice early on (measured: +2 MiB after ~30 k and ~50 k cooperative
    # launches, then flat for 140 k more: profiles/r06/soak_resident_330s.txt) -- the second half of the run must not grow
    half = marks[len(marks) // 2 - 1][1] if len(marks) >= 2 else 0
    sys.exit(0 if (free0 - free1) - half <= (1 << 20) else 2)


main()
