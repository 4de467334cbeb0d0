"""Dev aid: iteration log (hex) + cameras hash of a few windows with the library named by PBA_LIB -- run once per library, diff the outputs."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photobundle_amd import synthetic
from photobundle_amd.engine import Engine, default_solver_options
for kw, n_it in [(dict(n_frames=4, n_points=300, radius=2, size=(120, 160), K=(200.0, 200.0, 80.0, 60.0)), 8),
                 (dict(n_frames=8, n_points=3000, radius=2, size=(188, 621), K=(359.4, 359.4, 303.6, 92.6)), 8),
                 (dict(n_frames=16, n_points=1500, radius=2, size=(188, 621), K=(359.4, 359.4, 303.6, 92.6), dense_births=(0, 8)), 6),
                 (dict(n_frames=5, n_points=2000, radius=1, size=(188, 621), K=(359.4, 359.4, 303.6, 92.6)), 10)]:
    p = synthetic.make_window(**kw)
    with Engine(kw["size"][0], kw["size"][1], p.K, p.radius, p.n_frames, huber=p.huber) as e:
        e.load(p)
        r = e.solve(default_solver_options(max_num_iterations=n_it))
        print(kw["n_frames"], kw["n_points"], [i["cost"].hex() for i in r["iterations"]], hashlib.sha1(r["cams"].tobytes()).hexdigest()[:12],
              hashlib.sha1(e.obs_records().tobytes()).hexdigest()[:12])
