import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from photobundle_amd import synthetic, se3
from photobundle_amd.engine import Engine, default_solver_options
from gpu_util import make_engine
p = synthetic.make_window()
with make_engine(p, keep_reduced_system=False) as e:
    res = e.solve(default_solver_options(max_num_iterations=10, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0))
for tag, cams, xyz in (("initial", p.cams, p.xyz), ("after 10 its", res["cams"], res["xyz"])):
    fx, fy, cx, cy = p.K
    _, _, rows, cols = p.planes.shape
    R = p.radius; W = 2 * R + 1
    from scipy.spatial.transform import Rotation
    Rm = Rotation.from_rotvec(cams[:, :3]).as_matrix()
    X = xyz[p.obs_point]; Rs = Rm[p.obs_slot]; t = cams[p.obs_slot, 3:]
    xc = np.einsum("nij,nj->ni", Rs, X) + t
    u = fx * xc[:, 0] / xc[:, 2] + cx; v = fy * xc[:, 1] / xc[:, 2] + cy
    offs = np.arange(-R, R + 1)
    xf = (u[:, None] + offs[None, :]).astype(np.float32); yf = (v[:, None] + offs[None, :]).astype(np.float32)
    ix = np.trunc(xf).astype(np.int64); iy = np.trunc(yf).astype(np.int64)
    bx, by = ix[:, 0], iy[:, 0]
    inside = (bx >= 0) & (bx + W - 1 <= cols - 2) & (by >= 0) & (by + W - 1 <= rows - 2)
    consec = (ix == bx[:, None] + np.arange(W)[None, :]).all(1) & (iy == by[:, None] + np.arange(W)[None, :]).all(1)
    reg = inside & consec
    irr = ~reg
    def axis(x, size):
        i = np.trunc(x).astype(np.int64)
        a1 = np.where(i < 0, 0, np.where(i > size - 2, size - 1, i)); a2 = np.where(i < 0, 0, np.where(i > size - 2, size - 1, i + 1))
        return a1, a2
    y1, y2 = axis(yf, rows); x1, x2 = axis(xf, cols)
    fits = ((y2[:, -1] - y1[:, 0]) <= W) & (y2[:, -1] >= y1[:, 0]) & ((x2[:, -1] - x1[:, 0]) <= W) & (x2[:, -1] >= x1[:, 0])
    fits &= ((y1[:, 1:-1] >= y1[:, :1]) & (y2[:, 1:-1] <= y1[:, :1] + W) & (x1[:, 1:-1] >= x1[:, :1]) & (x2[:, 1:-1] <= x1[:, :1] + W)).all(1)
    win = irr & fits; wild = irr & ~fits
    waves = np.arange(p.n_obs) // 64
    print(tag, "irregular", int(irr.sum()), "border", int((irr & ~inside).sum()), "rounding", int((irr & inside).sum()), "windowed", int(win.sum()), "wild", int(wild.sum()),
          "waves with irregular", len(np.unique(waves[irr])), "waves with wild", len(np.unique(waves[wild])))
