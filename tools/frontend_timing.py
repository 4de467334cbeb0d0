"""Wall time of the drop-in class (run_kitti) on a synthetic KITTI-size sequence: total, solver part, front-end per frame.
Runs ON THE GPU BOX.  usage: python tools/frontend_timing.py [n_frames] [window] [max_points]"""
import os, re, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from photobundle_amd import synthetic
from test_gpu_dropin_class import _write_sequence, RUN

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 10
window = int(sys.argv[2]) if len(sys.argv) > 2 else 5
max_points = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
levels = int(sys.argv[4]) if len(sys.argv) > 4 else 1
tmp = tempfile.mkdtemp()
_write_sequence(tmp, n_frames, synthetic.KITTI_SIZE, synthetic.KITTI_K)
cfg = os.path.join(tmp, "t.cfg")
with open(cfg, "w") as f:
    f.write("DataDirectory = %s\nTrajectory = %s/init.txt\nmaxNumPoints = %d\nslidingWindowSize = %d\npatchRadius = 2\n"
            "minScore = 0.65\nrobustThreshold = 0.05\nverbose = 1\nnumLevels = %d\n" % (tmp, tmp, max_points, window, levels))
t = time.perf_counter()
r = subprocess.run([RUN, "-c", cfg, "-o", os.path.join(tmp, "out.txt")], capture_output=True, text=True, timeout=1200)
wall = time.perf_counter() - t
assert r.returncode == 0, r.stderr[-2000:]
solves = [(int(m.group(1)), float(m.group(2))) for m in re.finditer(r"iterations (\d+) \(successful \d+\)\s+([0-9.]+) s", r.stdout)]
used = re.findall(r"Using (\d+) points \((\d+) residual blocks\)", r.stderr)
print("frames %d, window %d, maxNumPoints %d: wall %.3f s (includes process start, file reads)" % (n_frames, window, max_points, wall))
print("optimisations: %d, points/blocks per window: %s" % (len(solves), used[:3]))
print("solver: %.1f ms total, %s iterations" % (1e3 * sum(s for _, s in solves), [n for n, _ in solves]))
m = re.findall(r"addFrame ([0-9.]+) ms \(front-end ([0-9.]+) ms", r.stderr)
if m:
    print("addFrame mean %.1f ms, front-end mean %.1f ms" % (np.mean([float(a) for a, _ in m]), np.mean([float(b) for _, b in m])))
m2 = re.findall(r"\[frame\+upload ([0-9.]+), visibility ([0-9.]+), saliency ([0-9.]+), candidates ([0-9.]+), top-N\+descriptors ([0-9.]+)\]", r.stderr)
if m2:
    a = np.array([[float(v) for v in row] for row in m2])
    print("front-end phases (mean ms): frame+upload %.2f, visibility %.2f, saliency %.2f, candidates %.2f, top-N+descriptors %.2f" % tuple(a.mean(0)))
for line in r.stderr.splitlines():
    if line.startswith("addFrame"):
        print(line)
for line in r.stderr.splitlines():
    if line.startswith("optimize phases"):
        print(line)
pb = re.findall(r"pyramid build ([0-9.]+) ms", r.stderr)
if pb:
    print("pyramid build mean %.2f ms per frame (%d levels); addFrame lines above are per LEVEL" % (np.mean([float(v) for v in pb[2:]]), levels))
