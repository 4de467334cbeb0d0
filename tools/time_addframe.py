"""Per-addFrame timing of the drop-in class at KITTI size (scratch tool): writes a synthetic sequence, runs
photobundle_amd/bin/run_kitti with verbose = 1 and prints the class's own phase timers.
usage: time_addframe.py [n_frames] [maxNumPoints] [slidingWindowSize] [patchRadius] [numLevels] [descriptorType]"""
import os, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from test_gpu_dropin_class import _write_sequence, RUN
from photobundle_amd import synthetic

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 12
max_pts = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
window = int(sys.argv[3]) if len(sys.argv) > 3 else 5
radius = int(sys.argv[4]) if len(sys.argv) > 4 else 1
levels = int(sys.argv[5]) if len(sys.argv) > 5 else 1
dtype = sys.argv[6] if len(sys.argv) > 6 else "Intensity"
tmp = tempfile.mkdtemp()
_write_sequence(tmp, n_frames, synthetic.KITTI_SIZE, synthetic.KITTI_K)
cfg = os.path.join(tmp, "t.cfg")
with open(cfg, "w") as f:
    f.write("DataDirectory = %s\nTrajectory = %s/init.txt\nnumLevels = %d\nmaxNumPoints = %d\nslidingWindowSize = %d\npatchRadius = %d\n"
            "minScore = 0.65\nrobustThreshold = 0.05\nverbose = 1\ndescriptorType = %s\n" % (tmp, tmp, levels, max_pts, window, radius, dtype))
r = subprocess.run([RUN, "-c", cfg, "-o", os.path.join(tmp, "o.txt")], capture_output=True, text=True, timeout=1500)
print("rc", r.returncode)
for l in (r.stderr + r.stdout).split("\n"):
    if l.startswith(("addFrame", "optimize phases", "pba_solve:", "Using", "get_state", "solve_async", "pyramid build")):
        print(l[:400])
