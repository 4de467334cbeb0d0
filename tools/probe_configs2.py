"""Scratch probe: window sizes the front-end produces at full KITTI size for a few (maxNumPoints, nonMaxSuppRadius)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_dropin_class import _write_sequence, RUN
from photobundle_amd import synthetic
tmp = tempfile.mkdtemp()
_write_sequence(tmp, 10, synthetic.KITTI_SIZE, synthetic.KITTI_K)
for mp, nms in [(100000, 1), (8000, 0), (60000, 1)]:
    cfg = os.path.join(tmp, "c.cfg")
    open(cfg, "w").write("DataDirectory = %s\nTrajectory = %s/init.txt\nnumLevels = 3\nmaxNumPoints = %d\nnonMaxSuppRadius = %d\n"
                         "slidingWindowSize = 8\npatchRadius = 2\nminScore = 0.75\nrobustThreshold = 0.05\nverbose = 0\n" % (tmp, tmp, mp, nms))
    r = subprocess.run([RUN, "-c", cfg, "-o", os.path.join(tmp, "o.txt")], capture_output=True, text=True)
    print(mp, nms, r.returncode, re.findall(r"Using (\d+) points \((\d+) residual blocks\)", r.stderr))
