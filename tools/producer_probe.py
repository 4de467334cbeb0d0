"""GPU box: the device-side producers at KITTI size timed on the host clock: 20 x BitPlanes, 20 x
IntensityAndGradient frames, 20 x a three-level image pyramid."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from photobundle_amd.engine import Engine
rows, cols, K = 376, 1241, (718.856, 718.856, 607.1928, 185.2157)
rng = np.random.default_rng(0)
img = rng.integers(0, 256, size=(rows, cols), dtype=np.uint8)
e8 = Engine(rows, cols, K, 1, 5, channels=8)
e3 = Engine(rows, cols, K, 1, 5, channels=3)
lv = [Engine(rows, cols, K, 1, 5), Engine((rows + 1) // 2, (cols + 1) // 2, K, 1, 5), Engine(((rows + 1) // 2 + 1) // 2, ((cols + 1) // 2 + 1) // 2, K, 1, 5)]
import time
def timed(name, fn, sync, n=20):
    fn(0); sync()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    sync()
    print("%-34s %.3f ms per frame (wall clock, %d frames back to back, incl. the pinned copy + upload)" % (name, 1e3 * (time.perf_counter() - t0) / n, n), flush=True)
timed("BitPlanes channels", lambda i: e8.set_frame_descriptor(i % 5, img, "BitPlanes"), lambda: e8.get_frame_channel(0, 0))
timed("IntensityAndGradient channels", lambda i: e3.set_frame_descriptor(i % 5, img, "IntensityAndGradient"), lambda: e3.get_frame_channel(0, 0))
timed("plain u8 frame", lambda i: lv[0].set_frame(i % 5, img), lambda: lv[0].get_frame_planes(0))
def pyr(i):
    lv[1].set_frame_pyr_down(i % 5, lv[0], i % 5, want_image=False)
    lv[2].set_frame_pyr_down(i % 5, lv[1], i % 5, want_image=False)
timed("two pyrDown levels (no read-back)", pyr, lambda: lv[2].get_frame_planes(0))
def pyr_rb(i):
    lv[1].set_frame_pyr_down(i % 5, lv[0], i % 5)
    lv[2].set_frame_pyr_down(i % 5, lv[1], i % 5)
timed("two pyrDown levels + images back", pyr_rb, lambda: None)
t0 = time.perf_counter(); e8.get_frame_channels(0); print("BitPlanes channel read-back (15 MB)    %.3f ms" % (1e3 * (time.perf_counter() - t0)))
for e in [e8, e3] + lv:
    e.close()
print("done")
