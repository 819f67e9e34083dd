import cProfile, pstats, sys, os, time, io
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch, numpy as np
import bench
torch.set_grad_enabled(False)
from quadraturefields_amd import parallel, synthetic
from quadraturefields_amd.render import FrameRenderer
dev = torch.device("cuda:0")
mesh, mi, field = bench.build_scene(dev)
fr = FrameRenderer(mi, field, render_step_size=bench.STEP)
w = h = 800
focal = synthetic.lego_focal(w)
cams = synthetic.orbit_cameras(6, seed=42)
rays = [synthetic.camera_rays(c, focal, w, h, device=dev) for c in cams]
n = 8
sh = [parallel.ShardedFrameRenderer(fr, r, n) for r in range(n)]
cuts = [0, 224, 296, 352, 400, 448, 504, 576, 800]
def frame(i):
    o, d = rays[i]
    return [sh[r].render_band(o, d, cams[i], focal, w, h, cuts[r], cuts[r + 1]) for r in range(n)]
for i in range(3): frame(i)
torch.cuda.synchronize()
# host time per band, back to back (no sync), and GPU time for the same
t0 = time.perf_counter(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True); a.record()
for i in range(3, 6): frame(i)
b.record(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host enqueue per band us", (t1 - t0) / 24 * 1e6, " wall per band us", (t2 - t0) / 24 * 1e6, " gpu per band us", a.elapsed_time(b) / 24 * 1e3)
pr = cProfile.Profile(); pr.enable()
for i in range(3, 6): frame(i)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35); print(s.getvalue()[:6000])
