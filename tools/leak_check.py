#!/usr/bin/env python3
"""Device-memory growth over many create / use / destroy cycles of every handle type (hipMemGetInfo before and after; the
allocation cache keeps blocks, so the figure to watch is growth BETWEEN two identical batches of cycles)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimosa_amd import capi, synth, synth_photo as sp, replay

hip = C.CDLL("libamdhip64.so")
def rss_mb():
    import psutil
    return psutil.Process().memory_info().rss / 2**20


def free_mb():
    f, t = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return f.value / 2**20

ctx = capi.Context(0)
m = synth.make_room(1234, 0, 0, room=np.array([8.0, 6.0, 3.0]))
pts, aux = synth.make_scan(n_rows=32, seed=5, n_cols=256, room=np.array([8.0, 6.0, 3.0]), sensor_local=np.array([3.0, 3.0, 1.2]))
rc = capi.make_reg_config(**synth.enwide_config())
pcfg = sp.photo_config(rows=64, cols=512)
fr = [sp.make_frame(pcfg, k) for k in range(2)]
raw, _ = synth.make_raw_scan(32, n_cols=256)

def cycle():
    gm = capi.VoxelMap(ctx)
    gm.insert(m)
    g2 = gm.copy()
    g2.insert(m[:5000] + np.float32(0.3))
    f = capi.ICPFactor(ctx, g2, pts, rc)
    f.linearize(aux["R_W_L"], aux["t_W_L"])
    c = f.clone()
    capi.linearize_batch([f, c], [aux["R_W_L"]] * 2, [aux["t_W_L"]] * 2)
    sc = capi.Scan(ctx)
    sc.prepare_input(raw, capi.make_input_config())
    sc.destroy()
    P = capi.Photo(ctx, pcfg)
    P.preprocess(fr[0]["raw"], fr[0]["deskewed"], fr[0]["unique_ns"], fr[0]["T_Le_Lt"])
    P.detect(30, fr[0]["R_W_Be"], fr[0]["t_W_Be"], sp.BIAS_DIRECTIONS)
    P.preprocess(fr[1]["raw"], fr[1]["deskewed"], fr[1]["unique_ns"], fr[1]["T_Le_Lt"])
    pf = P.make_factor()
    pf.linearize(fr[1]["R_W_Be"], fr[1]["t_W_Be"])
    pf.destroy(); P.destroy(); c.destroy(); f.destroy(); g2.release(); gm.release()

for _ in range(20):
    cycle()
ctx.synchronize()
a, ra = free_mb(), rss_mb()
for _ in range(300):
    cycle()
ctx.synchronize()
b, rb = free_mb(), rss_mb()
for _ in range(300):
    cycle()
ctx.synchronize()
c, rc_ = free_mb(), rss_mb()
print(f"host RSS MB {ra:.1f} -> {rb:.1f} -> {rc_:.1f}")
print(f"free MB after warm-up {a:.1f}, after 300 cycles {b:.1f}, after 600 cycles {c:.1f}; growth per cycle {(b - c) / 300 * 1024:.2f} KB")
assert abs(b - c) < 64, "device memory keeps growing"
assert rc_ - rb < 64, "host memory keeps growing"
print("OK")
