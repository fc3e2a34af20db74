#!/usr/bin/env python3
"""Diagnostic: per-wave phase timeline of icp_linearize_kernel (needs the -DMH_TIMELINE build).

Stamps (s_memtime, shader-clock... 100 MHz REFCLK on gfx950 — calibrated below against the HIP-event
kernel time): 0 entry, 1 neighbourhood list built, 2 candidate scan done, 3 per-point work done,
4 after the block barrier, 5 partial row stored, 6 last block: won the ticket, 7 last block: done.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mimosa_amd import build as hb, capi, synth  # noqa: E402

lib = hb.build(timeline=True)  # add -DMH_BALANCE by hand for the lane-balance counters (MH_BALANCE=1 here)
capi._build.LIB = lib  # load the diagnostic variant
ctx = capi.Context(0)
L = ctx.L
L.mh_icp_timeline.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
rooms = [xyz for _, _, xyz in synth.make_map_rooms(2, 5)]
pts, aux = synth.make_scan(128)
NPTS = int(os.environ.get("MH_TL_POINTS", "0"))  # e.g. 24576: the reference's real cloud size (256-thread workgroups)
if NPTS:
    pts = np.ascontiguousarray(pts[::5][:NPTS])
WPB = 8 if len(pts) > 65536 else 4
R, t = synth.query_pose()
cfg = synth.enwide_config()
m = capi.VoxelMap(ctx)
for xyz in rooms:
    m.insert(xyz)
f = capi.ICPFactor(ctx, m, pts, capi.make_reg_config(**cfg))
ctx.set_profiling(True)
for _ in range(5):
    f.reset()
    r = f.linearize(R, t)
n = C.c_size_t()
buf = np.zeros(8 * 16 * 4096, np.uint64)
rc = L.mh_icp_timeline(f.h, buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(n))
assert rc == 0, rc
T = buf[: n.value].reshape(-1, 16).astype(np.int64)
blk = np.arange(len(T)) // WPB
live = T[:, 0] > 0
T, blk = T[live], blk[live]
k3_us = r["gpu_ms_linearize"] * 1e3
print(f"points {len(pts)}  waves {len(T)}  K3 by HIP events {k3_us:.1f} us  (ticks = s_memtime; per-XCD counters, so cross-wave times are per XCD)")


def stat(name, x):
    print(f"{name:40s} mean {x.mean():9.0f}  p50 {np.percentile(x,50):9.0f}  p95 {np.percentile(x,95):9.0f}  max {x.max():9.0f} ticks")


stat("A: lookup + list (0->1)", T[:, 1] - T[:, 0])
stat("   A.1 transform + 8 probes (0->8)", T[:, 8] - T[:, 0])
stat("   A.2 cell loads arrived (8->9)", T[:, 9] - T[:, 8])
stat("   A.3 mask/pack/LDS list (9->1)", T[:, 1] - T[:, 9])
stat("   B.1 centre voxel scan (1->10)", T[:, 10] - T[:, 1])
stat("   B.2 prune (10->11)", T[:, 11] - T[:, 10])
stat("   B.3 neighbour scan (11->2)", T[:, 2] - T[:, 11])
stat("   C.1 exact tier (2->12)", T[:, 12] - T[:, 2])
stat("   C.2 proof+plane+residual+stores (12->3)", T[:, 3] - T[:, 12])
stat("B: candidate scan (1->2)", T[:, 2] - T[:, 1])
if os.environ.get("MH_BALANCE"):  # build with MH_BALANCE=1 python -m mimosa_amd.build --timeline --force
  q_sum, steps = (T[:, 13] & 0xFFFF).astype(float), (T[:, 14] & 0xFFFF).astype(float)
  sh = np.stack([(T[:, 13] >> s) & 0xFF for s in (16, 24, 32, 40, 48)], 1)      # share: posted, taken back, done by helpers, done for others, rounds
  print("share: jobs posted", int(sh[:, 0].sum()), " taken back by their owner", int(sh[:, 1].sum()), " done by helpers", int(sh[:, 2].sum()),
        " (claimed", int(sh[:, 3].sum()), ") waves that posted", int((sh[:, 0] > 0).sum()), " waves that helped", int((sh[:, 3] > 0).sum()),
        " help rounds max", int(sh[:, 4].max()))
  heavy = np.stack([(T[:, 14] >> s) & 0xFF for s in (16, 24, 32, 40)], 1)   # lanes with > 8 / 12 / 16 / 24 neighbour quads
  print("lanes per wave with > 8 / 12 / 16 / 24 neighbour quads: mean", heavy.mean(0).round(2), " p95", np.percentile(heavy, 95, 0), " max", heavy.max(0))
  for lo, hi in ((0, 12), (12, 16), (16, 20), (20, 24), (24, 28), (28, 40)):
    sel = (steps > lo) & (steps <= hi)
    if sel.any():
      print(f"  waves with {lo:2d} < quad steps <= {hi:2d}: {int(sel.sum()):5d}   their lanes > 8 / 12 / 16 / 24: mean {heavy[sel].mean(0).round(1)}  max {heavy[sel].max(0)}")
  print(f"scan lane balance: mean quads/lane (centre incl.) {q_sum.mean()/64:.2f}, mean neighbour quad steps per wave {steps.mean():.2f}, "
        f"p95 {np.percentile(steps,95):.0f}, max {steps.max():.0f}")
  i_sum, i_max = (T[:, 15] & 0xFFFFFFFF).astype(float), (T[:, 15] >> 32).astype(float)
  print(f"ideal pruning (box nearer than the final k-th distance): mean neighbour quads/lane {i_sum.mean()/64:.2f}, "
        f"mean per-wave max {i_max.mean():.2f}, p95 {np.percentile(i_max,95):.0f}, max {i_max.max():.0f}")
else:
  ok = T[:, 13] > 0
  stat("   C.2a proof check + 5 bucket loads (12->13)", (T[:, 13] - T[:, 12])[ok])
  stat("   C.2b mean + covariance (13->14)", (T[:, 14] - T[:, 13])[ok])
  stat("   C.2c plane_eigen (14->15)", (T[:, 15] - T[:, 14])[ok])
  stat("   C.2d gates + residual + Jacobian + stores (15->3)", (T[:, 3] - T[:, 15])[ok])
stat("C: plane/residual (2->3)", T[:, 3] - T[:, 2])
stat("barrier wait (3->4)", T[:, 4] - T[:, 3])
stat("D: block reduce+store (4->5)", T[:, 5] - T[:, 4])
stat("wave total (0->5)", T[:, 5] - T[:, 0])
# per XCD: start skew and end times relative to the XCD's first entry
skew, endt, spans = [], [], []
for x in range(8):
    sel = (blk % 8) == x
    if not sel.any():
        continue
    t0 = T[sel, 0].min()
    skew.append(T[sel, 0] - t0)
    endt.append(T[sel, 5] - t0)
    last = T[sel][T[sel, 7] > 0]
    spans.append(max(T[sel, 5].max(), last[:, 7].max() if len(last) else 0) - t0)
stat("start skew within XCD", np.concatenate(skew))
stat("wave end within XCD (5 - xcd start)", np.concatenate(endt))
print("per-XCD span (ticks):", spans, " => ticks/us ~", max(spans) / k3_us)
last = T[T[:, 7] > 0]
if len(last):
    stat("last block: ticket wait (5->6)", last[:, 6] - last[:, 5])
    stat("last block: fold+eigen (6->7)", last[:, 7] - last[:, 6])

# the kernel ends with its slowest wave: show the tail
tot = T[:, 5] - T[:, 0]
order = np.argsort(-tot)[:12]
print("slowest waves: total | A  B1  prune  B3  exact  C2  barrier  D")
for i in order:
    t = T[i]
    print(f"  {tot[i]:7d} | {t[1]-t[0]:6d} {t[10]-t[1]:6d} {t[11]-t[10]:6d} {t[2]-t[11]:6d} {t[12]-t[2]:6d} {t[3]-t[12]:6d} {t[4]-t[3]:6d} {t[5]-t[4]:6d}")
work = T[:, 3] - T[:, 0]
print(f"per-wave work before the barrier (0->3): mean {work.mean():.0f} p50 {np.percentile(work,50):.0f} p95 {np.percentile(work,95):.0f} p99 {np.percentile(work,99):.0f} max {work.max()}")
bw = np.array([work[blk == b].max() for b in np.unique(blk)])
print(f"per-block slowest wave: mean {bw.mean():.0f} p95 {np.percentile(bw,95):.0f} max {bw.max()}")
