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
buf = np.zeros(2 * 8 * 16 * 4096, np.uint64)
rc = L.mh_icp_timeline(f.h, buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(n))
assert rc == 0, rc
Tall = buf[: n.value].reshape(-1, 16).astype(np.int64)
T = Tall[: len(Tall) // 2]    # K3's waves
T4 = Tall[len(Tall) // 2:]    # K4's waves (4 per workgroup), stamps of icp_localizability_body
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

# ---- K4 (icp_localizability_kernel): 0 entry, 1 K3's rows folded, 2 record entries requested, 3 eigenbases (the two lanes that
# decompose) / nothing, 5 after the barrier (bases known), 6 projections + histogram done,
# 7 wave sums in LDS, 8 after the barrier, 9 flagged words acknowledged
blk4 = np.arange(len(T4)) // 4
live4 = T4[:, 0] > 0
T4, blk4 = T4[live4], blk4[live4]
if len(T4):
    print(f"K4 waves {len(T4)}  K4 by HIP events {r['gpu_ms_localizability'] * 1e3:.1f} us")
    stat("K4 entry -> K3's rows folded (0->1)", T4[:, 1] - T4[:, 0])
    stat("K4 record entries requested (1->2)", T4[:, 2] - T4[:, 1])
    stat("K4 eigen lanes / pass-through (2->3)", T4[:, 3] - T4[:, 2])
    stat("K4 barrier: bases known (3->5)", T4[:, 5] - T4[:, 3])
    stat("K4 projections + histogram (5->6)", T4[:, 6] - T4[:, 5])
    stat("K4 wave sums (6->7)", T4[:, 7] - T4[:, 6])
    stat("K4 barrier (7->8)", T4[:, 8] - T4[:, 7])
    stat("K4 flagged words out + ack (8->9)", T4[:, 9] - T4[:, 8])
    stat("K4 wave total (0->9)", T4[:, 9] - T4[:, 0])
    # same-XCD clocks: last K3 stamp of the XCD -> first / last K4 entry, K4 span
    gaps, spans4, skew4 = [], [], []
    for x in range(8):
        s3, s4 = (blk % 8) == x, (blk4 % 8) == x
        if not s3.any() or not s4.any():
            continue
        e3 = max(T[s3, 5].max(), T[s3, 7].max())
        gaps.append(T4[s4, 0].min() - e3)
        skew4.append(T4[s4, 0].max() - T4[s4, 0].min())
        spans4.append(T4[s4, 9].max() - T4[s4, 0].min())
    print("K3 last stamp -> first K4 entry per XCD (ticks):", gaps)
    print("K4 entry skew per XCD:", skew4, " K4 span per XCD:", spans4)
