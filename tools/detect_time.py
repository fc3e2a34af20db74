import sys, os, time, numpy as np
sys.path.insert(0, "/root/repo")
from mimosa_amd import capi, synth_photo as sp
ctx = capi.Context(0)
pcfg = sp.photo_config(patch=5)
pf = [sp.make_frame(pcfg, k) for k in range(2)]
for rep in range(3):
    G = capi.Photo(ctx, pcfg)
    G.preprocess(pf[0]["raw"], pf[0]["deskewed"], pf[0]["unique_ns"], pf[0]["T_Le_Lt"])
    ctx.synchronize(); a = time.perf_counter()
    G.detect(60, pf[0]["R_W_Be"], pf[0]["t_W_Be"], sp.BIAS_DIRECTIONS)
    print("detect total us", (time.perf_counter() - a) * 1e6, file=sys.stderr)
    G.destroy()
