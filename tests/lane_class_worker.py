"""Worker of tests/test_gpu_lane_classes.py: for every case "n_pts:binary" on the command line one cold + one warm linearize of
a cloud in the K3 launch class the environment selects (MH_QL2_MAX / MH_QL4_MAX are read once per process); results and
per-point state of all cases to one .npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mimosa_amd import capi, synth  # noqa: E402

out, cases = sys.argv[1], [(int(c.split(":")[0]), c.split(":")[1] == "1") for c in sys.argv[2:]]
ctx = capi.Context(0)
m = synth.make_room(synth.BASE_SEED, 0, 0)
scan, _ = synth.make_scan(64)
R, t = synth.query_pose()
gm = capi.VoxelMap(ctx)
gm.insert(m)
keys = ("H_ss", "b_s", "f", "status_hist", "loc_trans_comp", "loc_rot_comp", "n_knn", "mean_candidates", "n_exact_fallback", "H_st", "H_tt", "b_t")
blob = {}
for ci, (n_pts, binary) in enumerate(cases):
    pts = np.ascontiguousarray(scan[:: max(1, len(scan) // n_pts)][:n_pts])
    f = capi.ICPFactor(ctx, gm, pts, capi.make_reg_config(**synth.enwide_config()), binary=binary)
    kw = dict(R_tgt=synth.so3_exp(np.array([0.002, 0.001, -0.003])), t_tgt=np.array([0.02, -0.01, 0.005])) if binary else {}
    cold = f.linearize(R, t, **kw)
    st_cold = f.state()
    warm = f.linearize(R @ synth.so3_exp(np.array([0.003, -0.002, 0.004])), t + np.array([0.08, -0.05, 0.02]), **kw)  # part of the points re-associate
    st_warm = f.state()
    for ph, res, st in (("cold", cold, st_cold), ("warm", warm, st_warm)):
        blob.update({f"c{ci}_{ph}_{k}": np.asarray(res[k]) for k in keys})
        blob.update({f"c{ci}_{ph}_state{i}": x for i, x in enumerate(st)})
    f.destroy()
np.savez(out, **blob)
gm.release()
ctx.close()
print("OK")
