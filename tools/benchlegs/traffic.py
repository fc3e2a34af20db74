"""bench.py side leg (tools/benchlegs): HBM traffic of the dominant kernel (self-profiling rocprofv3 --pmc passes) and the compulsory-bytes bound"""
import ctypes as C  # noqa: F401
import json  # noqa: F401
import os  # noqa: F401
import sys  # noqa: F401
import time  # noqa: F401

import numpy as np  # noqa: F401

from .env import HBM_COPY_GBS, HBM_PEAK_GBS, INFLIGHT, INFLIGHT_ICP, ROOT  # noqa: F401

def measure_traffic(args):
    """({counter: mean per K3 launch}, note) from separate rocprofv3 --pmc passes of bench.py in --profile-mode, or (None, reason).
    FETCH_SIZE and WRITE_SIZE each in a pass of its own (MI355X_MICROARCH.md), SQ_INSTS_VALU in a third."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None, "rocprofv3 not found"
    got = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
        d = tempfile.mkdtemp(prefix="mh_pmc_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py"),
               "--steps", "40", "--warmup", "5", "--profile-mode", "--no-measure-traffic", "--rooms", args.rooms, "--rows", str(args.rows)]
        try:
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=240, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL)
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    kn = r.get("Kernel_Name", "")
                    if "icp_linearize_kernel" in kn and "batch" not in kn and r.get("Counter_Name") == ctr:
                        vals.append(float(r["Counter_Value"]))
            if len(vals) < 10:
                if ctr == "SQ_INSTS_VALU":
                    continue  # the traffic figure stands without the instruction count
                return None, f"the {ctr} pass produced {len(vals)} samples"
            got[ctr] = float(np.mean(vals[5:]))   # the first launches include the map's first touch
        except Exception as exc:  # noqa: BLE001
            if ctr == "SQ_INSTS_VALU":
                continue
            return None, f"the {ctr} pass failed: {type(exc).__name__}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    got["traffic"] = int((2.0 * got["FETCH_SIZE"] + got["WRITE_SIZE"]) * 1024)
    note = (f"measured IN THIS RUN: separate rocprofv3 --kernel-trace --pmc passes of this script (--profile-mode, 40 steps), FETCH_SIZE "
            f"{got['FETCH_SIZE']:.0f} KB and WRITE_SIZE {got['WRITE_SIZE']:.0f} KB per launch; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 "
            "(FETCH_SIZE counts 128-byte requests at 64 bytes on gfx950: an upper bound for 16-byte scattered gathers)")
    return got, note


def traffic(E):
    """({"traffic": HBM bytes per K3 launch, "SQ_INSTS_VALU": ...} or None, note).  PMC counters cannot be collected inside this
    process, so the run profiles ITSELF (measure_traffic).  Without a profiler on the box the committed summary is used ONLY if it
    was taken from the kernel source of this tree (hash of icp_kernels.hip); a stale one is reported as null, loudly."""
    got, note = measure_traffic(E.args)
    if got is not None:
        return got, note
    why = note
    try:
        import hashlib
        pm = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc.json")))
        src_hash = hashlib.sha256(open(os.path.join(ROOT, "mimosa_amd", "csrc", "icp_kernels.hip"), "rb").read()).hexdigest()[:16]
        if pm.get("kernel_source_sha16") == src_hash:
            got = {"traffic": int((2.0 * pm["FETCH_SIZE_KB"] + pm["WRITE_SIZE_KB"]) * 1024)}
            if pm.get("SQ_INSTS_VALU"):
                got["SQ_INSTS_VALU"] = float(pm["SQ_INSTS_VALU"])
            return got, f"NOT measured in this run ({why}); committed summary of the same kernel source: {pm['source']} @ {pm.get('commit')}"
        return None, (f"NOT measured in this run ({why}) and profiles/latest_pmc.json was taken from another version of icp_kernels.hip "
                      f"({pm.get('kernel_source_sha16')} != {src_hash}): no traffic figure")
    except Exception as exc:  # noqa: BLE001
        return None, f"NOT measured in this run ({why}); no usable committed summary ({type(exc).__name__})"


def compulsory(E):
    """(compulsory bytes per launch, voxels touched) — SURVEY.md 8(d)'s read-every-bucket-once bound, or (None, None)."""
    args, rank, gmap, cfgd, pts, R, t, n_pts, synth = E.args, E.rank, E.gmap, E.cfgd, E.pts, E.R, E.t, E.n_pts, E.synth
    # Compulsory lower bound of SURVEY.md §8(d): every touched voxel bucket read ONCE — N x 16 (source) + V_touched x 336
    # (16-B slot + 320-B bucket) + N x 64 (state out); V_touched = distinct occupied voxels in the 19-neighbourhoods of
    # all queries.  (Test-side numpy on the downloaded map; not in any timed region.)
    comp_bytes, v_touched = None, None
    if not args.profile_mode and rank == 0:
        cloud = gmap.get_cloud()
        leaf = cfgd["target_ivox_map_leaf_size"]
        def _keys(c):
            c = c.astype(np.int64) + (1 << 20)
            return (c[:, 0] << 42) | (c[:, 1] << 21) | c[:, 2]
        vm = np.floor(cloud.astype(np.float64) * (1.0 / leaf)).astype(np.int64)
        occ = np.unique(_keys(vm))
        q = synth.points_xyz(pts).astype(np.float64) @ R.T + t
        cq = np.floor(q * (1.0 / leaf)).astype(np.int64)
        offs = np.array([(i, j, k) for i in (-1, 0, 1) for j in (-1, 0, 1) for k in (-1, 0, 1) if not (i and j and k)], np.int64)
        touched = np.unique(np.concatenate([_keys(cq + o) for o in offs]))
        v_touched = int(np.isin(touched, occ, assume_unique=True).sum())
        comp_bytes = int(n_pts * 16 + v_touched * 336 + n_pts * 64)

    return comp_bytes, v_touched
