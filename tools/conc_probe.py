"""Bisect probe for the several-scans-in-flight figure (bench.py: value_concurrent).

Runs the SAME measurement against any checkout of this repository (`--tree PATH`: that tree's own mimosa_amd package,
library and synthetic world are used — the entry points it needs have been stable since round 1):

  * `conc_ms`      4 factors on 4 contexts (HIP streams) sharing one map, cold linearizes dealt round-robin by ONE host
                   thread, <= 32 in flight per stream: what bench.py reports as value_concurrent
  * `single_ms`    the same loop with one stream (bench.py's `value`, events off)
  * `enqueue_us`   host time of one reset + linearize_async pair while the stream is far from full (the rate at which a
                   single host thread can feed the device at all)
  * `threads_ms`   the 4 streams fed by 4 host threads (ctypes releases the GIL during the calls)

Prints one JSON line.  tools/conc_bisect.sh exports a list of commits into bisect_trees/, builds each and runs this.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tree", default=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    ap.add_argument("--label", default="")
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--streams", type=int, default=4)
    ap.add_argument("--no-components", action="store_true", help="switch K4 off where the tree has mh_icp_set_components")
    ap.add_argument("--pre-contexts", type=int, default=0,
                    help="contexts (HIP streams) created BEFORE the measured ones, as the legs that precede bench.py's concurrent pass leave them")
    ap.add_argument("--pre-destroy", action="store_true", help="... and closed again before the measured ones are created")
    args = ap.parse_args()
    tree = os.path.abspath(args.tree)
    sys.path.insert(0, tree)
    os.chdir(tree)
    import bench  # the tree's own world builder
    from mimosa_amd import capi, synth

    INFLIGHT = 32
    room_clouds, pts, R, t = bench.build_world(0, "2x5", 128)
    cfgd = synth.enwide_config()
    ctx = capi.Context(0)
    gmap = capi.VoxelMap(ctx, leaf=cfgd["target_ivox_map_leaf_size"], min_dist=cfgd["target_ivox_map_min_dist_in_voxel"],
                         max_pts=synth.MAX_PTS_PER_VOXEL, mode=synth.ENWIDE_NEIGHBOR_MODE, lru_horizon=synth.ENWIDE_LRU_HORIZON)
    for xyz in room_clouds:
        gmap.insert(xyz)
    pre = [capi.Context(0) for _ in range(args.pre_contexts)]
    for c in pre:
        c.synchronize()
    if args.pre_destroy:
        for c in pre:
            c.close()
        pre = []
    ctxs, factors = [], []
    for s in range(args.streams):
        c = ctx if s == 0 else capi.Context(0)
        p, _ = synth.make_scan(128, seed=synth.BASE_SEED + 1 + 1000 * s)
        f = capi.ICPFactor(c, gmap, p, capi.make_reg_config(**cfgd))
        f.linearize(R, t)
        if args.no_components and hasattr(f, "set_components"):
            f.set_components(False)
        ctxs.append(c)
        factors.append(f)
    n_pts = len(pts)

    def sync():
        for c in ctxs:
            c.synchronize()

    def run_steps(fs, k):
        done = 0
        while done < k:
            nb = min(INFLIGHT * len(fs), k - done)
            for i in range(nb):
                f = fs[i % len(fs)]
                f.reset()
                f.linearize_async(R, t)
            for f in fs:
                f.wait()
            done += nb

    def timed(fs, k):
        run_steps(fs, 40)
        sync()
        a = time.perf_counter()
        run_steps(fs, k)
        sync()
        return (time.perf_counter() - a) / k * 1e3

    out = {"label": args.label, "tree": tree, "streams": args.streams, "steps": args.steps, "pre_contexts": args.pre_contexts,
           "pre_destroyed": bool(args.pre_destroy), "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}
    out["single_ms"] = round(min(timed(factors[:1], args.steps // 2) for _ in range(3)), 5)
    out["conc_ms"] = round(min(timed(factors, args.steps) for _ in range(3)), 5)
    # host cost of an enqueue: 16 pairs on an idle stream (nothing blocks), best of 20 rounds
    best = 1e9
    for _ in range(20):
        sync()
        a = time.perf_counter()
        for _i in range(16):
            factors[0].reset()
            factors[0].linearize_async(R, t)
        b = time.perf_counter()
        factors[0].wait()
        best = min(best, (b - a) / 16 * 1e6)
    out["enqueue_us"] = round(best, 2)

    # one host thread per stream
    def worker(f, k):
        run_steps([f], k)

    def threads_timed(k):
        for f in factors:
            run_steps([f], 10)
        sync()
        ths = [threading.Thread(target=worker, args=(f, k // len(factors))) for f in factors]
        a = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        sync()
        return (time.perf_counter() - a) / k * 1e3

    out["threads_ms"] = round(min(threads_timed(args.steps) for _ in range(3)), 5)
    out["conc_mpts"] = round(n_pts / out["conc_ms"] / 1e3, 1)
    out["threads_mpts"] = round(n_pts / out["threads_ms"] / 1e3, 1)
    out["single_mpts"] = round(n_pts / out["single_ms"] / 1e3, 1)
    print(json.dumps(out), flush=True)
    os._exit(0)  # old trees: no orderly teardown needed for a probe


if __name__ == "__main__":
    main()
