"""Does a stream of SMALL kernel dispatches on another HIP stream slow K3 down?  (Round-4 question behind the side-stream K4:
every kernel dispatch begins with a cache acquire; if that invalidates the L2 lines K3 has warmed, a K4 starting in the middle
of the next K3 costs it more than the overlap buys.)

Main context: the bench's pipelined loop on the configs[1] factor, components OFF (K3 only, one stream).  Second context, own
host thread: a tiny factor (256 points) linearized in a tight loop — tens of thousands of small dispatches per second on another
hardware queue.  Prints the main loop's ms per step without and with the second thread."""
import json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mimosa_amd import capi, synth

room_clouds, pts, R, t = bench.build_world(0, "2x5", 128)
cfgd = synth.enwide_config()
ctx = capi.Context(0)
gmap = capi.VoxelMap(ctx, leaf=cfgd["target_ivox_map_leaf_size"], min_dist=cfgd["target_ivox_map_min_dist_in_voxel"],
                     max_pts=synth.MAX_PTS_PER_VOXEL, mode=synth.ENWIDE_NEIGHBOR_MODE, lru_horizon=synth.ENWIDE_LRU_HORIZON)
for xyz in room_clouds:
    gmap.insert(xyz)
cfg = capi.make_reg_config(**cfgd)
f = capi.ICPFactor(ctx, gmap, pts, cfg)
f.linearize(R, t)
f.set_components(False)
c2 = capi.Context(0)
g = capi.ICPFactor(c2, gmap, pts[:256], cfg)
g.linearize(R, t)
g.set_components(False)


def loop(k):
    done = 0
    while done < k:
        nb = min(32, k - done)
        for _ in range(nb):
            f.reset()
            f.linearize_async(R, t)
        f.wait()
        done += nb


def timed(k):
    loop(40)
    ctx.synchronize()
    a = time.perf_counter()
    loop(k)
    ctx.synchronize()
    return (time.perf_counter() - a) / k * 1e3


out = {"alone_ms": round(min(timed(400) for _ in range(3)), 5)}
stop = False
count = [0]


def pest():
    while not stop:
        for _ in range(16):
            g.reset()
            g.linearize_async(R, t)
        g.wait()
        count[0] += 16


th = threading.Thread(target=pest)
th.start()
time.sleep(0.2)
c0, a = count[0], time.perf_counter()
out["with_small_dispatches_ms"] = round(min(timed(400) for _ in range(3)), 5)
out["small_dispatches_per_s"] = round((count[0] - c0) / (time.perf_counter() - a))
stop = True
th.join()
print(json.dumps(out), flush=True)
os._exit(0)
