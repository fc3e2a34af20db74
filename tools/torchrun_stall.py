"""Where does the one-off 35-58 ms stall under torch.distributed go?  (tools/torchrun_probe.py finds it; rocprofv3 showed the GPU
idle and no long HIP call.)  This probe times EVERY library call of the pipelined bursts and, for the slow one, reports
  * which call it was (mh_icp_reset / mh_icp_linearize_async / mh_icp_wait) and what the main thread did meanwhile: its own CPU time
    (time.thread_time: on-CPU in native code vs blocked), voluntary / involuntary context switches, page faults (getrusage, thread);
  * which OTHER threads of the process burnt CPU in that window (/proc/self/task/*/stat utime + stime by thread name);
  * whether a 1 kHz Python sampler thread could run (a gap in its ticks = the interpreter lock was held through the stall).
usage: python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port P tools/torchrun_stall.py
       (PLAIN=1 python tools/torchrun_stall.py runs the same loop without torch.distributed)"""
import json, os, resource, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from mimosa_amd import capi, synth

PLAIN = bool(os.environ.get("PLAIN"))
lr = int(os.environ.get("LOCAL_RANK", "0"))
if not PLAIN:
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(lr)
    dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
    bt = torch.zeros(1, device="cuda")
ctx = capi.Context(lr)
room_clouds, pts, R, t = bench.build_world(0, "2x5", 128)
cfgd = synth.enwide_config()
gmap = capi.VoxelMap(ctx, leaf=cfgd["target_ivox_map_leaf_size"], min_dist=cfgd["target_ivox_map_min_dist_in_voxel"],
                     max_pts=synth.MAX_PTS_PER_VOXEL, mode=synth.ENWIDE_NEIGHBOR_MODE, lru_horizon=synth.ENWIDE_LRU_HORIZON)
for xyz in room_clouds:
    gmap.insert(xyz)
f = capi.ICPFactor(ctx, gmap, pts, capi.make_reg_config(**cfgd))
f.linearize(R, t)


def threads_cpu():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            st = open(f"/proc/self/task/{tid}/stat").read()
            name = st[st.index("(") + 1:st.rindex(")")]
            fld = st[st.rindex(")") + 2:].split()
            out[(int(tid), name)] = (int(fld[11]) + int(fld[12])) / os.sysconf("SC_CLK_TCK")  # utime + stime, seconds
        except (OSError, ValueError):
            pass
    return out


# CPython's cyclic garbage collector: every collection with its generation and duration (gc.callbacks runs on the collecting thread)
import gc
gc_log, _gc_t0 = [], [0.0]


def _gc_cb(phase, info):
    if phase == "start":
        _gc_t0[0] = time.perf_counter()
    else:
        gc_log.append((_gc_t0[0], time.perf_counter(), info["generation"], info["collected"]))


gc.callbacks.append(_gc_cb)
if os.environ.get("GC_FREEZE"):
    gc.collect()
    gc.freeze()  # everything alive now (torch's module graph) moves to the permanent generation: later collections do not walk it
ticks = []
stop = False


def sampler():
    while not stop:
        ticks.append(time.perf_counter())
        time.sleep(0.001)


th = threading.Thread(target=sampler, daemon=True)
th.start()
RU = resource.RUSAGE_THREAD
report = []
for blk in range(10):
    ctx.synchronize()
    if not PLAIN:
        dist.barrier(); torch.cuda.synchronize()
    calls = []
    cpu0, ru0, th0, w0 = time.thread_time(), resource.getrusage(RU), threads_cpu(), time.perf_counter()
    worst = (0.0, None)
    for i in range(64):
        for name, fn in (("reset", f.reset), ("linearize_async", lambda: f.linearize_async(R, t))):
            a, ca, ra = time.perf_counter(), time.thread_time(), resource.getrusage(RU)
            fn()
            b, cb, rb = time.perf_counter(), time.thread_time(), resource.getrusage(RU)
            if b - a > worst[0]:
                worst = (b - a, dict(call=name, index=i, wall_ms=round((b - a) * 1e3, 3), thread_cpu_ms=round((cb - ca) * 1e3, 3),
                                     vol_ctx_switches=rb.ru_nvcsw - ra.ru_nvcsw, invol_ctx_switches=rb.ru_nivcsw - ra.ru_nivcsw,
                                     minor_faults=rb.ru_minflt - ra.ru_minflt, major_faults=rb.ru_majflt - ra.ru_majflt, t0=a, t1=b))
    a, ca, ra = time.perf_counter(), time.thread_time(), resource.getrusage(RU)
    f.wait()
    b, cb, rb = time.perf_counter(), time.thread_time(), resource.getrusage(RU)
    if b - a > worst[0]:
        worst = (b - a, dict(call="wait", index=64, wall_ms=round((b - a) * 1e3, 3), thread_cpu_ms=round((cb - ca) * 1e3, 3),
                             vol_ctx_switches=rb.ru_nvcsw - ra.ru_nvcsw, invol_ctx_switches=rb.ru_nivcsw - ra.ru_nivcsw,
                             minor_faults=rb.ru_minflt - ra.ru_minflt, major_faults=rb.ru_majflt - ra.ru_majflt, t0=a, t1=b))
    w1, th1 = time.perf_counter(), threads_cpu()
    others = sorted(((round((th1[k] - th0.get(k, 0.0)) * 1e3, 1), k[1], k[0]) for k in th1 if th1[k] - th0.get(k, 0.0) > 0.002), reverse=True)
    w = worst[1]
    inside = [x for x in ticks if w["t0"] <= x <= w["t1"]]
    gaps = np.diff([w["t0"]] + inside + [w["t1"]]) if inside else np.array([w["t1"] - w["t0"]])
    w.pop("t0"); w.pop("t1")
    gcs = [dict(generation=g, ms=round((b_ - a_) * 1e3, 3), collected=c_) for a_, b_, g, c_ in gc_log if w0 <= a_ <= w1]
    report.append(dict(block=blk, gc_collections_in_burst=gcs, gc_objects_tracked=len(gc.get_objects()) if blk == 0 else None, burst_ms=round((w1 - w0) * 1e3, 2), main_thread_cpu_ms=round((time.thread_time() - cpu0) * 1e3, 2),
                       slowest_call=w, sampler_ticks_inside=len(inside), sampler_longest_gap_ms=round(float(gaps.max()) * 1e3, 2),
                       threads_cpu_ms=others[:8]))
stop = True
for r in report:
    print(json.dumps(r))
