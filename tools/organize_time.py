import sys, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from mimosa_amd import capi
import test_point_types as tp
ctx = capi.Context(0)
raw = tp.make_sensor_scan("hesai", rows=128, cols=1024, order="col")
L = capi.point_layout("hesai"); cfg = capi.make_input_config(point_skip_divisor=4)
sc = capi.Scan(ctx)
for org in (False, True):
    ts = []
    for _ in range(30):
        ctx.synchronize(); a = time.perf_counter()
        sc.prepare_input_layout(raw, L, cfg, header_ts=tp.HEADER_TS, organize_by_ring=org)
        ts.append(time.perf_counter() - a)
    print("organize", org, "median ms", np.median(ts) * 1e3, "min", np.min(ts) * 1e3)
raw0 = tp.make_sensor_scan("ouster", rows=128, cols=1024)
ts = []
for _ in range(30):
    ctx.synchronize(); a = time.perf_counter(); sc.prepare_input(raw0, cfg); ts.append(time.perf_counter() - a)
print("ouster direct median ms", np.median(ts) * 1e3)
