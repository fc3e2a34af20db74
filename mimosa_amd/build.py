"""Build libmimosa_hip.so (HIP kernels + C ABI) for gfx950, in-tree.

hipcc cross-compiles without a GPU; the built .so travels to the GPU box with the snapshot
(git-ignored, not gpurun-ignored).  `python -m mimosa_amd.build` rebuilds.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmimosa_hip.so")
ARCH = "gfx950"

# (source, extra flags).  deskew_kernels.hip must not contract a*b+c into FMA: the reference's f32
# transforms are plain mul/add (baseline x86-64 build) and the results feed voxel hashing.
SOURCES = [
    ("icp_kernels.hip", []),
    ("deskew_kernels.hip", ["-ffp-contract=off"]),
    ("order_kernels.hip", []),
    ("scan_kernels.hip", ["-ffp-contract=off"]),
    ("mh_api.hip", []),
    ("map_kernels.hip", ["-ffp-contract=off"]),
    ("map_api.hip", []),
    ("shard_kernels.hip", []),
    ("shard_api.hip", []),
    ("photo_kernels.hip", ["-ffp-contract=off"]),
    ("photo_api.hip", []),
]
HEADERS = ["icp_device.hpp", "wave_dpp.hpp", "scan_device.hpp", "math3.hpp", "voxel_map.hpp", "voxel_group.hpp", "mh_internal.hpp", "photo_device.hpp", "map_device.hpp", "shard_device.hpp", "exact_sort.hpp", os.path.join("..", "..", "include", "mimosa_hip.h")]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _stale(out: str, deps: list[str]) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


class _BuildLock:
    """One builder at a time across processes (the ranks of a torchrun job all import this package): an flock on a file next
    to the outputs.  Whoever waited re-checks staleness afterwards and usually finds nothing left to do."""

    def __enter__(self):
        import fcntl
        os.makedirs(LIBDIR, exist_ok=True)
        self.f = open(os.path.join(LIBDIR, ".build.lock"), "w")
        fcntl.flock(self.f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl
        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()
        return False


def _run_to(cmd: list, out: str, verbose: bool = False):
    """Run a compiler command whose output path is `out`, into a temporary name, then rename: a concurrent reader (or a
    process that is executing the old file) never sees a half-written one."""
    tmp = f"{out}.tmp.{os.getpid()}"
    cmd = [tmp if c == out else c for c in cmd]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, out)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


def build(force: bool = False, verbose: bool = False, timeline: bool = False) -> str:
    """timeline=True builds the diagnostic variant libmimosa_hip_timeline.so (-DMH_TIMELINE:
    per-wave s_memtime stamps + mh_icp_timeline); never loaded by the product path."""
    with _BuildLock():
        return _build_locked(force, verbose, timeline)


def _deps_of(depfile: str) -> list[str] | None:
    """Prerequisites recorded by the compiler (-MMD) at the last build of an object; None when there is no record."""
    try:
        txt = open(depfile).read()
    except OSError:
        return None
    txt = txt.replace("\\\n", " ")
    if ":" not in txt:
        return None
    return [d for d in txt.split(":", 1)[1].split() if d]


# Tuning knobs of icp_kernels.hip that give CORRECT results (tools/variant.sh builds variants with them).  The product build
# takes none: hipcc appends HIPCC_COMPILE_FLAGS_APPEND / HIPCC_LINK_FLAGS_APPEND from the environment, so a stray -DMH_* there
# would silently change the shipped kernels — refused.  (Timing-only experiments that give WRONG results live as patches under
# tools/variants/, never as #ifdef branches of the product source.)
def _refuse_stray_defines() -> None:
    for var in ("HIPCC_COMPILE_FLAGS_APPEND", "HIPCC_LINK_FLAGS_APPEND", "CXXFLAGS", "CPPFLAGS", "HIP_CLANG_FLAGS"):
        val = os.environ.get(var, "")
        if "-DMH_" in val or "-D MH_" in val:
            raise RuntimeError(f"{var} carries an MH_* define ({val!r}): the product build takes no kernel knobs; use tools/variant.sh")


def _build_locked(force: bool, verbose: bool, timeline: bool) -> str:
    from concurrent.futures import ThreadPoolExecutor

    _refuse_stray_defines()

    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(HERE, "build_timeline" if timeline else "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, jobs = [], []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        d = o + ".d"
        objs.append(o)
        # an object is rebuilt when one of the files the compiler actually read for it changed (its -MMD record); without
        # a record, when the source or any header of the package changed
        deps = _deps_of(d)
        if deps is not None and not all(os.path.exists(x) for x in deps):
            deps = None
        if force or _stale(o, deps if deps is not None else [s] + hdrs):
            cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wextra", "-DMH_BUILDING_LIBRARY", "-MMD", "-MF", d,
                   "-Wno-unused-parameter", *extra, *(["-DMH_TIMELINE"] if timeline else []), "-c", s, "-o", o]
            jobs.append((cmd, o))
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            for f in [ex.submit(_run_to, cmd, o, verbose) for cmd, o in jobs]:
                f.result()
    lib = LIB.replace(".so", "_timeline.so") if timeline else LIB
    if force or _stale(lib, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib, *objs, "-ldl"]
        _run_to(cmd, lib, verbose)
    return lib


def build_replay_native(force: bool = False) -> str:
    """The native sequence-replay driver (host/replay_main.cpp over the header-only host mirror): plain g++, links the library."""
    lib = build()
    exe = os.path.join(LIBDIR, "replay_native")
    host = os.path.join(HERE, "host")
    src = os.path.join(host, "replay_main.cpp")
    deps = [src, lib] + [os.path.join(host, "mimosa_hip", h) for h in ("replay.hpp", "sharded_replay.hpp", "sharded.hpp", "manager.hpp", "binio.hpp", "photometric.hpp", "lidar.hpp", "types.hpp")]
    deps += [os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(host, "gtsam_sig")) for f in fs]
    with _BuildLock():
        if force or _stale(exe, deps):
            _run_to(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.dirname(HERE), "-I", host, "-I", os.path.join(host, "gtsam_sig"), src, "-o", exe,
                     "-L", LIBDIR, "-lmimosa_hip", "-lpthread", "-Wl,-rpath,$ORIGIN"], exe)
    return exe


def build_sync_caller(force: bool = False) -> str:
    """tools/micro/sync_caller.c (plain C, links nothing): times synchronous mh_icp_linearize calls from C for bench.py's latency
    leg and tools/sync_probe.py.  Measurement harness, never loaded by the product path."""
    src = os.path.join(os.path.dirname(HERE), "tools", "micro", "sync_caller.c")
    out = os.path.join(LIBDIR, "libmh_sync_caller.so")
    os.makedirs(LIBDIR, exist_ok=True)
    with _BuildLock():
        if force or _stale(out, [src]):
            _run_to(["gcc", "-O2", "-Wall", "-Wextra", "-shared", "-fPIC", src, "-o", out], out)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, timeline="--timeline" in sys.argv))
