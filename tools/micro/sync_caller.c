/* What a C or C++ caller pays for one synchronous mh_icp_linearize: the calls are made from C with CLOCK_MONOTONIC around the
 * foreign call alone (a ctypes call adds 1-2 us of argument conversion on top).  Built by mimosa_amd/build.py
 * (build_sync_caller) into mimosa_amd/lib/libmh_sync_caller.so; loaded by bench.py's latency leg and tools/sync_probe.py only.
 * The library's entry points are passed in as function pointers: this file links against nothing. */
#include <stddef.h>
#include <time.h>

typedef int (*lin_fn)(void * icp, const double * R, const double * t, const double * Rt, const double * tt, const double * g, void * out);
typedef int (*reset_fn)(void * icp);
typedef int (*sync_fn)(void * ctx);

static double now_ns(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec * 1e9 + (double)ts.tv_nsec;
}

/* n calls; before each: reset (when `reset` != 0: a cold call) and a stream synchronisation.  ns_out[i] = wall time of call i.
 * Returns the first non-zero status of any call, 0 otherwise. */
int mh_sync_caller_run(void * lin, void * rst, void * syn, void * ctx, void * icp, const double * R, const double * t, const double * g,
                       void * out, int n, int reset, double * ns_out)
{
  for (int i = 0; i < n; ++i) {
    int rc = 0;
    if (reset) rc = ((reset_fn)rst)(icp);
    if (!rc) rc = ((sync_fn)syn)(ctx);
    if (rc) return rc;
    const double a = now_ns();
    rc = ((lin_fn)lin)(icp, R, t, NULL, NULL, g, out);
    const double b = now_ns();
    if (rc) return rc;
    ns_out[i] = b - a;
  }
  return 0;
}

/* The timed region of bench.py: n COLD synchronous calls back to back — mh_icp_reset, then mh_icp_linearize, whose return means
 * the result is on the host (in out + i * out_stride; stride 0 = one slot) — with nothing between them: what a C++ caller that
 * linearizes one factor at a time pays per step (geometric.cpp:194-196).  Returns the first non-zero status. */
int mh_sync_caller_steps(void * lin, void * rst, void * icp, const double * R, const double * t, const double * g, char * out,
                         size_t out_stride, int n)
{
  for (int i = 0; i < n; ++i) {
    int rc = ((reset_fn)rst)(icp);
    if (!rc) rc = ((lin_fn)lin)(icp, R, t, NULL, NULL, g, out + (size_t)i * out_stride);
    if (rc) return rc;
  }
  return 0;
}
