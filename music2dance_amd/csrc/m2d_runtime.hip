// m2d runtime: thread-local error string, version, and the event profiler.
// Nothing here touches tensors; it exists so that every other translation unit can
// report failures across the C-ABI without throwing and so that bench.py can time
// one kernel family with HIP events on the stream the kernels are launched on.
#include "m2d_common.h"
#include <stdarg.h>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";

void m2d_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

struct ProfRec {
  hipEvent_t a, b;
  int fam;
  double flops, bytes;
  char tag[40];
  int d[3];
};

static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof_recs;      // used records of the current session
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_pool;  // recycled events

M2dProfScope::M2dProfScope(int family, hipStream_t s, double flops, double bytes, const char* tag, int d0, int d1,
                           int d2)
    : fam(family), stream(s), slot(-1) {
  if (!g_prof_on) return;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRec r;
  if (!g_prof_pool.empty()) {
    r.a = g_prof_pool.back().first;
    r.b = g_prof_pool.back().second;
    g_prof_pool.pop_back();
  } else {
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
  }
  r.fam = family;
  r.flops = flops;
  r.bytes = bytes;
  snprintf(r.tag, sizeof(r.tag), "%s", tag ? tag : "");
  r.d[0] = d0;
  r.d[1] = d1;
  r.d[2] = d2;
  hipEventRecord(r.a, s);
  g_prof_recs.push_back(r);
  slot = (int)g_prof_recs.size() - 1;
}

M2dProfScope::~M2dProfScope() {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  hipEventRecord(g_prof_recs[slot].b, stream);
}

extern "C" {

const char* m2d_last_error(void) { return g_err; }

int m2d_version(void) { return 100; }

// Start collecting per-launch HIP-event timings (one event pair per kernel launch of
// every family, recorded on the launch stream).
int m2d_prof_begin(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof_recs) g_prof_pool.push_back({r.a, r.b});
  g_prof_recs.clear();
  g_prof_on = true;
  return M2D_OK;
}

// Per-launch dump of the records collected since m2d_prof_begin (call BEFORE m2d_prof_end):
// one line "family,tag,d0,d1,d2,ms,flops,bytes" per launch. Returns the number of bytes written.
int m2d_prof_dump(char* buf, int cap) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  int n = 0;
  for (auto& r : g_prof_recs) {
    if (hipEventSynchronize(r.b) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
    if (cap - n < 160) break;
    n += snprintf(buf + n, cap - n, "%d,%s,%d,%d,%d,%.6f,%.0f,%.0f\n", r.fam, r.tag, r.d[0], r.d[1], r.d[2], ms, r.flops,
                  r.bytes);
  }
  return n;
}

// Stop collecting; out[f*4 + {0,1,2,3}] = {total ms, launches, algorithmic flops,
// algorithmic bytes} for family f (M2D_FAM_COUNT families). Synchronises the events.
int m2d_prof_end(double* out, int n_out) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = false;
  if (n_out < M2D_FAM_COUNT * 4) M2D_FAIL(M2D_ERR_ARG, "m2d_prof_end: need %d doubles", M2D_FAM_COUNT * 4);
  for (int i = 0; i < M2D_FAM_COUNT * 4; ++i) out[i] = 0.0;
  for (auto& r : g_prof_recs) {
    if (hipEventSynchronize(r.b) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
    out[r.fam * 4 + 0] += ms;
    out[r.fam * 4 + 1] += 1.0;
    out[r.fam * 4 + 2] += r.flops;
    out[r.fam * 4 + 3] += r.bytes;
  }
  for (auto& r : g_prof_recs) g_prof_pool.push_back({r.a, r.b});
  g_prof_recs.clear();
  return M2D_OK;
}

}  // extern "C"
