// Error plumbing and ABI version of libgpmp_hip.so.
#include "common.h"
#include <cstdarg>
#include <cstdint>
#include <vector>
#include <cstdlib>
#include <mutex>

namespace gpmp {
namespace {
thread_local char g_err[512] = "";
}
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int hip_fail(hipError_t e, const char* what) {
  set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
  return -1000 - (int)e;
}
int device_cu_count() {
  static std::atomic<int> cache[64];           // per device ordinal; 0 = not asked yet (two threads racing both ask: same answer)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int v = cache[dev].load(std::memory_order_relaxed);
  if (v > 0) return v;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
  cache[dev].store(v, std::memory_order_relaxed);
  return v;
}
}  // namespace gpmp

namespace gpmp {
bool g_prof_on = false;
unsigned g_prof_mask = 0xFFFFFFFFu;
namespace {
struct Rec { hipEvent_t a, b; int kind; double work; };
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t g_open[PK_COUNT];
std::mutex g_prof_mu;      // the records are one table per process: profiling is a diagnostic of ONE device at a time
hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e; (void)hipEventCreate(&e); return e;
}
}  // namespace
void prof_start(int kind, hipStream_t st) {
  if (!((g_prof_mask >> kind) & 1u)) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  hipEvent_t e = get_event();
  (void)hipEventRecord(e, st);
  g_open[kind] = e;
}
void prof_stop(int kind, hipStream_t st, double work) {
  if (!((g_prof_mask >> kind) & 1u)) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  hipEvent_t e = get_event();
  (void)hipEventRecord(e, st);
  g_recs.push_back({g_open[kind], e, kind, work});
}
}  // namespace gpmp

extern "C" int gpmp_profile_begin_kinds(unsigned kinds) {
  std::lock_guard<std::mutex> lk(gpmp::g_prof_mu);
  for (auto& r : gpmp::g_recs) { gpmp::g_pool.push_back(r.a); gpmp::g_pool.push_back(r.b); }
  gpmp::g_recs.clear();
  gpmp::g_prof_mask = kinds;
  gpmp::g_prof_on = true;
  return 0;
}
extern "C" int gpmp_profile_begin(void) { return gpmp_profile_begin_kinds(0xFFFFFFFFu); }

extern "C" int gpmp_profile_end(double* table_host) {
  gpmp::g_prof_on = false;
  if (table_host == nullptr) return -1;
  std::lock_guard<std::mutex> lk(gpmp::g_prof_mu);
  for (int i = 0; i < 3 * gpmp::PK_COUNT; ++i) table_host[i] = 0.0;
  FILE* dump = nullptr;
  if (const char* path = getenv("GPMP_PROF_DUMP")) dump = fopen(path, "w");
  if (dump) fprintf(dump, "kind,work,ms\n");
  for (auto& r : gpmp::g_recs) {
    if (hipEventSynchronize(r.b) != hipSuccess) return gpmp::hip_fail(hipGetLastError(), "hipEventSynchronize");
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return gpmp::hip_fail(hipGetLastError(), "hipEventElapsedTime");
    table_host[3 * r.kind + 0] += 1.0;
    table_host[3 * r.kind + 1] += (double)ms;
    table_host[3 * r.kind + 2] += r.work;
    if (dump) fprintf(dump, "%d,%.6e,%.6f\n", r.kind, r.work, ms);
  }
  if (dump) fclose(dump);
  for (auto& r : gpmp::g_recs) { gpmp::g_pool.push_back(r.a); gpmp::g_pool.push_back(r.b); }
  gpmp::g_recs.clear();
  return 0;
}

// Stream whose kernels may use every CU except `reserve_cus` of them (spread over the XCDs: the driver deals the bits of
// a CU mask round-robin over XCDs and shader engines, so clearing the lowest bits takes one CU per XCD first).
extern "C" int gpmp_stream_create_reserving_cus(int reserve_cus, gpmp_stream_t* stream_out) {
  GPMP_ARG(stream_out != nullptr, 2, "stream_out is NULL");
  int dev = 0, ncu = 0;
  GPMP_HIP_TRY(hipGetDevice(&dev));
  GPMP_HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
  GPMP_ARG(reserve_cus >= 0 && reserve_cus < ncu, 1, "reserve_cus outside [0, number of CUs)");
  const int words = (ncu + 31) / 32;
  std::vector<uint32_t> mask(words, 0u);
  for (int i = reserve_cus; i < ncu; ++i) mask[i / 32] |= (1u << (i % 32));
  hipStream_t st = nullptr;
  GPMP_HIP_TRY(hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask.data()));
  *stream_out = (gpmp_stream_t)st;
  return 0;
}
extern "C" int gpmp_stream_release(gpmp_stream_t stream);
extern "C" int gpmp_stream_destroy(gpmp_stream_t stream) {
  if (stream == nullptr) return 0;
  if (int rc = gpmp_stream_release(stream)) return rc;      // the per-stream state of the one-launch solve goes with it
  GPMP_HIP_TRY(hipStreamDestroy((hipStream_t)stream));
  return 0;
}

namespace gpmp { extern int g_machine_busy; }
extern "C" int gpmp_hint_machine_busy(int on) {
  const int prev = gpmp::g_machine_busy;
  gpmp::g_machine_busy = on ? 1 : 0;
  return prev;
}

extern "C" int gpmp_hip_abi_version(void) { return 1; }
extern "C" const char* gpmp_last_error(void) { return gpmp::g_err; }
