// Host-side error plumbing shared by every translation unit of libmer_hip.so.
#include "common.h"
#include <string.h>
#include <atomic>
#include <mutex>
#include <vector>

namespace mer {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return MER_ELAUNCH;
  }
  return MER_OK;
}

// multiprocessor count of the current device (the persistent GEMM launches one workgroup per CU); cached per device
int device_cu_count() {
  static std::atomic<int> cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  int v = cached[dev].load(std::memory_order_relaxed);
  if (v > 0) return v;
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  cached[dev].store(n, std::memory_order_relaxed);
  return n;
}

struct ProfRec { const char* name; double flops, bytes; hipEvent_t e0, e1; };
static std::mutex g_prof_mu;
static std::vector<ProfRec*> g_prof;
static std::atomic<int> g_prof_on{0};

ProfScope::ProfScope(const char* name, double flops, double bytes, hipStream_t stream) : rec(nullptr), st(stream) {
  if (!g_prof_on.load(std::memory_order_relaxed)) return;
  ProfRec* r = new ProfRec{name, flops, bytes, nullptr, nullptr};
  if (hipEventCreate(&r->e0) != hipSuccess || hipEventCreate(&r->e1) != hipSuccess) { delete r; return; }
  (void)hipEventRecord(r->e0, st);
  rec = r;
}
ProfScope::~ProfScope() {
  if (!rec) return;
  ProfRec* r = (ProfRec*)rec;
  (void)hipEventRecord(r->e1, st);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof.push_back(r);
}
}  // namespace mer

extern "C" int mer_prof_enable(int on) {
  mer::g_prof_on.store(on ? 1 : 0);
  return MER_OK;
}

// Writes a JSON array [{"name":..,"calls":..,"ms":..,"flops":..,"bytes":..},..] aggregated per kernel
// name over everything recorded since the last report, and clears the records.  Synchronises the device.
extern "C" int mer_prof_report(char* buf, int buflen) {
  using namespace mer;
  if (!buf || buflen < 4) return MER_EINVAL;
  (void)hipDeviceSynchronize();
  std::lock_guard<std::mutex> lk(g_prof_mu);
  struct Agg { const char* name; long long calls; double ms, flops, bytes; };
  std::vector<Agg> aggs;
  for (ProfRec* r : g_prof) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r->e0, r->e1);
    Agg* a = nullptr;
    for (auto& x : aggs) if (strcmp(x.name, r->name) == 0) { a = &x; break; }
    if (!a) { aggs.push_back(Agg{r->name, 0, 0, 0, 0}); a = &aggs.back(); }
    a->calls++; a->ms += ms; a->flops += r->flops; a->bytes += r->bytes;
    (void)hipEventDestroy(r->e0); (void)hipEventDestroy(r->e1);
    delete r;
  }
  g_prof.clear();
  int off = snprintf(buf, buflen, "[");
  for (size_t i = 0; i < aggs.size() && off < buflen - 160; ++i)
    off += snprintf(buf + off, buflen - off, "%s{\"name\":\"%s\",\"calls\":%lld,\"ms\":%.6f,\"flops\":%.6e,\"bytes\":%.6e}",
                    i ? "," : "", aggs[i].name, aggs[i].calls, aggs[i].ms, aggs[i].flops, aggs[i].bytes);
  snprintf(buf + off, buflen - off, "]");
  return MER_OK;
}

extern "C" const char* mer_last_error(void) { return mer::g_err; }
extern "C" const char* mer_version(void) { return "mer_hip 0.1.0 (gfx950)"; }
extern "C" const char* mer_target_arch(void) { return "gfx950"; }
