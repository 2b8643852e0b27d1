// Meta entry points, thread-local error message, per-launch HIP-event profiling (bench.py roofline).
#include "common.h"
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <vector>

namespace bg {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

struct ProfRec {
  std::string name;
  double flops, bytes, exec_flops, useful_flops;
  hipEvent_t e0, e1;
};
static bool g_prof = false;
static std::vector<ProfRec> g_recs;
static std::mutex g_mu;

bool prof_on() { return g_prof; }

void prof_begin(hipStream_t s, const char* name, double flops, double bytes) {
  std::lock_guard<std::mutex> lk(g_mu);
  ProfRec r;
  r.name = name;
  r.flops = flops;
  r.bytes = bytes;
  r.exec_flops = flops;          // until the launcher says otherwise (prof_exec_flops)
  r.useful_flops = flops;        // likewise (prof_useful_flops)
  (void)hipEventCreate(&r.e0);
  (void)hipEventCreate(&r.e1);
  (void)hipEventRecord(r.e0, s);
  g_recs.push_back(r);
}

void prof_exec_flops(double f) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_recs.empty()) g_recs.back().exec_flops = f;
}

void prof_useful_flops(double f) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_recs.empty()) g_recs.back().useful_flops = f;
}

static thread_local double g_pending_useful = -1.0;
void set_pending_useful(double f) { g_pending_useful = f; }
double take_pending_useful() {
  const double f = g_pending_useful;
  g_pending_useful = -1.0;
  return f;
}

void prof_end(hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_recs.empty()) (void)hipEventRecord(g_recs.back().e1, s);
}

}  // namespace bg

extern "C" {

int bg_version(void) { return BG_ABI_VERSION; }

const char* bg_last_error(void) { return bg::g_err; }

const char* bg_status_string(int s) {
  switch (s) {
    case BG_OK: return "ok";
    case BG_ERR_BAD_SHAPE: return "bad shape";
    case BG_ERR_BAD_ALIGNMENT: return "bad alignment";
    case BG_ERR_UNSUPPORTED: return "unsupported configuration";
    case BG_ERR_HIP: return "HIP runtime error";
    case BG_ERR_WORKSPACE: return "workspace missing or too small";
    case BG_ERR_NULL: return "null pointer";
    case BG_ERR_RCCL: return "RCCL error";
    default: return "unknown status";
  }
}

int bg_prof_enable(int on) {
  bg::g_prof = on != 0;
  return BG_OK;
}

int bg_prof_reset(void) {
  std::lock_guard<std::mutex> lk(bg::g_mu);
  for (auto& r : bg::g_recs) {
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  bg::g_recs.clear();
  return BG_OK;
}

int bg_prof_count(void) {
  std::lock_guard<std::mutex> lk(bg::g_mu);
  if (!bg::g_recs.empty()) (void)hipEventSynchronize(bg::g_recs.back().e1);
  return (int)bg::g_recs.size();
}

int bg_prof_get(int i, char* name, int name_cap, float* ms, double* flops, double* bytes) {
  std::lock_guard<std::mutex> lk(bg::g_mu);
  if (i < 0 || i >= (int)bg::g_recs.size()) return bg::fail(BG_ERR_BAD_SHAPE, "bg_prof_get: index %d out of range", i);
  auto& r = bg::g_recs[i];
  if (name && name_cap > 0) {
    strncpy(name, r.name.c_str(), name_cap - 1);
    name[name_cap - 1] = 0;
  }
  float t = 0;
  hipError_t e = hipEventElapsedTime(&t, r.e0, r.e1);
  if (e != hipSuccess) return bg::fail(BG_ERR_HIP, "bg_prof_get: %s", hipGetErrorString(e));
  if (ms) *ms = t;
  if (flops) *flops = r.flops;
  if (bytes) *bytes = r.bytes;
  return BG_OK;
}

// ---- roctx ranges (rocprofv3 --marker-trace); the library is looked up at run time, never linked
namespace {
typedef int (*roctx_push_t)(const char*);
typedef int (*roctx_pop_t)(void);
roctx_push_t g_rpush = nullptr;
roctx_pop_t g_rpop = nullptr;
int g_range_on = 0;
}  // namespace

int bg_range_enable(int on) {
  g_range_on = 0;
  if (!on) return 0;
  if (!g_rpush) {
    for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
      void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (!h) continue;
      g_rpush = reinterpret_cast<roctx_push_t>(dlsym(h, "roctxRangePushA"));
      g_rpop = reinterpret_cast<roctx_pop_t>(dlsym(h, "roctxRangePop"));
      if (g_rpush && g_rpop) break;
      g_rpush = nullptr; g_rpop = nullptr;
    }
  }
  g_range_on = (g_rpush && g_rpop) ? 1 : 0;
  return g_range_on;
}

int bg_range_push(const char* name) {
  if (g_range_on && name) (void)g_rpush(name);
  return BG_OK;
}

int bg_range_pop(void) {
  if (g_range_on) (void)g_rpop();
  return BG_OK;
}

int bg_prof_get_exec(int i, double* exec_flops) {
  std::lock_guard<std::mutex> lk(bg::g_mu);
  if (i < 0 || i >= (int)bg::g_recs.size()) return bg::fail(BG_ERR_BAD_SHAPE, "bg_prof_get_exec: index %d out of range", i);
  if (exec_flops) *exec_flops = bg::g_recs[i].exec_flops;
  return BG_OK;
}

int bg_prof_get_useful(int i, double* useful_flops) {
  std::lock_guard<std::mutex> lk(bg::g_mu);
  if (i < 0 || i >= (int)bg::g_recs.size()) return bg::fail(BG_ERR_BAD_SHAPE, "bg_prof_get_useful: index %d out of range", i);
  if (useful_flops) *useful_flops = bg::g_recs[i].useful_flops;
  return BG_OK;
}

double bg_conv2d_useful_flops(int B, int H, int W, int Cin, int Cout, int ksize, int stride) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || ksize < 1 || stride < 1) return 0.0;
  return bg::conv_useful_flops(B, H, W, Cin, Cout, ksize, stride);
}

}  // extern "C"
