// Shared host-side plumbing for libbgan_hip.so: status/error strings, launch wrapper, profiling hooks.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include "../../include/bgan.h"
#include "launch.h"

namespace bg {

void set_error(const char* fmt, ...);
bool prof_on();
void prof_begin(hipStream_t s, const char* name, double flops, double bytes);
void prof_end(hipStream_t s);
void prof_exec_flops(double f);      // flops the launch ISSUES on the matrix pipe (tile padding in, skipped padding taps out)
void prof_useful_flops(double f);    // flops of the launch that multiply real data: no zero-padding taps, no tile padding
// The conv entry points announce the USEFUL flop count of the call (conv_useful_flops) before they dispatch; the first launch
// bracket of the call that carries flops takes it (every conv call has exactly one such launch; its reduce passes carry none).
void set_pending_useful(double f);   // < 0 clears
double take_pending_useful();        // returns and clears; < 0 when nothing is pending

// Exact multiply-accumulates x 2 of a k x k / stride-s SAME convolution over real data (SURVEY 8d's F_l charges every tap at every
// output position; the taps that fall on TF's zero padding multiply nothing).  The same (output pixel, tap) pairs are touched by
// the forward, the data gradient / transposed convolution and the filter gradient of the geometry, so one count serves all three.
// H, W, Cin: conv input side (for a Conv2DTranspose: its OUTPUT side).
inline double conv_useful_flops(int B, int H, int W, int Cin, int Cout, int k, int s) {
  auto live = [&](int n) {
    const int o = (n + s - 1) / s;
    int tot = (o - 1) * s + k - n;
    if (tot < 0) tot = 0;
    const int before = tot / 2;
    long c = 0;
    for (int i = 0; i < o; ++i)
      for (int t = 0; t < k; ++t) c += ((unsigned)(i * s + t - before) < (unsigned)n) ? 1 : 0;
    return c;
  };
  return 2.0 * B * (double)Cin * Cout * (double)live(H) * (double)live(W);
}

struct UsefulScope {                 // RAII: a conv entry point's announcement never outlives the call
  explicit UsefulScope(double f) { set_pending_useful(f); }
  ~UsefulScope() { set_pending_useful(-1.0); }
};

inline int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  set_error("%s", buf);
  return code;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// Scoped launch bracket: records profiling events, checks the launch.  While a step program is being recorded the bracket
// leaves its name / flops / bytes in the program as well, so that a replay under bg_prof_enable(1) produces the same records.
struct Launch {
  hipStream_t s;
  bool prof;          // the launcher should compute its metadata (profiling on, or a program is being recorded)
  bool live;          // events are recorded around this launch now
  bool rec;
  Launch(void* stream, const char* name, double flops = 0, double bytes = 0)
      : s(static_cast<hipStream_t>(stream)), live(prof_on()), rec(recording()) {
    prof = live || rec;
    if (live) prof_begin(s, name, flops, bytes);
    if (rec) rec_note(1, name, flops, bytes);
    if (prof && flops > 0) {
      const double u = take_pending_useful();
      if (u >= 0) useful_flops(u);
    }
  }
  void useful_flops(double f) {
    if (live) prof_useful_flops(f);
    if (rec) rec_note(4, nullptr, f, 0);
  }
  void exec_flops(double f) {
    if (live) prof_exec_flops(f);
    if (rec) rec_note(3, nullptr, f, 0);
  }
  int done(const char* what) {
    hipError_t e = hipGetLastError();
    if (live) prof_end(s);
    if (rec) rec_note(2, nullptr, 0, 0);
    if (e != hipSuccess) return fail(BG_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return BG_OK;
  }
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per call site, safe under concurrent host threads (the header promises
// thread safety for distinct streams): the flag and the result live beside the call.  BG_LDS_ATTR_ONCE returns the error;
// BG_LDS_ATTR_ONCE_V ignores it (the launch that follows reports it).
#define BG_LDS_ATTR_ONCE(kern, bytes, what)                                                                                      \
  do {                                                                                                                           \
    static std::once_flag flag_;                                                                                                 \
    static hipError_t err_ = hipSuccess;                                                                                         \
    std::call_once(flag_, [&] { err_ = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); }); \
    if (err_ != hipSuccess) return bg::fail(BG_ERR_HIP, what ": hipFuncSetAttribute: %s", hipGetErrorString(err_));              \
  } while (0)
#define BG_LDS_ATTR_ONCE_V(kern, bytes)                                                                                          \
  do {                                                                                                                           \
    static std::once_flag flag_;                                                                                                 \
    std::call_once(flag_, [&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); }); \
  } while (0)

#define BG_REQUIRE(cond, code, ...) \
  do {                              \
    if (!(cond)) return bg::fail(code, __VA_ARGS__); \
  } while (0)

}  // namespace bg
