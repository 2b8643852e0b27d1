// Shared host-side plumbing for libbgan_hip.so: status/error strings, launch wrapper, profiling hooks.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include "../../include/bgan.h"

namespace bg {

void set_error(const char* fmt, ...);
bool prof_on();
void prof_begin(hipStream_t s, const char* name, double flops, double bytes);
void prof_end(hipStream_t s);
void prof_exec_flops(double f);      // flops the launch ISSUES on the matrix pipe (tile padding in, skipped padding taps out)

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

// Scoped launch bracket: clears stale errors, records profiling events, checks the launch.
struct Launch {
  hipStream_t s;
  bool prof;
  Launch(void* stream, const char* name, double flops = 0, double bytes = 0)
      : s(static_cast<hipStream_t>(stream)), prof(prof_on()) {
    if (prof) prof_begin(s, name, flops, bytes);
  }
  void exec_flops(double f) {
    if (prof) prof_exec_flops(f);
  }
  int done(const char* what) {
    hipError_t e = hipGetLastError();
    if (prof) prof_end(s);
    if (e != hipSuccess) return fail(BG_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return BG_OK;
  }
};

#define BG_REQUIRE(cond, code, ...) \
  do {                              \
    if (!(cond)) return bg::fail(code, __VA_ARGS__); \
  } while (0)

}  // namespace bg
