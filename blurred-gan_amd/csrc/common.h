// Shared host-side plumbing for libbgan_hip.so: status/error strings, launch wrapper, profiling hooks.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include "../../include/bgan.h"
#include "launch.h"

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

#define BG_REQUIRE(cond, code, ...) \
  do {                              \
    if (!(cond)) return bg::fail(code, __VA_ARGS__); \
  } while (0)

}  // namespace bg
