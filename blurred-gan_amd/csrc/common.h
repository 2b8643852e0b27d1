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
