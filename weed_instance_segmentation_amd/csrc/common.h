// Shared helpers for the libwm2f kernels (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/wm2f.h"

namespace wm2f {

constexpr int kWave = 64;
constexpr int kNumXcd = 8;  // MI355X: 8 XCDs, blocks are dealt round-robin over them

void set_error(const char* fmt, ...);

#define WM2F_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      ::wm2f::set_error(__VA_ARGS__);      \
      return WM2F_EINVAL;                  \
    }                                      \
  } while (0)

#define WM2F_CHECK_LAUNCH(name)                                               \
  do {                                                                        \
    hipError_t e__ = hipGetLastError();                                       \
    if (e__ != hipSuccess) {                                                  \
      ::wm2f::set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return WM2F_ELAUNCH;                                                    \
    }                                                                         \
  } while (0)

struct LevelInfo {
  int h[WM2F_MAX_LEVELS];
  int w[WM2F_MAX_LEVELS];
  int start[WM2F_MAX_LEVELS];  // first token of the level inside S
};

// Map a hardware block id to a logical block id so that blocks sharing an XCD (bid % 8 equal)
// get one CONTIGUOUS range of logical ids.  `per_xcd` = ceil(n_logical / 8); ids >= n_logical
// must be skipped by the caller.  Placement is a speed matter only, never correctness.
__device__ __forceinline__ int xcd_contiguous_id(int bid, int per_xcd) {
  return (bid % kNumXcd) * per_xcd + bid / kNumXcd;
}

__host__ __device__ __forceinline__ int ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ __forceinline__ int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace wm2f
