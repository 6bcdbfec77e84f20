// Shared device/host helpers for libevmi_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <string>

#include "evmi.h"

namespace evmi {

// ---- error plumbing -----------------------------------------------------------------------
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define EVMI_HIP_CHECK(expr)                                                                   \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      return ::evmi::fail(EVMI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));    \
    }                                                                                          \
  } while (0)

#define EVMI_LAUNCH_CHECK(what)                                                                \
  do {                                                                                         \
    hipError_t _e = hipGetLastError();                                                         \
    if (_e != hipSuccess) {                                                                    \
      return ::evmi::fail(EVMI_ERR_HIP, std::string(what) + " launch: " + hipGetErrorString(_e)); \
    }                                                                                          \
  } while (0)

// ---- bf16 vectors ---------------------------------------------------------------------------
typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

// host-side round-to-nearest-even fp32 -> bf16 bits (NaN kept quiet)
inline uint16_t f32_to_bf16_bits(float f) {
  uint32_t u;
  __builtin_memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

}  // namespace evmi
