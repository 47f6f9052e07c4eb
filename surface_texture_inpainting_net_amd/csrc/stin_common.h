// Shared helpers for the libstin_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/stin_hip.h"

#define STIN_WAVE 64

#define STIN_REQUIRE(cond, code) \
    do {                         \
        if (!(cond)) return (code); \
    } while (0)

// hipGetLastError() is per-thread sticky state shared with every other HIP user in the process
// (PyTorch included): drop whatever an earlier, unrelated call left behind before we launch.
static inline void stin_clear_stale_error() { (void)hipGetLastError(); }

static inline int stin_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? STIN_OK : (int)e;
}

static inline bool stin_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Lanes that cooperate on one feature row: smallest power of two >= ceil(C/4), capped at a wave.
static inline int stin_group_lanes(int c4) {
    int g = 1;
    while (g < c4 && g < STIN_WAVE) g <<= 1;
    return g;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
