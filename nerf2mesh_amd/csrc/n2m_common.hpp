// Shared host/device plumbing for libn2m_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/n2m_hip.h"

// ------------------------------------------------------------------------------------------- error state
void n2m_set_error(const char* fmt, ...);

#define N2M_REQUIRE(cond, code, ...)      \
    do {                                  \
        if (!(cond)) {                    \
            n2m_set_error(__VA_ARGS__);   \
            return (code);                \
        }                                 \
    } while (0)

#define N2M_NOTNULL(p) N2M_REQUIRE((p) != nullptr, N2M_ENULL, "%s: %s is NULL", __func__, #p)

// Checks the launch that was just issued; returns the hipError_t as a positive int.
#define N2M_CHECK_LAUNCH()                                                              \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess) {                                                        \
            n2m_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__));   \
            return (int)e__;                                                            \
        }                                                                               \
    } while (0)

#define N2M_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t e__ = (call);                                                        \
        if (e__ != hipSuccess) {                                                        \
            n2m_set_error("%s: %s failed: %s", __func__, #call, hipGetErrorString(e__)); \
            return (int)e__;                                                            \
        }                                                                               \
    } while (0)

static inline uint32_t n2m_ceil_div(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// --------------------------------------------------------------------------------------------- profiling
// RAII bracket: records a hipEvent pair on `stream` around a launch when profiling is enabled.
struct N2mProfScope {
    int slot;
    hipStream_t stream;
    N2mProfScope(int kernel_id, hipStream_t s, double algo_bytes);
    ~N2mProfScope();
};
#define N2M_PROF(kernel_id, stream, bytes) N2mProfScope prof_scope__((kernel_id), (stream), (double)(bytes))

// ------------------------------------------------------------------------------------------- device bits
#define N2M_WAVE 64

__device__ __forceinline__ float n2m_clampf(float v, float lo, float hi) { return fminf(hi, fmaxf(lo, v)); }

// 10-bit-per-axis Morton code, x in bit 0.
__device__ __forceinline__ uint32_t n2m_spread3(uint32_t v) {
    // same bit moves as the multiply form (v*0x00010001 == v | v<<16 for non-overlapping fields), but every step is
    // one full-rate v_lshl_or_b32 + v_and instead of a quarter-rate v_mul_lo_u32
    v = (v | (v << 16)) & 0xFF0000FFu;
    v = (v | (v << 8)) & 0x0F00F00Fu;
    v = (v | (v << 4)) & 0xC30C30C3u;
    v = (v | (v << 2)) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t n2m_morton(uint32_t x, uint32_t y, uint32_t z) {
    return n2m_spread3(x) | (n2m_spread3(y) << 1) | (n2m_spread3(z) << 2);
}
__device__ __forceinline__ uint32_t n2m_gather3(uint32_t v) {
    v &= 0x49249249u;
    v = (v | (v >> 2)) & 0xC30C30C3u;
    v = (v | (v >> 4)) & 0x0F00F00Fu;
    v = (v | (v >> 8)) & 0xFF0000FFu;
    v = (v | (v >> 16)) & 0x0000FFFFu;
    return v;
}

// wave64 reductions / scans on shuffles (compiler lowers to DPP / ds_bpermute)
__device__ __forceinline__ float n2m_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t n2m_wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// inclusive prefix sum across the wave
__device__ __forceinline__ float n2m_wave_scan_add(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    return v;
}
__device__ __forceinline__ uint32_t n2m_wave_scan_add_u32(uint32_t v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    return v;
}
// inclusive prefix product across the wave
__device__ __forceinline__ float n2m_wave_scan_mul(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float u = __shfl_up(v, o, 64);
        if (lane >= o) v *= u;
    }
    return v;
}
