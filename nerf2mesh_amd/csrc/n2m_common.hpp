// Shared host/device plumbing for libn2m_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_ext.h>
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
    bool kernel_events;
    N2mProfScope(int kernel_id, hipStream_t s, double algo_bytes, bool kernel_events = false);
    ~N2mProfScope();
};
#define N2M_PROF(kernel_id, stream, bytes) N2mProfScope prof_scope__((kernel_id), (stream), (double)(bytes))
// N2M_PROF_K + N2M_LAUNCH: the event pair rides ON the kernel dispatches of the entry (hipExtLaunchKernel: start of the first kernel, end of
// the last) instead of in marker packets around them.  A marker pair serialises the queue and adds 10-30 us to what it brackets (round 4:
// 237 us by markers for a table backward whose two kernels take 203 us in the rocprofv3 trace of the same run); kernel-attached events
// read the dispatch's own timestamps, like the tracer does, and add no packet.  Every launch inside an N2M_PROF_K entry goes through
// N2M_LAUNCH (template kernels in parentheses).
#define N2M_PROF_K(kernel_id, stream, bytes) N2mProfScope prof_scope__((kernel_id), (stream), (double)(bytes), true)
struct N2mProfLaunchState { hipEvent_t a, b; int armed; };        // armed: 0 off, 1 the entry's first launch is still to come, 2 later launches, 3 markers
N2mProfLaunchState& n2m_prof_launch_state();
void n2m_prof_fall_back_to_markers(hipStream_t s);       // an N2M_PROF_K entry that ends up on plain launches: marker events from here on
#define N2M_LAUNCH(kernel, grid, block, shmem, stream, ...)                                                                         \
    do {                                                                                                                            \
        N2mProfLaunchState& pl__ = n2m_prof_launch_state();                                                                         \
        if (pl__.armed) {                                                                                                           \
            hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(block), (uint32_t)(shmem), (stream), pl__.armed == 1 ? pl__.a : nullptr, \
                                  pl__.b, 0, __VA_ARGS__);                                                                          \
            pl__.armed = 2;                                                                                                         \
        } else kernel<<<(grid), (block), (shmem), (stream)>>>(__VA_ARGS__);                                                         \
    } while (0)

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

// wave64 scans and reductions on the DPP network (no LDS-pipe round trips: a ds_bpermute-based Hillis-Steele scan is six dependent
// ~60-cycle trips, this is six VALU operations).  Inclusive scan: Hillis-Steele inside each row of 16 lanes (row_shr 1/2/4/8; lanes
// without a source keep the identity), then lane 15 of rows 0/2 into rows 1/3 (row_bcast:15) and lane 31 into rows 2/3 (row_bcast:31).
// Sums/products are associated in that order (fp results differ from a serial sum in the last bits, like any parallel reduction).
// PRECONDITION of every helper below (scans, n2m_wave_sum*, n2m_lane63, n2m_lane_below): ALL 64 LANES ACTIVE at the call -- wave-uniform
// control flow only.  A DPP source lane that is masked off leaves the identity in place (silently dropping its value from the scan) and
// v_readlane of lane 63 returns a stale register when that lane is inactive; the ds_bpermute helpers these replaced tolerated divergence.
// Callers mask with a select (`valid ? x : identity`), never with a branch around the call.
#define N2M_DPP_STEP(OP, ident, ctrl, rows) v = OP(v, __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (T)(ident)), __builtin_bit_cast(int, v), (ctrl), (rows), 0xf, false)))
template <typename T, class F>
__device__ __forceinline__ T n2m_wave_scan_dpp(T v, T ident, F OP) {
    N2M_DPP_STEP(OP, ident, 0x111, 0xf);   // row_shr:1
    N2M_DPP_STEP(OP, ident, 0x112, 0xf);   // row_shr:2
    N2M_DPP_STEP(OP, ident, 0x114, 0xf);   // row_shr:4
    N2M_DPP_STEP(OP, ident, 0x118, 0xf);   // row_shr:8
    N2M_DPP_STEP(OP, ident, 0x142, 0xa);   // row_bcast:15 -> rows 1, 3
    N2M_DPP_STEP(OP, ident, 0x143, 0xc);   // row_bcast:31 -> rows 2, 3
    return v;
}
#undef N2M_DPP_STEP
struct N2mAddF { __device__ __forceinline__ float operator()(float a, float b) const { return a + b; } };
struct N2mMulF { __device__ __forceinline__ float operator()(float a, float b) const { return a * b; } };
struct N2mAddU { __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return a + b; } };
// inclusive prefix sum / product across the wave (`lane` kept in the signature for the callers; unused)
__device__ __forceinline__ float n2m_wave_scan_add(float v, int) { return n2m_wave_scan_dpp<float>(v, 0.0f, N2mAddF()); }
__device__ __forceinline__ uint32_t n2m_wave_scan_add_u32(uint32_t v, int) { return n2m_wave_scan_dpp<uint32_t>(v, 0u, N2mAddU()); }
__device__ __forceinline__ float n2m_wave_scan_mul(float v, int) { return n2m_wave_scan_dpp<float>(v, 1.0f, N2mMulF()); }
// wave totals: lane 63 of the inclusive scan, broadcast through an SGPR
__device__ __forceinline__ float n2m_wave_sum(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, n2m_wave_scan_add(v, 0)), 63));
}
__device__ __forceinline__ uint32_t n2m_wave_sum_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane((int)n2m_wave_scan_add_u32(v, 0), 63);
}
// value of lane 63 in every lane (through an SGPR) / value of the lane below, `first` in lane 0 (wave_shr:1)
__device__ __forceinline__ float n2m_lane63(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63)); }
__device__ __forceinline__ float n2m_lane_below(float v, float first) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, first), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
