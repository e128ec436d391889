// Peer-store exchange of the sharded optimizer -- include/n2m_peer.h (no reference counterpart: the reference's data parallelism is DDP's
// all-reduce, nerf/utils.py:517-519).  Memory that other processes map (hipIpc*), epoch flags with system-scope release / acquire, a bounded
// spin so that a missing peer costs a timeout and not the GPU, the owner-side sum of the W gradient slots in rank order, and the one-to-all
// store of refreshed rows.  Built and tested between two processes on one GPU; not run over xGMI (see the header).
#include <cstring>
#include "n2m_common.hpp"
#include "../../include/n2m_peer.h"

namespace {

struct PtrsK { void* ptr[N2M_PEER_MAX]; uint32_t count; };

__global__ void peer_signal_kernel(PtrsK flags, uint32_t value) {
    const uint32_t i = threadIdx.x;
    if (i >= flags.count) return;
    // the kernel boundary in front of this launch has completed (and written back) everything the stream did before; the store itself is a
    // system-scope release so that it cannot pass anything either
    __hip_atomic_store(reinterpret_cast<uint32_t*>(flags.ptr[i]), value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void peer_wait_kernel(const uint32_t* __restrict__ flags, uint32_t count, uint32_t stride, uint32_t value, unsigned long long ticks,
                                 uint32_t* __restrict__ error) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    for (uint32_t i = 0; i < count; ++i) {
        const uint32_t* f = flags + (size_t)i * stride;
        while ((int32_t)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - value) < 0) {
            if (wall_clock64() - t0 > ticks) {
                if (error) __hip_atomic_store(error, i + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return;
            }
            __builtin_amdgcn_s_sleep(64);
        }
    }
}

// one word per thread and destination: the form for ranges that start on a 4-byte boundary only (an odd number of rows per slot)
__global__ void __launch_bounds__(256)
peer_copy_words_kernel(const uint32_t* __restrict__ src, PtrsK dst, size_t words) {
    const size_t w = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (w >= words) return;
    const uint32_t v = src ? src[w] : 0u;
    for (uint32_t d = 0; d < dst.count; ++d) reinterpret_cast<uint32_t*>(dst.ptr[d])[w] = v;
}

// 16 bytes per thread and destination; bytes % 16 handled by the tail threads in words
__global__ void __launch_bounds__(256)
peer_copy_kernel(const uint32_t* __restrict__ src, PtrsK dst, size_t words) {
    const size_t q = (size_t)blockIdx.x * 256u + threadIdx.x, w0 = q * 4u;
    if (w0 >= words) return;
    if (w0 + 4u <= words) {
        const uint4 v = src ? reinterpret_cast<const uint4*>(src)[q] : make_uint4(0u, 0u, 0u, 0u);
        for (uint32_t d = 0; d < dst.count; ++d) reinterpret_cast<uint4*>(dst.ptr[d])[q] = v;
    } else {
        for (size_t w = w0; w < words; ++w) {
            const uint32_t v = src ? src[w] : 0u;
            for (uint32_t d = 0; d < dst.count; ++d) reinterpret_cast<uint32_t*>(dst.ptr[d])[w] = v;
        }
    }
}

// one thread per row: a 4-byte load per slot and table (fp32 value / fp16 pair) that bypasses the caches -- the slots were written by other
// agents.  (Rows per slot may be odd -- the coarse half at 8 ranks -- so wider loads would misalign every other slot.)
__global__ void __launch_bounds__(256)
peer_reduce_kernel(const float* __restrict__ stage1, const _Float16* __restrict__ stage2, uint32_t world, uint32_t rows, float* __restrict__ g1,
                   _Float16* __restrict__ g2, float* __restrict__ found_inf) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= rows) return;
    auto load32 = [](const void* at) { return __hip_atomic_load(reinterpret_cast<const uint32_t*>(at), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); };
    bool bad = false;
    if (g1) {
        float a = 0.0f;
        for (uint32_t s = 0; s < world; ++s) a += __uint_as_float(load32(stage1 + (size_t)s * rows + r));
        bad |= !(fabsf(a) <= 3.0e38f);
        g1[r] = a;
    }
    if (g2) {
        float a = 0.0f, b = 0.0f;
        for (uint32_t s = 0; s < world; ++s) {
            const h2 p = __builtin_bit_cast(h2, load32(stage2 + ((size_t)s * rows + r) * 2u));
            a += (float)p.x; b += (float)p.y;
        }
        bad |= !(fabsf(a) <= 65504.0f) || !(fabsf(b) <= 65504.0f);
        h2 o;
        o.x = (_Float16)a; o.y = (_Float16)b;
        reinterpret_cast<h2*>(g2)[r] = o;
    }
    if (bad && found_inf) *found_inf = 1.0f;
}

PtrsK to_k(const N2mPeerPtrs* p) {
    PtrsK k{};
    k.count = p->count;
    for (uint32_t i = 0; i < p->count && i < N2M_PEER_MAX; ++i) k.ptr[i] = p->ptr[i];
    return k;
}

}  // namespace

extern "C" int n2m_peer_alloc(size_t bytes, int fine_grained, void** out) {
    N2M_REQUIRE(out && bytes > 0, N2M_EINVAL, "peer_alloc: bytes > 0 and a place for the pointer");
    void* p = nullptr;
    if (fine_grained) N2M_HIP(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained));
    else N2M_HIP(hipMalloc(&p, bytes));
    N2M_HIP(hipMemset(p, 0, bytes));
    N2M_HIP(hipDeviceSynchronize());
    *out = p;
    return 0;
}

extern "C" int n2m_peer_free(void* ptr) {
    if (ptr) N2M_HIP(hipFree(ptr));
    return 0;
}

extern "C" int n2m_peer_export(void* ptr, void* handle) {
    N2M_REQUIRE(ptr && handle, N2M_ENULL, "peer_export: NULL");
    static_assert(sizeof(hipIpcMemHandle_t) == N2M_PEER_HANDLE_BYTES, "handle size");
    hipIpcMemHandle_t h;
    N2M_HIP(hipIpcGetMemHandle(&h, ptr));
    std::memcpy(handle, &h, sizeof(h));
    return 0;
}

extern "C" int n2m_peer_import(const void* handle, void** out) {
    N2M_REQUIRE(handle && out, N2M_ENULL, "peer_import: NULL");
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle, sizeof(h));
    void* p = nullptr;
    N2M_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    *out = p;
    return 0;
}

extern "C" int n2m_peer_unmap(void* ptr) {
    if (ptr) N2M_HIP(hipIpcCloseMemHandle(ptr));
    return 0;
}

extern "C" int n2m_peer_signal(const N2mPeerPtrs* flags, uint32_t value, void* stream) {
    N2M_REQUIRE(flags && flags->count >= 1 && flags->count <= N2M_PEER_MAX, N2M_EINVAL, "peer_signal: 1..%d flags", N2M_PEER_MAX);
    for (uint32_t i = 0; i < flags->count; ++i) N2M_REQUIRE(flags->ptr[i], N2M_ENULL, "peer_signal: NULL flag %u", i);
    peer_signal_kernel<<<1, 64, 0, (hipStream_t)stream>>>(to_k(flags), value);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_peer_wait(const uint32_t* flags, uint32_t count, uint32_t stride_words, uint32_t value, uint32_t timeout_ms, uint32_t* error,
                             void* stream) {
    N2M_REQUIRE(flags && count >= 1 && stride_words >= 1, N2M_EINVAL, "peer_wait: flags, count >= 1, stride >= 1");
    static int khz = 0;
    if (!khz) {
        int dev = 0;
        N2M_HIP(hipGetDevice(&dev));
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;      // 100 MHz
    }
    const unsigned long long ticks = (unsigned long long)(timeout_ms ? timeout_ms : 10000u) * (unsigned long long)khz;
    peer_wait_kernel<<<1, 64, 0, (hipStream_t)stream>>>(flags, count, stride_words, value, ticks, error);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_peer_copy(const void* src, const N2mPeerPtrs* dst, size_t bytes, void* stream) {
    N2M_REQUIRE(dst && dst->count <= N2M_PEER_MAX && bytes % 4 == 0, N2M_EINVAL, "peer_copy: up to %d destinations, bytes a multiple of 4", N2M_PEER_MAX);
    if (bytes == 0 || dst->count == 0) return 0;
    bool wide = ((uintptr_t)src & 15u) == 0;
    for (uint32_t i = 0; i < dst->count; ++i) {
        N2M_REQUIRE(dst->ptr[i] && ((uintptr_t)dst->ptr[i] & 3u) == 0, N2M_EINVAL, "peer_copy: destination %u NULL or not 4-byte aligned", i);
        wide = wide && ((uintptr_t)dst->ptr[i] & 15u) == 0;
    }
    N2M_REQUIRE(((uintptr_t)src & 3u) == 0, N2M_EINVAL, "peer_copy: source not 4-byte aligned");
    const size_t words = bytes / 4, quads = (words + 3) / 4;
    if (wide) peer_copy_kernel<<<n2m_ceil_div(quads, 256), 256, 0, (hipStream_t)stream>>>(reinterpret_cast<const uint32_t*>(src), to_k(dst), words);
    else peer_copy_words_kernel<<<n2m_ceil_div(words, 256), 256, 0, (hipStream_t)stream>>>(reinterpret_cast<const uint32_t*>(src), to_k(dst), words);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_peer_reduce_slices(const float* stage1, const void* stage2, uint32_t world, uint32_t rows, float* g1, void* g2, float* found_inf,
                                      void* stream) {
    N2M_REQUIRE(world >= 1 && world <= N2M_PEER_MAX, N2M_EINVAL, "peer_reduce_slices: 1..%d ranks", N2M_PEER_MAX);
    N2M_REQUIRE((stage1 == nullptr) == (g1 == nullptr) && (stage2 == nullptr) == (g2 == nullptr) && (g1 || g2), N2M_ENULL,
                "peer_reduce_slices: a staging buffer and its output come together");
    if (rows == 0) return 0;
    peer_reduce_kernel<<<n2m_ceil_div(rows, 256), 256, 0, (hipStream_t)stream>>>(stage1, reinterpret_cast<const _Float16*>(stage2), world, rows, g1,
                                                                                       reinterpret_cast<_Float16*>(g2), found_inf);
    N2M_CHECK_LAUNCH();
    return 0;
}
