// Library runtime: error reporting and the hipEvent-based per-kernel timing pool behind n2m_prof_*.
#include <stdarg.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "n2m_common.hpp"

static thread_local char g_err[512] = "";

void n2m_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int n2m_abi_version(void) { return N2M_ABI_VERSION; }
extern "C" const char* n2m_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------ stream copy (measurement support)
// The practical HBM streaming ceiling of this device, as bench.py's second denominator: a grid-stride copy of 16 bytes per lane (the form
// MI355X_MICROARCH.md measures 6.29 TB/s with), read + write counted by the caller.  Nontemporal on both sides: nothing is re-read.
typedef float n2m_f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) stream_copy_f4_kernel(const n2m_f4* __restrict__ src, n2m_f4* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const n2m_f4 v = __builtin_nontemporal_load(src + i);
        __builtin_nontemporal_store(v, dst + i);
    }
}

extern "C" int n2m_stream_copy(const void* src, void* dst, uint64_t bytes, uint32_t workgroups, void* stream) {
    N2M_REQUIRE(src && dst && bytes % 16u == 0 && ((uintptr_t)src | (uintptr_t)dst) % 16u == 0, N2M_EINVAL,
                "n2m_stream_copy: 16-byte aligned buffers and a multiple of 16 bytes");
    if (bytes == 0) return 0;
    const size_t n = (size_t)(bytes / 16u);
    size_t wg = workgroups ? workgroups : 256u * 8u;                   // 8 workgroups of 256 threads per CU
    if (wg > (n + 255u) / 256u) wg = (n + 255u) / 256u;
    hipLaunchKernelGGL(stream_copy_f4_kernel, dim3((unsigned)wg), dim3(256), 0, (hipStream_t)stream, (const n2m_f4*)src, (n2m_f4*)dst, n);
    N2M_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------ profiling
namespace {
struct Slot {
    hipEvent_t a, b;
    int kernel;
    double bytes;
};
constexpr int kPool = 16384;
std::mutex g_mu;
std::vector<Slot> g_slots;     // events are created lazily and reused after n2m_prof_reset
int g_used = 0;
int g_every = 0;                // 0 = off, 1 = every launch, n = every n-th launch of each kernel id
uint64_t g_seen[N2M_K_COUNT];
uint64_t g_untimed[N2M_K_COUNT];
const char* kNames[N2M_K_COUNT] = {"grid_encode_forward", "grid_encode_backward", "grad_total_variation",
                                   "march_rays_train_count", "march_rays_train_write", "composite_rays_train_forward",
                                   "composite_rays_train_backward", "near_far_from_aabb", "packbits", "mlp_forward",
                                   "mlp_backward", "rasterize", "grid_encode_forward_packed", "adam_step",
                                   "interpolate_forward", "interpolate_backward", "antialias_forward", "antialias_backward", "rasterize_backward"};
}  // namespace

N2mProfLaunchState& n2m_prof_launch_state() {
    static thread_local N2mProfLaunchState st{nullptr, nullptr, 0};
    return st;
}

void n2m_prof_fall_back_to_markers(hipStream_t s) {
    N2mProfLaunchState& st = n2m_prof_launch_state();
    if (st.armed == 1) { (void)hipEventRecord(st.a, s); st.armed = 3; }
}

N2mProfScope::N2mProfScope(int kernel_id, hipStream_t s, double algo_bytes, bool kernel_events_) : slot(-1), stream(s), kernel_events(kernel_events_) {
    if (g_every == 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    // sampled timing: every g_every-th launch of a kernel id, the ids staggered so that one step does not carry every entry's events
    if (((g_seen[kernel_id]++ + (uint64_t)kernel_id * 3u) % (uint64_t)g_every) != 0) return;
    if (g_used >= kPool) {
        g_untimed[kernel_id]++;
        return;
    }
    if (g_used >= (int)g_slots.size()) {
        Slot sl;
        if (hipEventCreate(&sl.a) != hipSuccess || hipEventCreate(&sl.b) != hipSuccess) return;
        g_slots.push_back(sl);
    }
    slot = g_used++;
    g_slots[slot].kernel = kernel_id;
    g_slots[slot].bytes = algo_bytes;
    if (kernel_events) {
        N2mProfLaunchState& st = n2m_prof_launch_state();
        st.a = g_slots[slot].a; st.b = g_slots[slot].b; st.armed = 1;
    } else (void)hipEventRecord(g_slots[slot].a, stream);
}

N2mProfScope::~N2mProfScope() {
    if (slot < 0) return;
    if (kernel_events) {
        N2mProfLaunchState& st = n2m_prof_launch_state();
        if (st.armed == 1) {                                         // the entry launched nothing (empty batch): a defined, empty interval
            (void)hipEventRecord(g_slots[slot].a, stream);
            (void)hipEventRecord(g_slots[slot].b, stream);
        } else if (st.armed == 3) (void)hipEventRecord(g_slots[slot].b, stream);
        st.armed = 0;
    } else (void)hipEventRecord(g_slots[slot].b, stream);
}

extern "C" int n2m_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_every = on < 0 ? 0 : on;
    return 0;
}

extern "C" int n2m_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_used = 0;
    memset(g_untimed, 0, sizeof(g_untimed));
    memset(g_seen, 0, sizeof(g_seen));
    return 0;
}

extern "C" int n2m_prof_read(int kernel_id, uint64_t* launches, double* total_ms, double* algo_bytes) {
    N2M_REQUIRE(kernel_id >= 0 && kernel_id < N2M_K_COUNT, N2M_EINVAL, "n2m_prof_read: bad kernel id %d", kernel_id);
    std::lock_guard<std::mutex> lk(g_mu);
    uint64_t n = 0;
    double ms = 0, bytes = 0;
    for (int i = 0; i < g_used; ++i) {
        if (g_slots[i].kernel != kernel_id) continue;
        N2M_HIP(hipEventSynchronize(g_slots[i].b));
        float t = 0;
        N2M_HIP(hipEventElapsedTime(&t, g_slots[i].a, g_slots[i].b));
        ms += t;
        bytes += g_slots[i].bytes;
        ++n;
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
    if (algo_bytes) *algo_bytes = bytes;
    return 0;
}

extern "C" int n2m_prof_seen(int kernel_id, uint64_t* launches_seen) {
    N2M_REQUIRE(kernel_id >= 0 && kernel_id < N2M_K_COUNT, N2M_EINVAL, "n2m_prof_seen: bad kernel id %d", kernel_id);
    std::lock_guard<std::mutex> lk(g_mu);
    if (launches_seen) *launches_seen = g_seen[kernel_id];
    return 0;
}

extern "C" const char* n2m_prof_name(int kernel_id) {
    return (kernel_id >= 0 && kernel_id < N2M_K_COUNT) ? kNames[kernel_id] : "?";
}
