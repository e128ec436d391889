// Microbenchmark: float atomic-add throughput on MI355X by scope and XCD locality (design input for
// grid_encode_backward).  Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_bench.hip -o tools/atomic_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
    return v;
}
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}

// MODE 0: agent scope (unsafeAtomicAdd), 1: workgroup scope, 2: asm with sc1, 3: wavefront scope
// lanes of one wave instruction grouped G-wise onto consecutive dwords of a random 64*k-byte aligned run
template <int G>
__global__ void k_grouped(float* table, uint32_t rows, uint32_t per_thread) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t grp = tid / G, sub = tid % G;
    for (uint32_t i = 0; i < per_thread; ++i) {
        const uint32_t r = (hash32(grp * per_thread + i) % (rows / G)) * G + sub;
        unsafeAtomicAdd(table + r, 1.0f);
    }
}
template <int G>
float run_grouped(float* table, uint32_t rows, uint32_t blocks, uint32_t per_thread) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_grouped<G><<<blocks, 256>>>(table, rows, 4);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k_grouped<G><<<blocks, 256>>>(table, rows, per_thread);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

template <int MODE>
__global__ void k_atomics(float* table, uint32_t rows, uint32_t per_thread, int only_xcc, int partition) {
    const uint32_t xcc = xcc_id();
    if (only_xcc >= 0 && (int)xcc != only_xcc) return;
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    // partition: each XCD only touches its own 1/8 slice of the table (rows/8), else the whole table
    const uint32_t span = partition ? rows / 8 : rows;
    const uint32_t base = partition ? xcc * span : 0;
    for (uint32_t i = 0; i < per_thread; ++i) {
        const uint32_t r = base + hash32(tid * per_thread + i) % span;
        float* p = table + r;
        if (MODE == 0) unsafeAtomicAdd(p, 1.0f);
        else if (MODE == 1) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 2) asm volatile("global_atomic_add_f32 %0, %1, off sc1" ::"v"(p), "v"(1.0f) : "memory");
        else __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
}

__global__ void k_census(uint32_t* counts) { if (threadIdx.x == 0) atomicAdd(&counts[xcc_id()], 1u); }

template <int MODE>
float run(float* table, uint32_t rows, uint32_t blocks, uint32_t per_thread, int only_xcc, int partition) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipMemset(table, 0, rows * 4);
    k_atomics<MODE><<<blocks, 256>>>(table, rows, 4, only_xcc, partition);   // warm
    hipDeviceSynchronize();
    hipMemset(table, 0, rows * 4);
    hipEventRecord(a);
    k_atomics<MODE><<<blocks, 256>>>(table, rows, per_thread, only_xcc, partition);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    const uint32_t blocks = 2048, per_thread = 64;       // 2048*256*64 = 33.5 M atomics
    uint32_t* counts; hipMalloc(&counts, 64); hipMemset(counts, 0, 64);
    k_census<<<blocks, 256>>>(counts);
    uint32_t h[16]; hipMemcpy(h, counts, 64, hipMemcpyDeviceToHost);
    printf("blocks per XCC:"); for (int i = 0; i < 8; ++i) printf(" %u", h[i]); printf("\n");
    for (uint32_t rows : {4920u, 524288u, 8u * 524288u}) {
        float* table; hipMalloc(&table, (size_t)rows * 4);
        const double n = (double)blocks * 256 * per_thread;
        printf("rows=%u (%.1f KB)\n", rows, rows * 4 / 1024.0);
        float ms;
        ms = run<0>(table, rows, blocks, per_thread, -1, 0); printf("  agent scope, all XCDs, whole table : %8.3f ms  %7.2f G atomics/s\n", ms, n / ms / 1e6);
        ms = run<2>(table, rows, blocks, per_thread, -1, 0); printf("  asm sc1,     all XCDs, whole table : %8.3f ms  %7.2f G atomics/s\n", ms, n / ms / 1e6);
        ms = run<1>(table, rows, blocks, per_thread, -1, 0); printf("  wg scope,    all XCDs, whole table : %8.3f ms  %7.2f G atomics/s (results wrong across XCDs)\n", ms, n / ms / 1e6);
        ms = run<1>(table, rows, blocks, per_thread, -1, 1); printf("  wg scope,    all XCDs, XCD slices  : %8.3f ms  %7.2f G atomics/s\n", ms, n / ms / 1e6);
        // verify the sliced version sums correctly (every atomic landed once)
        std::vector<float> hv(rows); hipMemcpy(hv.data(), table, (size_t)rows * 4, hipMemcpyDeviceToHost);
        double s = 0; for (float v : hv) s += v; printf("    slice-sum check: %.0f of %.0f\n", s, n);
        ms = run<0>(table, rows, blocks, per_thread, -1, 1); printf("  agent scope, all XCDs, XCD slices  : %8.3f ms  %7.2f G atomics/s\n", ms, n / ms / 1e6);
        ms = run<3>(table, rows, blocks, per_thread, -1, 1); printf("  wave scope,  all XCDs, XCD slices  : %8.3f ms  %7.2f G atomics/s\n", ms, n / ms / 1e6);
        ms = run<0>(table, rows, blocks, per_thread, 0, 0); printf("  agent scope, XCD0 only (1/8 work)  : %8.3f ms  %7.2f G atomics/s\n", ms, n / 8 / ms / 1e6);
        ms = run<1>(table, rows, blocks, per_thread, 0, 0); printf("  wg scope,    XCD0 only (1/8 work)  : %8.3f ms  %7.2f G atomics/s\n", ms, n / 8 / ms / 1e6);
        ms = run_grouped<2>(table, rows, blocks, per_thread);  printf("  agent, lanes paired on adjacent dwords     : %8.3f ms  %7.2f G atomics/s\n", ms, n / ms / 1e6);
        ms = run_grouped<4>(table, rows, blocks, per_thread);  printf("  agent, 4 lanes per 16 B                    : %8.3f ms  %7.2f G atomics/s\n", ms, n / ms / 1e6);
        ms = run_grouped<16>(table, rows, blocks, per_thread); printf("  agent, 16 lanes per 64 B line              : %8.3f ms  %7.2f G atomics/s\n", ms, n / ms / 1e6);
        ms = run_grouped<32>(table, rows, blocks, per_thread); printf("  agent, 32 lanes per 128 B line             : %8.3f ms  %7.2f G atomics/s\n", ms, n / ms / 1e6);
        ms = run_grouped<64>(table, rows, blocks, per_thread); printf("  agent, 64 lanes contiguous (256 B)         : %8.3f ms  %7.2f G atomics/s\n", ms, n / ms / 1e6);
        hipFree(table);
    }
    return 0;
}
