// Microbenchmark: LDS atomic-add throughput per CU on MI355X (design input for grid_encode_backward).
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lds_atomic_bench.hip -o tools/lds_atomic_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// MODE 0: ds_add_f32   1: ds_add_u32   2: plain ds_write_b32 (no atomic)   3: ds_add_f32 all lanes same row per 8-lane group
//      4: ds_add_rtn_f32 (returning)   5: ds_pk_add_f16   6: ds_add_u64
template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, uint32_t per_thread, uint32_t rows) {
    extern __shared__ float lds[];
    for (uint32_t i = threadIdx.x; i < rows; i += 1024) lds[i] = 0.f;
    __syncthreads();
    const uint32_t tid = blockIdx.x * 1024 + threadIdx.x;
    float acc = 0.f;
    for (uint32_t i = 0; i < per_thread; ++i) {
        uint32_t r = hash32(tid * per_thread + i) % rows;
        if (MODE == 3) r = hash32((tid >> 3) * per_thread + i) % rows;
        if (MODE == 0 || MODE == 3) __hip_atomic_fetch_add(&lds[r], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 1) __hip_atomic_fetch_add((uint32_t*)&lds[r], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 2) lds[r] = (float)i;
        else if (MODE == 4) acc += __hip_atomic_fetch_add(&lds[r], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 5) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            h2 v = {(_Float16)1, (_Float16)1};
            asm volatile("ds_pk_add_f16 %0, %1" ::"v"((uint32_t)(uintptr_t)(r * 4)), "v"(v) : "memory");
        } else if (MODE == 6) {
            asm volatile("ds_add_u64 %0, %1" ::"v"((uint32_t)(uintptr_t)((r & ~1u) * 4)), "v"(1ull) : "memory");
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[0] + acc;
}

template <int MODE>
void run(const char* name, float* out, uint32_t rows) {
    const uint32_t blocks = 256, per_thread = 512;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 1024, 131072>>>(out, 8, rows);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<blocks, 1024, 131072>>>(out, per_thread, rows);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double n = (double)blocks * 1024 * per_thread;
    printf("  %-46s rows=%6u : %8.3f ms  %7.2f G ops/s chip  %6.3f G/s per CU  %5.2f lanes/clk/CU @2.1GHz\n", name, rows, ms, n / ms / 1e6, n / ms / 1e6 / 256, n / ms / 1e6 / 256 / 2.1);
}

int main() {
    float* out; hipMalloc(&out, 4096);
    for (uint32_t rows : {4920u, 32768u}) {
        run<0>("ds_add_f32 random", out, rows);
        run<4>("ds_add_rtn_f32 random", out, rows);
        run<1>("ds_add_u32 random", out, rows);
        run<6>("ds_add_u64 random", out, rows);
        run<5>("ds_pk_add_f16 random", out, rows);
        run<2>("ds_write_b32 random (no atomic)", out, rows);
        run<3>("ds_add_f32, 8 lanes per address", out, rows);
    }
    return 0;
}
