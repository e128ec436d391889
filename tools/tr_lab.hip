// Semantics check of ds_read_b64_tr_b16 on a [32 samples][pitch] fp16 tile (the layout of mlp.hip's transpose scratch): every lane passes
// the address of "its" 4 contiguous halves of a [4][16] block, the 16-lane group gets the block back column-major.
//   hipcc --offload-arch=gfx950 -O3 tools/tr_lab.hip -o tools/tr_lab && tools/tr_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef short s4v __attribute__((__vector_size__(8)));
constexpr int PITCH = 68;
__global__ void k(h4* out, const _Float16* in, int kq, int fb) {
    __shared__ __attribute__((aligned(16))) _Float16 t[32 * PITCH];
    for (int i = threadIdx.x; i < 32 * PITCH; i += 64) t[i] = in[i];
    __syncthreads();
    const int lane = threadIdx.x;
    const _Float16* p = t + (8 * kq + 4 * (lane >> 5) + ((lane & 15) >> 2)) * PITCH + 32 * fb + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const s4v r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)p);
    out[lane] = __builtin_bit_cast(h4, r);
}
int main() {
    std::vector<_Float16> h(32 * PITCH);
    for (int r = 0; r < 32; ++r) for (int c = 0; c < PITCH; ++c) h[r * PITCH + c] = (_Float16)(float)(r * 64 + (c % 64));
    _Float16* d; h4* o;
    hipMalloc(&d, h.size() * 2); hipMalloc(&o, 64 * 8);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    int bad = 0;
    for (int kq = 0; kq < 4; ++kq) for (int fb = 0; fb < 2; ++fb) {
        k<<<1, 64>>>(o, d, kq, fb);
        h4 res[64];
        hipMemcpy(res, o, sizeof(res), hipMemcpyDeviceToHost);
        for (int lane = 0; lane < 64; ++lane) for (int i = 0; i < 4; ++i) {
            const float want = (float)((8 * kq + 4 * (lane >> 5) + i) * 64 + 32 * fb + (lane & 31));     // tile_get's element
            if ((float)res[lane][i] != want) { if (bad < 8) printf("kq %d fb %d lane %d i %d: got %g want %g\n", kq, fb, lane, i, (float)res[lane][i], want); ++bad; }
        }
    }
    printf("tr_lab: %d mismatches\n", bad);
    return bad != 0;
}
