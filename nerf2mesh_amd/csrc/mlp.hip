// Fused field MLPs (sigma_net, color_net, specular_net of nerf/network.py:66-75) for gfx950 -- include/n2m_mlp.h.
//
// Why fuse: per sample the three networks are 7 648 MACs with K <= 64.  As nn.Linear calls they become 7 GEMM launches
// forward and ~14 backward; the weight-gradient GEMMs are (out x 262144) x (262144 x in) with out, in <= 64, which the
// BLAS library runs as ONE workgroup each (1.8 ms + 2 x 0.5 ms per step measured, profiles/r01_*).  Here each wavefront
// carries a tile of 32 samples through the whole head with the weights in LDS and everything else in registers.
//
// Orientation.  v_mfma_f32_32x32x8_f16 computes D[32 x 32] += A[32 x 8] * B[8 x 32] with the lane maps
//     A: lane l -> A[m = l%32][k = 4*(l/32) + i]      B: lane l -> B[k = 4*(l/32) + i][n = l%32]      (i = 0..3)
//     D: lane l, reg r -> D[m = 8*(r/4) + 4*(l/32) + r%4][n = l%32]
// Putting the SAMPLES on n and the FEATURES on m/k (H^T = W * X^T) makes the D layout of one layer exactly the B
// layout of the next: regs 4kb..4kb+3 of lane l hold features 8kb + 4*(l/32) + 0..3 of sample l%32.  So a layer is
// "ReLU, round to fp16, pack 4 values" -- no LDS round trip, no shuffles -- and the chain stays in VGPRs.  The same
// holds for the backward activation-gradient chain (dX^T = W^T * dY^T).
// Weight gradients contract over SAMPLES (dW = dY^T X), which live on lanes; for those the 32-sample tiles of X and dY
// are transposed through a small per-wave LDS scratch and accumulated with MFMA into per-wave fp32 register tiles
// for the whole launch (persistent waves), then reduced once per workgroup in LDS and flushed with coalesced atomics.
//
// Numerics = autocast: operands fp16, fp32 accumulate, layer outputs rounded to fp16; exp (trunc_exp) in fp32.
// Input feature order inside the kernel is [encoder features | xyz | 0-pad] (the weight columns are permuted while
// staging), so the encoder rows load as aligned 8-byte chunks.
#include <map>
#include <mutex>

#include "n2m_common.hpp"
#include "../../include/n2m_mlp.h"

namespace {

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16x __attribute__((ext_vector_type(16)));

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x8f16((a), (b), (c), 0, 0, 0)
// gfx950's double-K form: one instruction for two K blocks at the issue cost of one (8 passes either way).  Lane (x, g) supplies
// k = 8g + i, i = 0..7, to A and B alike; with i < 4 taken from K block 2j and i >= 4 from block 2j + 1 -- i.e. the two 32x32x8
// fragments side by side -- every (g, i) names the same feature on both operands, so the sum runs over the same 16 features and no
// image has to be re-staged.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

// ---------------------------------------------------------------------------------------------- LDS layout (halves)
// forward operands W[out_pad][k_pad + 4]; backward operands W^T[in_pad][out_pad + 4]
constexpr int P_S0 = 28, P_S1 = 36, P_C0 = 44, P_C1 = 68, P_C2 = 68, P_P0 = 12, P_P1 = 36;        // forward pitches
constexpr int P_S1T = 12, P_S0T = 36, P_C2T = 12, P_C1T = 68, P_C0T = 68, P_P1T = 12, P_P0T = 36; // backward pitches
constexpr int O_S0 = 0;
constexpr int O_S1 = O_S0 + 32 * P_S0;
constexpr int O_C0 = O_S1 + 32 * P_S1;
constexpr int O_C1 = O_C0 + 64 * P_C0;
constexpr int O_C2 = O_C1 + 64 * P_C1;
constexpr int O_P0 = O_C2 + 32 * P_C2;
constexpr int O_P1 = O_P0 + 32 * P_P0;
constexpr int FWD_HALVES = O_P1 + 32 * P_P1;
constexpr int O_S1T = FWD_HALVES;
constexpr int O_S0T = O_S1T + 32 * P_S1T;
constexpr int O_C2T = O_S0T + 32 * P_S0T;
constexpr int O_C1T = O_C2T + 64 * P_C2T;
constexpr int O_C0T = O_C1T + 64 * P_C1T;
constexpr int O_P1T = O_C0T + 64 * P_C0T;
constexpr int O_P0T = O_P1T + 32 * P_P1T;
constexpr int ALL_W_HALVES = O_P0T + 32 * P_P0T;
constexpr int TILE_PITCH = 68;                                   // per-wave transpose tiles [32 samples][64 + 4]
constexpr int TILE_HALVES = 32 * TILE_PITCH;
constexpr int BWD_HALVES = ALL_W_HALVES + 4 * 2 * TILE_HALVES;   // 4 waves x (X tile + dY tile)
constexpr int BWD_PC_TILES_HALVES = ALL_W_HALVES + 4 * 4 * TILE_HALVES; // producer/consumer form: 4 pairs x 2 buffers x (X tile + dY tile)
constexpr int BWD_PC_HALVES = BWD_PC_TILES_HALVES > 65536 ? BWD_PC_TILES_HALVES : 65536;    // >= 128 KB: the epilogue's 4 x 2 x (64 x 64) fp32 copies

// logical input feature k -> column of the nn.Linear weight (or -1 for padding)
__device__ __forceinline__ int col_color0(int k) { return k < 32 ? 3 + k : (k < 35 ? k - 32 : -1); }
__device__ __forceinline__ int col_sigma0(int k) { return k < 16 ? 3 + k : (k < 19 ? k - 16 : -1); }
__device__ __forceinline__ int col_plain(int k, int in) { return k < in ? k : -1; }

enum Perm { PERM_PLAIN = 0, PERM_COLOR0 = 1, PERM_SIGMA0 = 2 };
__device__ __forceinline__ int col_of(int perm, int k, int in) {
    return perm == PERM_COLOR0 ? col_color0(k) : perm == PERM_SIGMA0 ? col_sigma0(k) : col_plain(k, in);
}

// W[out,in] fp32 (global) -> LDS fp16 [m_pad][pitch], logical k order, zero padded
__device__ void stage_w(_Float16* dst, int pitch, const float* __restrict__ W, int out, int in, int m_pad, int k_pad, int perm) {
    for (int idx = threadIdx.x; idx < m_pad * k_pad; idx += blockDim.x) {
        const int m = idx / k_pad, k = idx - m * k_pad;
        const int c = col_of(perm, k, in);
        dst[m * pitch + k] = (_Float16)((m < out && c >= 0) ? W[m * in + c] : 0.0f);
    }
}
// transposed: LDS [m = logical in feature][k = out feature]
__device__ void stage_wt(_Float16* dst, int pitch, const float* __restrict__ W, int out, int in, int m_pad, int k_pad, int perm) {
    for (int idx = threadIdx.x; idx < m_pad * k_pad; idx += blockDim.x) {
        const int m = idx / k_pad, k = idx - m * k_pad;
        const int c = col_of(perm, m, in);
        dst[m * pitch + k] = (_Float16)((k < out && c >= 0) ? W[k * in + c] : 0.0f);
    }
}

struct FieldArgs {
    const float* xyz; const float* dirs; const float* h1; const _Float16* h2;
    const float* w[7];       // sigma0, sigma1, color0, color1, color2, spec0, spec1
    uint32_t M; int shading; int normalize_dirs;
    int raw_density;         // SDF head: sigma = the fp16 Linear output as it is (nerf/network.py:100-101), no trunc_exp
    float* sigma; float* rgb; float* specular;
    // backward only
    const float* d_sigma; const float* d_rgb; const float* d_specular;
    float* d_h1; _Float16* d_h2; float* dw[7];
    float* found_inf;        // set to 1 when a weight gradient is not finite (GradScaler's check), may be NULL
    int dbg;                 // measurement switches (N2M_FIELD_DEBUG): 1 skip the tile loop, 2 skip the dW reduction
    float* dw_partial;       // [gridDim.x][kDwTotal]: per-workgroup weight-gradient sums, reduced by dw_finalize_kernel
    // specular regulariser lambda * mean_m sum_c specular^2 (nerf/utils.py:733-737), folded into the field kernels (*_train entry points):
    float* spec_sq_partial;  // forward: [kSpecPartials] per-workgroup sums of specular^2 (unused slots zeroed), NULL = off
    float spec_reg;          // backward: d loss / d specular += specular * (*seed * spec_reg), spec_reg = 2 lambda / M; 0 = off
    const float* seed;       // device scalar: the seed gradient (loss scale [/ world]); read only when spec_reg != 0
};
constexpr int kSpecPartials = 512;      // >= the largest forward grid (2 x 256 workgroups)

// Staging, fast form.  stage_w / stage_wt above walk the PADDED image and fetch one weight per iteration (integer division, a
// dependent global load, a 2-byte LDS store: ~46 serial round trips per thread, 8-20 us of every launch).  Here every thread first
// requests its share of the 7 648 real weights (coalesced, all loads in flight at once), the image area is cleared with 16-byte
// stores, and each weight is then written to its place in the forward image and -- backward kernels -- the transposed image.
template <int PERM>
__device__ __forceinline__ int logical_k(int c) {      // inverse of col_of: weight column -> logical input feature
    return PERM == PERM_COLOR0 ? (c < 3 ? 32 + c : c - 3) : PERM == PERM_SIGMA0 ? (c < 3 ? 16 + c : c - 3) : c;
}
template <int NT, int OUT, int IN>
struct WRegs { float v[(OUT * IN + NT - 1) / NT]; };
template <int NT, int OUT, int IN>
__device__ __forceinline__ void w_fetch(WRegs<NT, OUT, IN>& r, const float* __restrict__ W) {
#pragma unroll
    for (int i = 0; i < (OUT * IN + NT - 1) / NT; ++i) {
        const int e = (int)threadIdx.x + i * NT;
        r.v[i] = e < OUT * IN ? W[e] : 0.f;
    }
}
template <int NT, int OUT, int IN, int PERM, bool TR>
__device__ __forceinline__ void w_place(const WRegs<NT, OUT, IN>& r, _Float16* fwd, int pitch, _Float16* tr, int pitch_t) {
#pragma unroll
    for (int i = 0; i < (OUT * IN + NT - 1) / NT; ++i) {
        const int e = (int)threadIdx.x + i * NT;
        if (e < OUT * IN) {
            const int m = e / IN, k = logical_k<PERM>(e - m * IN);
            const _Float16 h = (_Float16)r.v[i];
            fwd[m * pitch + k] = h;
            if (TR) tr[k * pitch_t + m] = h;
        }
    }
}
template <int NT, bool TR, bool DO_DENSITY, bool DO_COLOR>
__device__ __forceinline__ void stage_images(_Float16* lds, const FieldArgs& a) {
    WRegs<NT, 32, 19> s0; WRegs<NT, 1, 32> s1; WRegs<NT, 64, 35> c0; WRegs<NT, 64, 64> c1; WRegs<NT, 6, 64> c2; WRegs<NT, 32, 6> p0; WRegs<NT, 3, 32> p1;
    const bool spec = DO_COLOR && a.shading != 0;
    if (DO_DENSITY) { w_fetch(s0, a.w[0]); w_fetch(s1, a.w[1]); }
    if (DO_COLOR) { w_fetch(c0, a.w[2]); w_fetch(c1, a.w[3]); w_fetch(c2, a.w[4]); }
    if (spec) { w_fetch(p0, a.w[5]); w_fetch(p1, a.w[6]); }
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (int i = threadIdx.x; i < (TR ? ALL_W_HALVES : FWD_HALVES) / 8; i += NT) reinterpret_cast<uint4*>(lds)[i] = z;
    __syncthreads();
    if (DO_DENSITY) {
        w_place<NT, 32, 19, PERM_SIGMA0, TR>(s0, lds + O_S0, P_S0, lds + O_S0T, P_S0T);
        w_place<NT, 1, 32, PERM_PLAIN, TR>(s1, lds + O_S1, P_S1, lds + O_S1T, P_S1T);
    }
    if (DO_COLOR) {
        w_place<NT, 64, 35, PERM_COLOR0, TR>(c0, lds + O_C0, P_C0, lds + O_C0T, P_C0T);
        w_place<NT, 64, 64, PERM_PLAIN, TR>(c1, lds + O_C1, P_C1, lds + O_C1T, P_C1T);
        w_place<NT, 6, 64, PERM_PLAIN, TR>(c2, lds + O_C2, P_C2, lds + O_C2T, P_C2T);
    }
    if (spec) {
        w_place<NT, 32, 6, PERM_PLAIN, TR>(p0, lds + O_P0, P_P0, lds + O_P0T, P_P0T);
        w_place<NT, 3, 32, PERM_PLAIN, TR>(p1, lds + O_P1, P_P1, lds + O_P1T, P_P1T);
    }
    __syncthreads();
}

__device__ __forceinline__ h4 ld_a(const _Float16* W, int pitch, int mb, int kb, int lane) {
    return *reinterpret_cast<const h4*>(W + (32 * mb + (lane & 31)) * pitch + 8 * kb + 4 * (lane >> 5));
}

// All A fragments (weights) of a layer into registers, and the layer's MFMAs with the M blocks interleaved.  Written apart because
// the compiler otherwise re-reads ONE register pair per two MFMAs (ds_read -> s_waitcnt lgkmcnt(0) -> 2 MFMAs -> ds_read into the same
// pair ...: the full LDS latency exposed forty times per tile) and runs each accumulator's K loop as one dependent chain (64-cycle
// result latency against a 32-cycle issue).  The loads of layer L+1 are issued before the MFMAs of layer L.
template <int MB, int KB>
__device__ __forceinline__ void ld_layer(h4 (&w)[MB][KB], const _Float16* W, int pitch, int lane) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) w[mb][kb] = ld_a(W, pitch, mb, kb, lane);
}
__device__ __forceinline__ h8 cat8(const h4& lo, const h4& hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }
template <int MB, int KB>
__device__ __forceinline__ void mm_layer(f16x (&d)[MB], const h4 (&w)[MB][KB], const h4 (&b)[KB]) {
#pragma unroll
    for (int kb = 0; kb + 1 < KB; kb += 2)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) d[mb] = MFMA16(cat8(w[mb][kb], w[mb][kb + 1]), cat8(b[kb], b[kb + 1]), d[mb]);
    if (KB & 1) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) d[mb] = MFMA(w[mb][KB - 1], b[KB - 1], d[mb]);
    }
}
// the K loop of one M block straight from the LDS image (small layers)
template <int KB>
__device__ __forceinline__ f16x mm_k(const _Float16* W, int pitch, int mb, const h4 (&b)[KB], f16x acc, int lane) {
#pragma unroll
    for (int kb = 0; kb + 1 < KB; kb += 2)
        acc = MFMA16(cat8(ld_a(W, pitch, mb, kb, lane), ld_a(W, pitch, mb, kb + 1, lane)), cat8(b[kb], b[kb + 1]), acc);
    if (KB & 1) acc = MFMA(ld_a(W, pitch, mb, KB - 1, lane), b[KB - 1], acc);
    return acc;
}

__device__ __forceinline__ h4 zero4() { h4 z = {(_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0}; return z; }
__device__ __forceinline__ f16x zero16() {
    f16x z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// regs 4q..4q+3 of an accumulator -> fp16 fragment, with ReLU (the next layer's B operand for K-block q).
// Packed forms: two conversions are one v_cvt_pk_f16_f32; ReLU on the half BIT PATTERNS is a signed 16-bit max with 0 (negative
// halves, -0 included, are negative integers; NaN passes through like torch's relu); the backward's "activation was positive" test
// is the sign smear of the negated pattern.  7 VALU instructions per pair of values became 2 (forward) / 4 (backward mask).
typedef _Float16 h2v_t __attribute__((ext_vector_type(2)));
typedef short s2v_t __attribute__((ext_vector_type(2)));
typedef uint32_t u2v_t __attribute__((ext_vector_type(2)));
typedef float f2v_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk(float a, float b) {
    const f2v_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h2v_t));       // one fptrunc <2 x float>: v_cvt_pk_f16_f32
}
__device__ __forceinline__ uint32_t relu_pk(uint32_t hh) {
    const s2v_t z = {0, 0};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s2v_t, hh), z));
}
__device__ __forceinline__ uint32_t positive_mask(uint32_t relu_out) {     // 0xFFFF per half that is > 0 (input: a ReLU output, never negative)
    // sign smear of the negated pattern, two packed instructions (written in C++ the compiler turns it back into two compares, two
    // selects and a permute); op_sel_hi:[0,1] feeds the shift count 15 to both halves
    uint32_t m;
    asm("v_pk_sub_i16 %0, 0, %1\n\tv_pk_ashrrev_i16 %0, 15, %0 op_sel_hi:[0,1]" : "=v"(m) : "v"(relu_out));
    return m;
}
template <int Q>
__device__ __forceinline__ h4 relu_pack(const f16x& d) {
    u2v_t r;
    r.x = relu_pk(cvt_pk(d[4 * Q], d[4 * Q + 1]));
    r.y = relu_pk(cvt_pk(d[4 * Q + 2], d[4 * Q + 3]));
    return __builtin_bit_cast(h4, r);
}
// gradient fragment: round to fp16 and apply the ReLU mask of the forward fragment
template <int Q>
__device__ __forceinline__ h4 mask_pack(const f16x& d, const h4& fwd) {
    const u2v_t f = __builtin_bit_cast(u2v_t, fwd);
    u2v_t r;
    r.x = cvt_pk(d[4 * Q], d[4 * Q + 1]) & positive_mask(f.x);
    r.y = cvt_pk(d[4 * Q + 2], d[4 * Q + 3]) & positive_mask(f.y);
    return __builtin_bit_cast(h4, r);
}

__device__ __forceinline__ float sigmoid_h(float pre_acc) {
    // autocast: the Linear output is fp16, sigmoid evaluates it and returns fp16
    const float x = (float)(_Float16)pre_acc;
    return (float)(_Float16)(1.0f / (1.0f + expf(-x)));
}


// View direction of sample s as fp16 MLP operands.  normalize_dirs: the march kernel hands out the raw ray direction
// (raymarching.cu:466-468) and the reference normalises it per sample in torch (safe_normalize, nerf/renderer.py:704,
// nerf/utils.py:43-44: x / sqrt(clamp(sum x^2, 1e-20))); folding that here saves five [M,3] elementwise passes.
template <typename P>
__device__ __forceinline__ void load_dir(const FieldArgs& a, uint32_t s, P& bp) {
    float d0 = a.dirs[(size_t)s * 3], d1 = a.dirs[(size_t)s * 3 + 1], d2 = a.dirs[(size_t)s * 3 + 2];
    if (a.normalize_dirs) {
        const float n = sqrtf(fmaxf((d0 * d0 + d1 * d1) + d2 * d2, 1e-20f));
        d0 /= n; d1 /= n; d2 /= n;
    }
    bp[0] = (_Float16)d0; bp[1] = (_Float16)d1; bp[2] = (_Float16)d2;
}

// ---------------------------------------------------------------------------------------------- input fragments
// Encoder features arrive LEVEL-major (the layout grid_encode_forward writes fastest): h1 [16][M] fp32, h2 [16][M][2] fp16.
// For a fixed feature the 32 samples of a tile are contiguous, so every access below is a coalesced 128-byte row.
__device__ __forceinline__ void load_density_inputs(const FieldArgs& a, uint32_t s, bool valid, int g, h4 (&b)[3]) {
    const size_t Mz = a.M;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        b[kb] = zero4();
        if (valid) {
#pragma unroll
            for (int i = 0; i < 4; ++i) b[kb][i] = (_Float16)a.h1[(size_t)(8 * kb + 4 * g + i) * Mz + s];
        }
    }
    b[2] = zero4();
    if (valid && g == 0) {
        b[2][0] = (_Float16)a.xyz[(size_t)s * 3]; b[2][1] = (_Float16)a.xyz[(size_t)s * 3 + 1]; b[2][2] = (_Float16)a.xyz[(size_t)s * 3 + 2];
    }
}
__device__ __forceinline__ void load_color_inputs(const FieldArgs& a, uint32_t s, bool valid, int g, h4 (&b)[5]) {
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    const size_t Mz = a.M;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        b[kb] = zero4();
        if (valid) {   // features 8kb+4g .. +3 = levels 4kb+2g and 4kb+2g+1, two channels each
            const h2v lo = *reinterpret_cast<const h2v*>(a.h2 + ((size_t)(4 * kb + 2 * g) * Mz + s) * 2);
            const h2v hi = *reinterpret_cast<const h2v*>(a.h2 + ((size_t)(4 * kb + 2 * g + 1) * Mz + s) * 2);
            b[kb][0] = lo.x; b[kb][1] = lo.y; b[kb][2] = hi.x; b[kb][3] = hi.y;
        }
    }
    b[4] = zero4();
    if (valid && g == 0) {
        b[4][0] = (_Float16)a.xyz[(size_t)s * 3]; b[4][1] = (_Float16)a.xyz[(size_t)s * 3 + 1]; b[4][2] = (_Float16)a.xyz[(size_t)s * 3 + 2];
    }
}

// Software prefetch.  A wave works through its tiles one after the other with one or two waves per SIMD, so nothing hides a global load
// issued at its point of use: the backward's producer wave had SIX exposed round trips per tile (h1, xyz, d_sigma, h2 + xyz, d_rgb +
// d_specular, dirs: each its own branch with its own s_waitcnt) -- more than half of its time.  All per-sample inputs of the NEXT tile are
// therefore requested, raw and unconverted, before the current tile's arithmetic starts, and turned into MFMA fragments one iteration
// later.  Same loads, same conversions, same values.
struct RawTile {
    float h1v[8];        // density features 8kb + 4g + i   (kb = 0, 1)
    uint32_t h2v[8];     // colour features: levels 4kb + 2g, 4kb + 2g + 1 as half2 bit patterns (kb = 0..3)
    float xyz[3], dir[3], dsig, drgb[3], dspec[3];      // g = 0 lanes only
};
template <bool DO_DENSITY, bool DO_COLOR, bool BWD>
__device__ __forceinline__ void fetch_tile(const FieldArgs& a, uint32_t tile, int n, int g, RawTile& r) {
    const uint32_t s = tile * 32 + (uint32_t)n;
    const size_t Mz = a.M;
#pragma unroll
    for (int i = 0; i < 8; ++i) { r.h1v[i] = 0.f; r.h2v[i] = 0u; }
#pragma unroll
    for (int i = 0; i < 3; ++i) { r.xyz[i] = 0.f; r.dir[i] = 0.f; r.drgb[i] = 0.f; r.dspec[i] = 0.f; }
    r.dsig = 0.f;
    if (s >= a.M) return;
    if (DO_DENSITY) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i) r.h1v[4 * kb + i] = a.h1[(size_t)(8 * kb + 4 * g + i) * Mz + s];
    }
    if (DO_COLOR) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            r.h2v[2 * kb] = *reinterpret_cast<const uint32_t*>(a.h2 + ((size_t)(4 * kb + 2 * g) * Mz + s) * 2);
            r.h2v[2 * kb + 1] = *reinterpret_cast<const uint32_t*>(a.h2 + ((size_t)(4 * kb + 2 * g + 1) * Mz + s) * 2);
        }
    }
    if (g == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) r.xyz[i] = a.xyz[(size_t)s * 3 + i];
        if (DO_COLOR && a.shading != 0) {
#pragma unroll
            for (int i = 0; i < 3; ++i) r.dir[i] = a.dirs[(size_t)s * 3 + i];
        }
        if (BWD) {
            if (DO_DENSITY) r.dsig = a.d_sigma[s];
            if (DO_COLOR) {
#pragma unroll
                for (int i = 0; i < 3; ++i) r.drgb[i] = a.d_rgb[(size_t)s * 3 + i];
                if (a.d_specular && a.shading != 0) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) r.dspec[i] = a.d_specular[(size_t)s * 3 + i];
                }
            }
        }
    }
}
// "These registers are read here": makes the compiler finish the tile's loads at THIS point (the top of the iteration that consumes them,
// a whole tile after they were issued) instead of carrying them as pending across the loop edge -- with the in-order vmcnt counter a
// pending OLD load met in mid-iteration would also wait for the prefetch issued after it.
__device__ __forceinline__ void settle_tile(RawTile& r) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { asm volatile("" : "+v"(r.h1v[i])); asm volatile("" : "+v"(r.h2v[i])); }
#pragma unroll
    for (int i = 0; i < 3; ++i) { asm volatile("" : "+v"(r.xyz[i])); asm volatile("" : "+v"(r.dir[i])); asm volatile("" : "+v"(r.drgb[i])); asm volatile("" : "+v"(r.dspec[i])); }
    asm volatile("" : "+v"(r.dsig));
}
// fragments from the raw tile: what load_density_inputs / load_color_inputs / load_dir build (lanes without a sample carry zeros)
__device__ __forceinline__ void density_frags(const RawTile& r, h4 (&b)[3]) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 4; ++i) b[kb][i] = (_Float16)r.h1v[4 * kb + i];
    b[2] = zero4();
    b[2][0] = (_Float16)r.xyz[0]; b[2][1] = (_Float16)r.xyz[1]; b[2][2] = (_Float16)r.xyz[2];
}
__device__ __forceinline__ void color_frags(const RawTile& r, h4 (&b)[5]) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const u2v_t v = {r.h2v[2 * kb], r.h2v[2 * kb + 1]};
        b[kb] = __builtin_bit_cast(h4, v);
    }
    b[4] = zero4();
    b[4][0] = (_Float16)r.xyz[0]; b[4][1] = (_Float16)r.xyz[1]; b[4][2] = (_Float16)r.xyz[2];
}
template <typename P>
__device__ __forceinline__ void dir_frag(const FieldArgs& a, const RawTile& r, P& bp) {
    float d0 = r.dir[0], d1 = r.dir[1], d2 = r.dir[2];
    if (a.normalize_dirs) {
        const float nn = sqrtf(fmaxf((d0 * d0 + d1 * d1) + d2 * d2, 1e-20f));
        d0 /= nn; d1 /= nn; d2 /= nn;
    }
    bp[0] = (_Float16)d0; bp[1] = (_Float16)d1; bp[2] = (_Float16)d2;
}

// ================================================================================================== forward
template <bool DO_DENSITY, bool DO_COLOR>
__global__ void __launch_bounds__(256) field_forward_kernel(FieldArgs a) {
    // These kernels run beside the next batch's marcher (second stream): their waves win the SIMD's issue arbitration
    __builtin_amdgcn_s_setprio(3);
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    stage_images<256, false, DO_DENSITY, DO_COLOR>(lds, a);

    const int lane = threadIdx.x & 63, n = lane & 31, g = lane >> 5;
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    const uint32_t n_tiles = (a.M + 31) / 32;
    RawTile nxt;
    float spec_sq = 0.f;             // this lane's sum of specular^2 over its samples (g = 0 lanes; a.spec_sq_partial)
    fetch_tile<DO_DENSITY, DO_COLOR, false>(a, wave, n, g, nxt);
    for (uint32_t tile = wave; tile < n_tiles; tile += n_waves) {
        const uint32_t s = tile * 32 + n;
        const bool valid = s < a.M;
        settle_tile(nxt);
        const RawTile cur = nxt;
        fetch_tile<DO_DENSITY, DO_COLOR, false>(a, tile + n_waves, n, g, nxt);      // next tile's inputs in flight under this tile's MFMA chains
        if (DO_DENSITY) {   // ---- density: [h1 | xyz] -> 32 -> 1 -> exp
            h4 b0[3];
            density_frags(cur, b0);
            f16x d = zero16();
            d = mm_k<3>(lds + O_S0, P_S0, 0, b0, d, lane);
            const h4 b1[4] = {relu_pack<0>(d), relu_pack<1>(d), relu_pack<2>(d), relu_pack<3>(d)};
            f16x o = zero16();
            o = mm_k<4>(lds + O_S1, P_S1, 0, b1, o, lane);
            if (valid && g == 0) a.sigma[s] = a.raw_density ? (float)(_Float16)o[0] : expf((float)(_Float16)o[0]);
        }
        if (DO_COLOR) {   // ---- colour: [h2 | xyz] -> 64 -> 64 -> 6 -> sigmoid ; specular: [d | feat] -> 32 -> 3 -> sigmoid
            h4 b0[5];
            color_frags(cur, b0);
            f16x d1[2] = {zero16(), zero16()};
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
                d1[mb] = mm_k<5>(lds + O_C0, P_C0, mb, b0, d1[mb], lane);
            const h4 b1[8] = {relu_pack<0>(d1[0]), relu_pack<1>(d1[0]), relu_pack<2>(d1[0]), relu_pack<3>(d1[0]),
                              relu_pack<0>(d1[1]), relu_pack<1>(d1[1]), relu_pack<2>(d1[1]), relu_pack<3>(d1[1])};
            f16x d2[2] = {zero16(), zero16()};
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
                d2[mb] = mm_k<8>(lds + O_C1, P_C1, mb, b1, d2[mb], lane);
            const h4 b2[8] = {relu_pack<0>(d2[0]), relu_pack<1>(d2[0]), relu_pack<2>(d2[0]), relu_pack<3>(d2[0]),
                              relu_pack<0>(d2[1]), relu_pack<1>(d2[1]), relu_pack<2>(d2[1]), relu_pack<3>(d2[1])};
            f16x d3 = zero16();
            d3 = mm_k<8>(lds + O_C2, P_C2, 0, b2, d3, lane);
            // rows 0..3 live in regs 0..3 of the g = 0 lanes, rows 4,5 in regs 0,1 of the g = 1 lanes
            const float q0 = sigmoid_h(d3[0]), q1 = sigmoid_h(d3[1]), q2 = sigmoid_h(d3[2]), q3 = sigmoid_h(d3[3]);
            float cr = q0, cg = q1, cb = q2;          // diffuse (g = 0 lanes)
            if (a.shading != 0) {
                h4 bp = zero4();                      // K order of specular_net layer 0: d0 d1 d2 f0 | f1 f2 0 0
                if (g == 0) {
                    if (valid) dir_frag(a, cur, bp);
                    bp[3] = (_Float16)q3;
                } else { bp[0] = (_Float16)q0; bp[1] = (_Float16)q1; }   // g = 1: q0,q1 are rows 4,5 = feat1, feat2
                f16x p1 = MFMA(ld_a(lds + O_P0, P_P0, 0, 0, lane), bp, zero16());
                const h4 bq[4] = {relu_pack<0>(p1), relu_pack<1>(p1), relu_pack<2>(p1), relu_pack<3>(p1)};
                f16x p2 = zero16();
                p2 = mm_k<4>(lds + O_P1, P_P1, 0, bq, p2, lane);
                const float s0 = sigmoid_h(p2[0]), s1 = sigmoid_h(p2[1]), s2 = sigmoid_h(p2[2]);
                if (valid && g == 0 && a.specular) {
                    a.specular[(size_t)s * 3] = s0; a.specular[(size_t)s * 3 + 1] = s1; a.specular[(size_t)s * 3 + 2] = s2;
                }
                if (valid && g == 0) spec_sq += (s0 * s0 + s1 * s1) + s2 * s2;      // fp32 squares of the fp16 outputs, like autocast's pow
                if (a.shading == 2) { cr = s0; cg = s1; cb = s2; }
                else {   // (specular + diffuse).clamp(0, 1) evaluated in fp16 like the autocast graph
                    cr = fminf(fmaxf((float)(_Float16)(s0 + q0), 0.f), 1.f);
                    cg = fminf(fmaxf((float)(_Float16)(s1 + q1), 0.f), 1.f);
                    cb = fminf(fmaxf((float)(_Float16)(s2 + q2), 0.f), 1.f);
                }
            }
            if (valid && g == 0 && a.rgb) {
                a.rgb[(size_t)s * 3] = cr; a.rgb[(size_t)s * 3 + 1] = cg; a.rgb[(size_t)s * 3 + 2] = cb;
            }
        }
    }
    if (DO_COLOR && a.spec_sq_partial) {
        // fixed order: lanes by the wave scan, waves 0..3, one slot per workgroup; workgroup 0 clears the slots no workgroup owns
        __shared__ float spec_wave[4];
        const float w = n2m_wave_sum(spec_sq);             // (all 64 lanes are active here: the tile loop is wave-uniform)
        if (lane == 0) spec_wave[threadIdx.x >> 6] = w;
        __syncthreads();
        if (threadIdx.x == 0) a.spec_sq_partial[blockIdx.x] = ((spec_wave[0] + spec_wave[1]) + spec_wave[2]) + spec_wave[3];
        if (blockIdx.x == 0)
            for (int i = (int)gridDim.x + (int)threadIdx.x; i < kSpecPartials; i += 256) a.spec_sq_partial[i] = 0.f;
    }
}

// ================================================================================================= backward
// per-wave transpose scratch: tile[sample][feature] fp16
__device__ __forceinline__ void tile_put(_Float16* tile, int kb, const h4& f, int lane) {
    *reinterpret_cast<h4*>(tile + (lane & 31) * TILE_PITCH + 8 * kb + 4 * (lane >> 5)) = f;
}
// operand of the sample-contraction MFMA: element i = tile[sample 8*kq + 4*(lane/32) + i][feature 32*fb + lane%32]
__device__ __forceinline__ h4 tile_get(const _Float16* tile, int kq, int fb, int lane) {
    const _Float16* p = tile + (8 * kq + 4 * (lane >> 5)) * TILE_PITCH + 32 * fb + (lane & 31);
    h4 r;
    r[0] = p[0]; r[1] = p[TILE_PITCH]; r[2] = p[2 * TILE_PITCH]; r[3] = p[3 * TILE_PITCH];
    return r;
}
// The same operand with ONE gfx950 transpose read instead of four 2-byte reads: the 16 lanes of a group (fixed lane/32 and feature
// half) hand ds_read_b64_tr_b16 the addresses of the 4-half pieces of a [4 samples][16 features] block (lane i: sample i/4, features
// 4(i%4)..+3; the row stride is free) and get the block back column-major -- lane c holds feature c of the four samples (tools/tr_lab.hip
// checks the mapping against tile_get element by element).
typedef short s4v_t __attribute__((__vector_size__(8)));
__device__ __forceinline__ h4 tile_get_tr(const _Float16* tile, int kq, int fb, int lane) {
    const _Float16* p = tile + (8 * kq + 4 * (lane >> 5) + ((lane & 15) >> 2)) * TILE_PITCH + 32 * fb + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    return __builtin_bit_cast(h4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v_t*)p));
}
// acc[mb][nb] += dY^T (features 32mb..) x X (features 32nb..) over the 32 samples of the tile
template <int MB, int NB, bool TR = false>
__device__ __forceinline__ void dw_tile(f16x (&acc)[MB][NB], const _Float16* tY, const _Float16* tX, int lane) {
    if constexpr (TR) {
        // all operands of the stage first (4 x (MB + NB) transpose reads, issued back to back), then the MFMAs
        h4 ay[4][MB], bx[4][NB];
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) ay[kq][mb] = tile_get_tr(tY, kq, mb, lane);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bx[kq][nb] = tile_get_tr(tX, kq, nb, lane);
        }
#pragma unroll
        for (int kq = 0; kq < 4; kq += 2)                 // 16 samples per instruction (32x32x16)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[mb][nb] = MFMA16(cat8(ay[kq][mb], ay[kq + 1][mb]), cat8(bx[kq][nb], bx[kq + 1][nb]), acc[mb][nb]);
        return;
    }
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
        h4 ay[MB], bx[NB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) ay[mb] = tile_get(tY, kq, mb, lane);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bx[nb] = tile_get(tX, kq, nb, lane);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = MFMA(ay[mb], bx[nb], acc[mb][nb]);
    }
}
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// accumulator tile -> LDS fp32 staging (logical [out_pad][in_pad]) with LDS atomics
template <int MB, int NB>
__device__ __forceinline__ void dw_to_lds(float* stage, int in_pad, const f16x (&acc)[MB][NB], int lane) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * mb + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3), col = 32 * nb + (lane & 31);
                __hip_atomic_fetch_add(&stage[row * in_pad + col], acc[mb][nb][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
}
// LDS staging -> global fp32 weight gradient [out, in] (nn.Linear layout), undoing the logical column order
__device__ void dw_flush(float* __restrict__ dW, const float* stage, int in_pad, int out, int in, int k_real, int perm, float* found_inf) {
    for (int idx = threadIdx.x; idx < out * k_real; idx += blockDim.x) {
        const int m = idx / k_real, k = idx - m * k_real;
        const int c = col_of(perm, k, in);
        const float v = stage[m * in_pad + k];
        if (c >= 0 && v != 0.f) {
            unsafeAtomicAdd(dW + m * in + c, v);
            if (!(fabsf(v) <= 3.0e38f) && found_inf) *found_inf = 1.0f;
        }
    }
}

// Second stage of the weight-gradient reduction.  Every workgroup used to add its 7 648 sums onto dW with global float atomics:
// 256 workgroups x 7 648 = 2 M atomics onto the same 240 cache lines, 64 us -- HALF the backward kernel (N2M_FIELD_DEBUG ablation).
// Now a workgroup stores its sums as one row of a [workgroups][7 648] scratch (plain coalesced stores) and this kernel adds the rows
// up in a fixed order (sixteen interleaved slices per element, then slice 0..15): ~2 M L2-resident loads, no atomics, and the result no
// longer depends on the order in which workgroups finish.
constexpr int kDwOff[8] = {0, 608, 640, 2880, 6976, 7360, 7552, 7648};      // sigma0 sigma1 color0 color1 color2 spec0 spec1
constexpr int kDwTotal = 7648;
struct DwOut { float* dw[7]; };
// accumulator tile of ONE consumer wave -> its own LDS copy (plain stores: four waves adding into one copy with ds_add_f32 measured
// 62 us per launch for 224 instructions per lane -- the LDS float atomic is that slow), then the four copies are summed in wave order
template <int MB, int NB>
__device__ __forceinline__ void dw_to_copy(float* copy, int in_pad, const f16x (&acc)[MB][NB], int lane) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * mb + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3), col = 32 * nb + (lane & 31);
                copy[row * in_pad + col] = acc[mb][nb][r];
            }
}
__device__ void dw_store_partial(float* __restrict__ part, const float* stage, int copy_stride, int in_pad, int out, int in, int k_real, int perm) {
    for (int idx = threadIdx.x; idx < out * k_real; idx += blockDim.x) {
        const int m = idx / k_real, k = idx - m * k_real;
        const int c = col_of(perm, k, in);
        const float* p = stage + m * in_pad + k;
        if (c >= 0) part[m * in + c] = ((p[0] + p[copy_stride]) + p[2 * copy_stride]) + p[3 * copy_stride];
    }
}
__global__ void __launch_bounds__(1024) dw_finalize_kernel(const float* __restrict__ part, uint32_t n_rows, DwOut o, uint32_t mask, float* found_inf) {
    __shared__ float sm[16][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    int mat = 0;
#pragma unroll
    for (int i = 1; i < 7; ++i) mat += e >= kDwOff[i] ? 1 : 0;
    const bool live = e < kDwTotal && ((mask >> mat) & 1u);
    float acc = 0.f;
    if (live) {
        // rows slice, slice + 16, ...: <= 16 per thread for 256 workgroups, all loads of a thread in flight together
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t w = (uint32_t)slice + 16u * i;
            v[i] = w < n_rows ? part[(size_t)w * kDwTotal + e] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += v[i];
        for (uint32_t w = (uint32_t)slice + 256u; w < n_rows; w += 16u) acc += part[(size_t)w * kDwTotal + e];     // larger grids (not used today)
    }
    sm[slice][lane] = acc;
    __syncthreads();
    if (slice == 0 && live) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) v += sm[i][lane];
        if (v != 0.f) {
            o.dw[mat][e - kDwOff[mat]] += v;
            if (!(fabsf(v) <= 3.0e38f) && found_inf) *found_inf = 1.0f;
        }
    }
}

template <bool DO_DENSITY, bool DO_COLOR>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) field_backward_kernel(FieldArgs a) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    if (DO_DENSITY) {
        stage_w(lds + O_S0, P_S0, a.w[0], 32, 19, 32, 24, PERM_SIGMA0);
        stage_w(lds + O_S1, P_S1, a.w[1], 1, 32, 32, 32, PERM_PLAIN);
        stage_wt(lds + O_S1T, P_S1T, a.w[1], 1, 32, 32, 8, PERM_PLAIN);
        stage_wt(lds + O_S0T, P_S0T, a.w[0], 32, 19, 32, 32, PERM_SIGMA0);
    }
    if (DO_COLOR) {
        stage_w(lds + O_C0, P_C0, a.w[2], 64, 35, 64, 40, PERM_COLOR0);
        stage_w(lds + O_C1, P_C1, a.w[3], 64, 64, 64, 64, PERM_PLAIN);
        stage_w(lds + O_C2, P_C2, a.w[4], 6, 64, 32, 64, PERM_PLAIN);
        stage_wt(lds + O_C2T, P_C2T, a.w[4], 6, 64, 64, 8, PERM_PLAIN);
        stage_wt(lds + O_C1T, P_C1T, a.w[3], 64, 64, 64, 64, PERM_PLAIN);
        stage_wt(lds + O_C0T, P_C0T, a.w[2], 64, 35, 64, 64, PERM_COLOR0);
        if (a.shading != 0) {
            stage_w(lds + O_P0, P_P0, a.w[5], 32, 6, 32, 8, PERM_PLAIN);
            stage_w(lds + O_P1, P_P1, a.w[6], 3, 32, 32, 32, PERM_PLAIN);
            stage_wt(lds + O_P1T, P_P1T, a.w[6], 3, 32, 32, 8, PERM_PLAIN);
            stage_wt(lds + O_P0T, P_P0T, a.w[5], 32, 6, 32, 32, PERM_PLAIN);
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, n = lane & 31, g = lane >> 5, wid = threadIdx.x >> 6;
    _Float16* tX = lds + ALL_W_HALVES + wid * 2 * TILE_HALVES;
    _Float16* tY = tX + TILE_HALVES;
    const uint32_t wave = blockIdx.x * 4 + wid, n_waves = gridDim.x * 4;
    const uint32_t n_tiles = (a.M + 31) / 32;
    const size_t Mz = a.M;
    const float spec_k = a.spec_reg != 0.f ? *a.seed * a.spec_reg : 0.f;     // seed gradient x 2 lambda / M

    // weight-gradient accumulators, fp32, live for the whole launch
    f16x gS0[1][1] = {{zero16()}}, gS1[1][1] = {{zero16()}};
    f16x gC0[2][2], gC1[2][2], gC2[1][2], gP0[1][1] = {{zero16()}}, gP1[1][1] = {{zero16()}};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { gC0[i][j] = zero16(); gC1[i][j] = zero16(); gC2[0][j] = zero16(); }

    for (uint32_t tile = wave; tile < n_tiles; tile += n_waves) {
        const uint32_t s = tile * 32 + n;
        const bool valid = s < a.M;
        if (DO_DENSITY) {   // --------------------------------------------------------------------------- density net
            h4 b0[3];
            load_density_inputs(a, s, valid, g, b0);
            f16x d = zero16();
#pragma unroll
            for (int kb = 0; kb < 3; ++kb) d = MFMA(ld_a(lds + O_S0, P_S0, 0, kb, lane), b0[kb], d);
            const h4 b1[4] = {relu_pack<0>(d), relu_pack<1>(d), relu_pack<2>(d), relu_pack<3>(d)};
            f16x o = zero16();
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) o = MFMA(ld_a(lds + O_S1, P_S1, 0, kb, lane), b1[kb], o);
            // trunc_exp backward: g * exp(clamp(x, -15, 15)) (activation.py:13-17), then into the fp16 Linear backward
            h4 dy1 = zero4();
            if (valid && g == 0) {
                const float pre = (float)(_Float16)o[0];
                dy1[0] = a.raw_density ? (_Float16)a.d_sigma[s] : (_Float16)(a.d_sigma[s] * expf(fminf(fmaxf(pre, -15.f), 15.f)));
            }
            // dW1 = dy1^T x H1
            wave_lds_fence();
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) { tile_put(tX, kb, b1[kb], lane); tile_put(tY, kb, kb == 0 ? dy1 : zero4(), lane); }
            wave_lds_fence();
            dw_tile<1, 1>(gS1, tY, tX, lane);
            // dH1 = W1^T dy1, masked
            const f16x dh = MFMA(ld_a(lds + O_S1T, P_S1T, 0, 0, lane), dy1, zero16());
            const h4 dy0[4] = {mask_pack<0>(dh, b1[0]), mask_pack<1>(dh, b1[1]), mask_pack<2>(dh, b1[2]), mask_pack<3>(dh, b1[3])};
            wave_lds_fence();
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) { tile_put(tY, kb, dy0[kb], lane); tile_put(tX, kb, kb < 3 ? b0[kb] : zero4(), lane); }
            wave_lds_fence();
            dw_tile<1, 1>(gS0, tY, tX, lane);
            // dX0 = W0^T dH1 : rows 0..15 are d h1 -> level-major [16][M] fp32 (values carry fp16 precision like autocast)
            f16x dx = zero16();
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) dx = MFMA(ld_a(lds + O_S0T, P_S0T, 0, kb, lane), dy0[kb], dx);
            if (valid) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int row = 8 * (r >> 2) + 4 * g + (r & 3);
                    a.d_h1[(size_t)row * Mz + s] = (float)(_Float16)dx[r];
                }
            }
        }
        if (DO_COLOR) {   // ---------------------------------------------------------------------------- colour + specular
            h4 b0[5];
            load_color_inputs(a, s, valid, g, b0);
            f16x d1[2] = {zero16(), zero16()};
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int kb = 0; kb < 5; ++kb) d1[mb] = MFMA(ld_a(lds + O_C0, P_C0, mb, kb, lane), b0[kb], d1[mb]);
            const h4 b1[8] = {relu_pack<0>(d1[0]), relu_pack<1>(d1[0]), relu_pack<2>(d1[0]), relu_pack<3>(d1[0]),
                              relu_pack<0>(d1[1]), relu_pack<1>(d1[1]), relu_pack<2>(d1[1]), relu_pack<3>(d1[1])};
            f16x d2[2] = {zero16(), zero16()};
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) d2[mb] = MFMA(ld_a(lds + O_C1, P_C1, mb, kb, lane), b1[kb], d2[mb]);
            const h4 b2[8] = {relu_pack<0>(d2[0]), relu_pack<1>(d2[0]), relu_pack<2>(d2[0]), relu_pack<3>(d2[0]),
                              relu_pack<0>(d2[1]), relu_pack<1>(d2[1]), relu_pack<2>(d2[1]), relu_pack<3>(d2[1])};
            f16x d3 = zero16();
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) d3 = MFMA(ld_a(lds + O_C2, P_C2, 0, kb, lane), b2[kb], d3);
            const float q0 = sigmoid_h(d3[0]), q1 = sigmoid_h(d3[1]), q2 = sigmoid_h(d3[2]), q3 = sigmoid_h(d3[3]);

            // upstream gradients (g = 0 lanes own sample s)
            float gr = 0.f, gg = 0.f, gb = 0.f, es0 = 0.f, es1 = 0.f, es2 = 0.f;
            if (valid && g == 0) {
                gr = a.d_rgb[(size_t)s * 3]; gg = a.d_rgb[(size_t)s * 3 + 1]; gb = a.d_rgb[(size_t)s * 3 + 2];
                if (a.d_specular && a.shading != 0) {
                    es0 = a.d_specular[(size_t)s * 3]; es1 = a.d_specular[(size_t)s * 3 + 1]; es2 = a.d_specular[(size_t)s * 3 + 2];
                }
            }
            float dq0 = 0.f, dq1 = 0.f, dq2 = 0.f, dq3 = 0.f;   // d geo rows 0..3 (g = 0) / rows 4,5 in dq0,dq1 (g = 1)
            if (a.shading == 0) { dq0 = gr; dq1 = gg; dq2 = gb; }
            else {
                h4 bp = zero4();
                if (g == 0) {
                    if (valid) load_dir(a, s, bp);
                    bp[3] = (_Float16)q3;
                } else { bp[0] = (_Float16)q0; bp[1] = (_Float16)q1; }
                const f16x p1 = MFMA(ld_a(lds + O_P0, P_P0, 0, 0, lane), bp, zero16());
                const h4 bq[4] = {relu_pack<0>(p1), relu_pack<1>(p1), relu_pack<2>(p1), relu_pack<3>(p1)};
                f16x p2 = zero16();
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) p2 = MFMA(ld_a(lds + O_P1, P_P1, 0, kb, lane), bq[kb], p2);
                const float s0 = sigmoid_h(p2[0]), s1 = sigmoid_h(p2[1]), s2 = sigmoid_h(p2[2]);
                float ts0 = es0, ts1 = es1, ts2 = es2;          // d loss / d specular, total
                if (valid && g == 0) { ts0 += s0 * spec_k; ts1 += s1 * spec_k; ts2 += s2 * spec_k; }     // + d (lambda mean sum specular^2); lanes without a sample contribute nothing
                if (a.shading == 2) { ts0 += gr; ts1 += gg; ts2 += gb; }
                else {   // clamp backward passes the gradient where 0 <= x <= 1
                    const float t0 = (float)(_Float16)(s0 + q0), t1 = (float)(_Float16)(s1 + q1), t2 = (float)(_Float16)(s2 + q2);
                    const float p0g = (t0 >= 0.f && t0 <= 1.f) ? gr : 0.f, p1g = (t1 >= 0.f && t1 <= 1.f) ? gg : 0.f,
                                p2g = (t2 >= 0.f && t2 <= 1.f) ? gb : 0.f;
                    ts0 += p0g; ts1 += p1g; ts2 += p2g;
                    dq0 = p0g; dq1 = p1g; dq2 = p2g;
                }
                h4 dsp = zero4();                                // d specular_net output (pre-sigmoid), rows 0..2
                if (g == 0) {
                    dsp[0] = (_Float16)(ts0 * s0 * (1.f - s0)); dsp[1] = (_Float16)(ts1 * s1 * (1.f - s1)); dsp[2] = (_Float16)(ts2 * s2 * (1.f - s2));
                }
                wave_lds_fence();
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) { tile_put(tX, kb, bq[kb], lane); tile_put(tY, kb, kb == 0 ? dsp : zero4(), lane); }
                wave_lds_fence();
                dw_tile<1, 1>(gP1, tY, tX, lane);
                const f16x dhp = MFMA(ld_a(lds + O_P1T, P_P1T, 0, 0, lane), dsp, zero16());
                const h4 dyp[4] = {mask_pack<0>(dhp, bq[0]), mask_pack<1>(dhp, bq[1]), mask_pack<2>(dhp, bq[2]), mask_pack<3>(dhp, bq[3])};
                wave_lds_fence();
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) { tile_put(tY, kb, dyp[kb], lane); tile_put(tX, kb, kb == 0 ? bp : zero4(), lane); }
                wave_lds_fence();
                dw_tile<1, 1>(gP0, tY, tX, lane);
                f16x dxp = zero16();
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) dxp = MFMA(ld_a(lds + O_P0T, P_P0T, 0, kb, lane), dyp[kb], dxp);
                // rows 3 (g=0, reg 3), 4 and 5 (g=1, regs 0,1) are d feat
                if (g == 0) dq3 = (float)(_Float16)dxp[3];
                else { dq0 = (float)(_Float16)dxp[0]; dq1 = (float)(_Float16)dxp[1]; }
            }
            // through the sigmoid of color_net's output
            h4 dc = zero4();
            if (g == 0) {
                dc[0] = (_Float16)(dq0 * q0 * (1.f - q0)); dc[1] = (_Float16)(dq1 * q1 * (1.f - q1));
                dc[2] = (_Float16)(dq2 * q2 * (1.f - q2)); dc[3] = (_Float16)(dq3 * q3 * (1.f - q3));
            } else { dc[0] = (_Float16)(dq0 * q0 * (1.f - q0)); dc[1] = (_Float16)(dq1 * q1 * (1.f - q1)); }
            if (!valid) dc = zero4();
            // layer 3: dW = dc^T x H2 ; dH2 = W3^T dc
            wave_lds_fence();
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) tile_put(tX, kb, b2[kb], lane);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) tile_put(tY, kb, kb == 0 ? dc : zero4(), lane);
            wave_lds_fence();
            dw_tile<1, 2>(gC2, tY, tX, lane);
            f16x e2[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) e2[mb] = MFMA(ld_a(lds + O_C2T, P_C2T, mb, 0, lane), dc, zero16());
            const h4 dy2[8] = {mask_pack<0>(e2[0], b2[0]), mask_pack<1>(e2[0], b2[1]), mask_pack<2>(e2[0], b2[2]), mask_pack<3>(e2[0], b2[3]),
                               mask_pack<0>(e2[1], b2[4]), mask_pack<1>(e2[1], b2[5]), mask_pack<2>(e2[1], b2[6]), mask_pack<3>(e2[1], b2[7])};
            // layer 2: dW = dy2^T x H1 ; dH1 = W2^T dy2
            wave_lds_fence();
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) { tile_put(tX, kb, b1[kb], lane); tile_put(tY, kb, dy2[kb], lane); }
            wave_lds_fence();
            dw_tile<2, 2>(gC1, tY, tX, lane);
            f16x e1[2] = {zero16(), zero16()};
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) e1[mb] = MFMA(ld_a(lds + O_C1T, P_C1T, mb, kb, lane), dy2[kb], e1[mb]);
            const h4 dy1[8] = {mask_pack<0>(e1[0], b1[0]), mask_pack<1>(e1[0], b1[1]), mask_pack<2>(e1[0], b1[2]), mask_pack<3>(e1[0], b1[3]),
                               mask_pack<0>(e1[1], b1[4]), mask_pack<1>(e1[1], b1[5]), mask_pack<2>(e1[1], b1[6]), mask_pack<3>(e1[1], b1[7])};
            // layer 1: dW = dy1^T x X0 ; d h2 = rows 0..31 of W1^T dy1
            wave_lds_fence();
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) { tile_put(tY, kb, dy1[kb], lane); tile_put(tX, kb, kb < 5 ? b0[kb] : zero4(), lane); }
            wave_lds_fence();
            dw_tile<2, 2>(gC0, tY, tX, lane);
            f16x e0 = zero16();
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) e0 = MFMA(ld_a(lds + O_C0T, P_C0T, 0, kb, lane), dy1[kb], e0);
            if (valid) {   // rows (2l, 2l+1) = level l of the C=2 encoder -> [16][M][2] fp16
                typedef _Float16 h2v __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int row = 8 * (r >> 2) + 4 * g + (r & 3);      // even
                    h2v v;
                    v.x = (_Float16)e0[r]; v.y = (_Float16)e0[r + 1];
                    *reinterpret_cast<h2v*>(a.d_h2 + ((size_t)(row >> 1) * Mz + s) * 2) = v;
                }
            }
        }
    }

    // ---------------------------------------------------------------------------- reduce dW: registers -> LDS -> HBM
    __syncthreads();
    float* stage = reinterpret_cast<float*>(lds);        // weights are dead now; 64 x 64 floats fit in the weight area
    auto reduce = [&](auto& acc, int out_pad, int in_pad, float* dW, int out, int in, int k_real, int perm) {
        for (int i = threadIdx.x; i < out_pad * in_pad; i += blockDim.x) stage[i] = 0.f;
        __syncthreads();
        dw_to_lds(stage, in_pad, acc, lane);
        __syncthreads();
        dw_flush(dW, stage, in_pad, out, in, k_real, perm, a.found_inf);
        __syncthreads();
    };
    if (DO_DENSITY) {
        reduce(gS1, 32, 32, a.dw[1], 1, 32, 32, PERM_PLAIN);
        reduce(gS0, 32, 32, a.dw[0], 32, 19, 19, PERM_SIGMA0);
    }
    if (DO_COLOR) {
        reduce(gC2, 32, 64, a.dw[4], 6, 64, 64, PERM_PLAIN);
        reduce(gC1, 64, 64, a.dw[3], 64, 64, 64, PERM_PLAIN);
        reduce(gC0, 64, 64, a.dw[2], 64, 35, 35, PERM_COLOR0);
        if (a.shading != 0) {
            reduce(gP1, 32, 32, a.dw[6], 3, 32, 32, PERM_PLAIN);
            reduce(gP0, 32, 32, a.dw[5], 32, 6, 6, PERM_PLAIN);
        }
    }
}

// Producer / consumer form of the backward: eight waves per workgroup = four PAIRS.  The producer wave of a pair runs the forward
// recompute and the activation-gradient chain of its 32-sample tile and, stage by stage, leaves the (X, dY) tiles of one layer in
// LDS; the consumer wave holds the fp32 weight-gradient accumulators (224 registers) and contracts them over the samples one stage
// behind (two tile buffers per pair, ONE workgroup barrier per stage).  Neither role needs more than 256 registers, so two waves
// share a SIMD and each hides the other's MFMA -> convert -> MFMA dependency stalls -- the single-wave kernel above spends 40 % of its
// cycles in issue stalls with nobody to switch to.  Same arithmetic in the same order per tile: results are bit-identical per
// accumulator; only the order in which tiles reach an accumulator (hence fp32 rounding of dW) differs with the tile-to-wave map.
template <bool DO_DENSITY, bool DO_COLOR>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) field_backward_pc_kernel(FieldArgs a) {
    __builtin_amdgcn_s_setprio(3);
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    stage_images<512, true, DO_DENSITY, DO_COLOR>(lds, a);

    const int lane = threadIdx.x & 63, n = lane & 31, g = lane >> 5, wid = threadIdx.x >> 6;
    const int pair = wid & 3;
    const bool producer = wid < 4;
    // stage st of the pair uses buffer st & 1: X tile then dY tile
    _Float16* const tbase = lds + ALL_W_HALVES + pair * 4 * TILE_HALVES;
    auto TX = [&](uint32_t st) { return tbase + (st & 1u) * 2 * TILE_HALVES; };
    auto TY = [&](uint32_t st) { return tbase + (st & 1u) * 2 * TILE_HALVES + TILE_HALVES; };
    const uint32_t n_tiles = (a.dbg & 1) ? 0u : (a.M + 31) / 32;
    const size_t Mz = a.M;
    uint32_t st = 0;                 // running stage counter, identical in both waves of a pair
    const float spec_k = a.spec_reg != 0.f ? *a.seed * a.spec_reg : 0.f;     // seed gradient x 2 lambda / M

    // weight-gradient accumulators, fp32, live for the whole launch
    f16x gS0[1][1] = {{zero16()}}, gS1[1][1] = {{zero16()}};
    f16x gC0[2][2], gC1[2][2], gC2[1][2], gP0[1][1] = {{zero16()}}, gP1[1][1] = {{zero16()}};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { gC0[i][j] = zero16(); gC1[i][j] = zero16(); gC2[0][j] = zero16(); }

    if (producer) {
    RawTile nxt;
    fetch_tile<DO_DENSITY, DO_COLOR, true>(a, blockIdx.x * 4 + pair, n, g, nxt);
    for (uint32_t base = blockIdx.x * 4; base < n_tiles; base += gridDim.x * 4) {
        const uint32_t tile = base + pair;      // may lie beyond the last tile: all lanes invalid, the stages still run (barriers)
        const uint32_t s = tile * 32 + n;
        const bool valid = s < a.M;
        settle_tile(nxt);
        const RawTile cur = nxt;
        fetch_tile<DO_DENSITY, DO_COLOR, true>(a, tile + gridDim.x * 4, n, g, nxt);   // next tile's inputs: requested now, used one iteration later
        if (DO_DENSITY) {   // --------------------------------------------------------------------------- density net
            h4 b0[3];
            density_frags(cur, b0);
            f16x d = zero16();
            d = mm_k<3>(lds + O_S0, P_S0, 0, b0, d, lane);
            const h4 b1[4] = {relu_pack<0>(d), relu_pack<1>(d), relu_pack<2>(d), relu_pack<3>(d)};
            f16x o = zero16();
            o = mm_k<4>(lds + O_S1, P_S1, 0, b1, o, lane);
            // trunc_exp backward: g * exp(clamp(x, -15, 15)) (activation.py:13-17), then into the fp16 Linear backward
            h4 dy1 = zero4();
            if (valid && g == 0) {
                const float pre = (float)(_Float16)o[0];
                dy1[0] = a.raw_density ? (_Float16)cur.dsig : (_Float16)(cur.dsig * expf(fminf(fmaxf(pre, -15.f), 15.f)));
            }
            // stage S1: dW1 = dy1^T x H1
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) { tile_put(TX(st), kb, b1[kb], lane); tile_put(TY(st), kb, kb == 0 ? dy1 : zero4(), lane); }
            __syncthreads(); ++st;
            // dH1 = W1^T dy1, masked
            const f16x dh = MFMA(ld_a(lds + O_S1T, P_S1T, 0, 0, lane), dy1, zero16());
            const h4 dy0[4] = {mask_pack<0>(dh, b1[0]), mask_pack<1>(dh, b1[1]), mask_pack<2>(dh, b1[2]), mask_pack<3>(dh, b1[3])};
            // stage S0
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) { tile_put(TY(st), kb, dy0[kb], lane); tile_put(TX(st), kb, kb < 3 ? b0[kb] : zero4(), lane); }
            __syncthreads(); ++st;
            // dX0 = W0^T dH1 : rows 0..15 are d h1 -> level-major [16][M] fp32 (values carry fp16 precision like autocast)
            f16x dx = zero16();
            dx = mm_k<4>(lds + O_S0T, P_S0T, 0, dy0, dx, lane);
            if (valid) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int row = 8 * (r >> 2) + 4 * g + (r & 3);
                    a.d_h1[(size_t)row * Mz + s] = (float)(_Float16)dx[r];
                }
            }
        }
        if (DO_COLOR) {   // ---------------------------------------------------------------------------- colour + specular
            h4 b0[5];
            h4 wC0[2][5], wC1[2][8], wC2[1][8];
            ld_layer(wC0, lds + O_C0, P_C0, lane);
            color_frags(cur, b0);
            ld_layer(wC1, lds + O_C1, P_C1, lane);
            __builtin_amdgcn_sched_barrier(0);                       // keep the reads above the MFMAs they are meant to hide behind
            f16x d1[2] = {zero16(), zero16()};
            mm_layer(d1, wC0, b0);
            ld_layer(wC2, lds + O_C2, P_C2, lane);
            __builtin_amdgcn_sched_barrier(0);
            const h4 b1[8] = {relu_pack<0>(d1[0]), relu_pack<1>(d1[0]), relu_pack<2>(d1[0]), relu_pack<3>(d1[0]),
                              relu_pack<0>(d1[1]), relu_pack<1>(d1[1]), relu_pack<2>(d1[1]), relu_pack<3>(d1[1])};
            f16x d2[2] = {zero16(), zero16()};
            mm_layer(d2, wC1, b1);
            const h4 b2[8] = {relu_pack<0>(d2[0]), relu_pack<1>(d2[0]), relu_pack<2>(d2[0]), relu_pack<3>(d2[0]),
                              relu_pack<0>(d2[1]), relu_pack<1>(d2[1]), relu_pack<2>(d2[1]), relu_pack<3>(d2[1])};
            f16x d3v[1] = {zero16()};
            mm_layer(d3v, wC2, b2);
            const f16x d3 = d3v[0];
            const float q0 = sigmoid_h(d3[0]), q1 = sigmoid_h(d3[1]), q2 = sigmoid_h(d3[2]), q3 = sigmoid_h(d3[3]);

            // upstream gradients (g = 0 lanes own sample s)
            float gr = 0.f, gg = 0.f, gb = 0.f, es0 = 0.f, es1 = 0.f, es2 = 0.f;
            if (valid && g == 0) {
                gr = cur.drgb[0]; gg = cur.drgb[1]; gb = cur.drgb[2];
                es0 = cur.dspec[0]; es1 = cur.dspec[1]; es2 = cur.dspec[2];       // zeros without d_specular / with diffuse shading
            }
            float dq0 = 0.f, dq1 = 0.f, dq2 = 0.f, dq3 = 0.f;   // d geo rows 0..3 (g = 0) / rows 4,5 in dq0,dq1 (g = 1)
            if (a.shading == 0) { dq0 = gr; dq1 = gg; dq2 = gb; }
            else {
                h4 bp = zero4();
                if (g == 0) {
                    if (valid) dir_frag(a, cur, bp);
                    bp[3] = (_Float16)q3;
                } else { bp[0] = (_Float16)q0; bp[1] = (_Float16)q1; }
                const f16x p1 = MFMA(ld_a(lds + O_P0, P_P0, 0, 0, lane), bp, zero16());
                const h4 bq[4] = {relu_pack<0>(p1), relu_pack<1>(p1), relu_pack<2>(p1), relu_pack<3>(p1)};
                f16x p2 = zero16();
                p2 = mm_k<4>(lds + O_P1, P_P1, 0, bq, p2, lane);
                const float s0 = sigmoid_h(p2[0]), s1 = sigmoid_h(p2[1]), s2 = sigmoid_h(p2[2]);
                float ts0 = es0, ts1 = es1, ts2 = es2;          // d loss / d specular, total
                if (valid && g == 0) { ts0 += s0 * spec_k; ts1 += s1 * spec_k; ts2 += s2 * spec_k; }     // + d (lambda mean sum specular^2); lanes without a sample contribute nothing
                if (a.shading == 2) { ts0 += gr; ts1 += gg; ts2 += gb; }
                else {   // clamp backward passes the gradient where 0 <= x <= 1
                    const float t0 = (float)(_Float16)(s0 + q0), t1 = (float)(_Float16)(s1 + q1), t2 = (float)(_Float16)(s2 + q2);
                    const float p0g = (t0 >= 0.f && t0 <= 1.f) ? gr : 0.f, p1g = (t1 >= 0.f && t1 <= 1.f) ? gg : 0.f,
                                p2g = (t2 >= 0.f && t2 <= 1.f) ? gb : 0.f;
                    ts0 += p0g; ts1 += p1g; ts2 += p2g;
                    dq0 = p0g; dq1 = p1g; dq2 = p2g;
                }
                h4 dsp = zero4();                                // d specular_net output (pre-sigmoid), rows 0..2
                if (g == 0) {
                    dsp[0] = (_Float16)(ts0 * s0 * (1.f - s0)); dsp[1] = (_Float16)(ts1 * s1 * (1.f - s1)); dsp[2] = (_Float16)(ts2 * s2 * (1.f - s2));
                }
                // stage P1
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) { tile_put(TX(st), kb, bq[kb], lane); tile_put(TY(st), kb, kb == 0 ? dsp : zero4(), lane); }
                __syncthreads(); ++st;
                const f16x dhp = MFMA(ld_a(lds + O_P1T, P_P1T, 0, 0, lane), dsp, zero16());
                const h4 dyp[4] = {mask_pack<0>(dhp, bq[0]), mask_pack<1>(dhp, bq[1]), mask_pack<2>(dhp, bq[2]), mask_pack<3>(dhp, bq[3])};
                // stage P0
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) { tile_put(TY(st), kb, dyp[kb], lane); tile_put(TX(st), kb, kb == 0 ? bp : zero4(), lane); }
                __syncthreads(); ++st;
                f16x dxp = zero16();
                dxp = mm_k<4>(lds + O_P0T, P_P0T, 0, dyp, dxp, lane);
                // rows 3 (g=0, reg 3), 4 and 5 (g=1, regs 0,1) are d feat
                if (g == 0) dq3 = (float)(_Float16)dxp[3];
                else { dq0 = (float)(_Float16)dxp[0]; dq1 = (float)(_Float16)dxp[1]; }
            }
            // through the sigmoid of color_net's output
            h4 dc = zero4();
            if (g == 0) {
                dc[0] = (_Float16)(dq0 * q0 * (1.f - q0)); dc[1] = (_Float16)(dq1 * q1 * (1.f - q1));
                dc[2] = (_Float16)(dq2 * q2 * (1.f - q2)); dc[3] = (_Float16)(dq3 * q3 * (1.f - q3));
            } else { dc[0] = (_Float16)(dq0 * q0 * (1.f - q0)); dc[1] = (_Float16)(dq1 * q1 * (1.f - q1)); }
            if (!valid) dc = zero4();
            // layer 3: dW = dc^T x H2 ; dH2 = W3^T dc
            // stage C2
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) tile_put(TX(st), kb, b2[kb], lane);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) tile_put(TY(st), kb, kb == 0 ? dc : zero4(), lane);
            __syncthreads(); ++st;
            h4 wC1T[2][8];
            ld_layer(wC1T, lds + O_C1T, P_C1T, lane);                 // next layer's operands: in flight under this one
            __builtin_amdgcn_sched_barrier(0);
            f16x e2[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) e2[mb] = MFMA(ld_a(lds + O_C2T, P_C2T, mb, 0, lane), dc, zero16());
            const h4 dy2[8] = {mask_pack<0>(e2[0], b2[0]), mask_pack<1>(e2[0], b2[1]), mask_pack<2>(e2[0], b2[2]), mask_pack<3>(e2[0], b2[3]),
                               mask_pack<0>(e2[1], b2[4]), mask_pack<1>(e2[1], b2[5]), mask_pack<2>(e2[1], b2[6]), mask_pack<3>(e2[1], b2[7])};
            // layer 2: dW = dy2^T x H1 ; dH1 = W2^T dy2
            // stage C1
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) { tile_put(TX(st), kb, b1[kb], lane); tile_put(TY(st), kb, dy2[kb], lane); }
            __syncthreads(); ++st;
            h4 wC0T[1][8];
            ld_layer(wC0T, lds + O_C0T, P_C0T, lane);
            __builtin_amdgcn_sched_barrier(0);
            f16x e1[2] = {zero16(), zero16()};
            mm_layer(e1, wC1T, dy2);
            const h4 dy1[8] = {mask_pack<0>(e1[0], b1[0]), mask_pack<1>(e1[0], b1[1]), mask_pack<2>(e1[0], b1[2]), mask_pack<3>(e1[0], b1[3]),
                               mask_pack<0>(e1[1], b1[4]), mask_pack<1>(e1[1], b1[5]), mask_pack<2>(e1[1], b1[6]), mask_pack<3>(e1[1], b1[7])};
            // layer 1: dW = dy1^T x X0 ; d h2 = rows 0..31 of W1^T dy1
            // stage C0
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) { tile_put(TY(st), kb, dy1[kb], lane); tile_put(TX(st), kb, kb < 5 ? b0[kb] : zero4(), lane); }
            __syncthreads(); ++st;
            f16x e0v[1] = {zero16()};
            mm_layer(e0v, wC0T, dy1);
            const f16x e0 = e0v[0];
            if (valid) {   // rows (2l, 2l+1) = level l of the C=2 encoder -> [16][M][2] fp16
                typedef _Float16 h2v __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int row = 8 * (r >> 2) + 4 * g + (r & 3);      // even
                    h2v v;
                    v.x = (_Float16)e0[r]; v.y = (_Float16)e0[r + 1];
                    *reinterpret_cast<h2v*>(a.d_h2 + ((size_t)(row >> 1) * Mz + s) * 2) = v;
                }
            }
        }
    }
    }
    else
    for (uint32_t base = blockIdx.x * 4; base < n_tiles; base += gridDim.x * 4) {
        // consumer: one barrier per stage (the pair's producer has filled buffer st & 1 by then), then the sample contraction
        if (DO_DENSITY) {
            __syncthreads(); dw_tile<1, 1, true>(gS1, TY(st), TX(st), lane); ++st;
            __syncthreads(); dw_tile<1, 1, true>(gS0, TY(st), TX(st), lane); ++st;
        }
        if (DO_COLOR) {
            if (a.shading != 0) {
                __syncthreads(); dw_tile<1, 1, true>(gP1, TY(st), TX(st), lane); ++st;
                __syncthreads(); dw_tile<1, 1, true>(gP0, TY(st), TX(st), lane); ++st;
            }
            __syncthreads(); dw_tile<1, 2, true>(gC2, TY(st), TX(st), lane); ++st;
            __syncthreads(); dw_tile<2, 2, true>(gC1, TY(st), TX(st), lane); ++st;
            __syncthreads(); dw_tile<2, 2, true>(gC0, TY(st), TX(st), lane); ++st;
        }
    }

    // ---------------------------------------------------------------------------- reduce dW: registers -> LDS -> HBM
    __syncthreads();
    float* stage = reinterpret_cast<float*>(lds);        // weights are dead now; 64 x 64 floats fit in the weight area
    float* const part_row = a.dw_partial + (size_t)blockIdx.x * kDwTotal;
    if (a.dbg & 2) return;
    // every consumer wave leaves its tiles in its own LDS copy (plain stores), the workgroup sums the four copies in wave order and
    // stores the row.  Two rounds: the two 64 x 64 layers (4 x 2 x 16 KB = 128 KB of the dead LDS), then the five small ones (96 KB).
    auto put = [&](auto& acc, int base, int out_pad, int in_pad, int total) {           // total: floats of one wave's copy in this round
        if (!producer) dw_to_copy(stage + pair * total + base, in_pad, acc, lane);
    };
    auto sum = [&](int base, int total, int in_pad, int mat, int out, int in, int k_real, int perm) {
        dw_store_partial(part_row + kDwOff[mat], stage + base, total, in_pad, out, in, k_real, perm);
    };
    if (DO_COLOR) {
        put(gC1, 0, 64, 64, 8192); put(gC0, 4096, 64, 64, 8192);
        __syncthreads();
        sum(0, 8192, 64, 3, 64, 64, 64, PERM_PLAIN); sum(4096, 8192, 64, 2, 64, 35, 35, PERM_COLOR0);
        __syncthreads();
    }
    constexpr int kSmall = 6144;       // C2 32x64 | S1 | S0 | P1 | P0 (32x32 each)
    if (DO_COLOR) {
        put(gC2, 0, 32, 64, kSmall);
        if (a.shading != 0) { put(gP1, 4096, 32, 32, kSmall); put(gP0, 5120, 32, 32, kSmall); }
    }
    if (DO_DENSITY) { put(gS1, 2048, 32, 32, kSmall); put(gS0, 3072, 32, 32, kSmall); }
    __syncthreads();
    if (DO_COLOR) {
        sum(0, kSmall, 64, 4, 6, 64, 64, PERM_PLAIN);
        if (a.shading != 0) { sum(4096, kSmall, 32, 6, 3, 32, 32, PERM_PLAIN); sum(5120, kSmall, 32, 5, 32, 6, 6, PERM_PLAIN); }
    }
    if (DO_DENSITY) { sum(2048, kSmall, 32, 1, 1, 32, 32, PERM_PLAIN); sum(3072, kSmall, 32, 0, 32, 19, 19, PERM_SIGMA0); }
}

int check_field(const char* fn, const float* xyz, const float* h1, const float* const* w, bool density, int shading) {
    N2M_REQUIRE(shading >= 0 && shading <= 2, N2M_EINVAL, "%s: shading must be 0 (diffuse), 1 (full) or 2 (specular)", fn);
    N2M_REQUIRE(xyz, N2M_ENULL, "%s: xyz is NULL", fn);
    if (density) N2M_REQUIRE(h1 && w[0] && w[1], N2M_ENULL, "%s: density branch needs h1 and sigma_net weights", fn);
    return 0;
}

// [256][7 648] floats per stream that ever ran a backward (calls on one stream are ordered, so they can share it); never freed
float* dw_scratch(hipStream_t s) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, float*> bufs;       // (device, stream): the null stream exists on every device
    std::lock_guard<std::mutex> lk(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    const auto key = std::make_pair(dev, s);
    auto it = bufs.find(key);
    if (it != bufs.end()) return it->second;
    float* p = nullptr;
    if (hipMalloc((void**)&p, (size_t)256 * kDwTotal * sizeof(float)) != hipSuccess) return nullptr;
    bufs[key] = p;
    return p;
}

uint32_t persistent_grid(uint32_t M) {
    const uint32_t tiles = (M + 31) / 32;
    const uint32_t need = (tiles + 3) / 4;
    return need < 256u ? (need ? need : 1u) : 256u;      // one 4-wave workgroup per CU, waves stride over the tiles
}

}  // namespace

static int field_forward_impl(const float* xyz, const float* dirs, const float* h1, const void* h2, const float* w_sigma0,
                                 const float* w_sigma1, const float* w_color0, const float* w_color1, const float* w_color2,
                                 const float* w_spec0, const float* w_spec1, uint32_t M, int shading, int normalize_dirs, float* sigma, float* rgb,
                                 float* specular, float* spec_sq_partial, void* stream) {
    const float* w[7] = {w_sigma0, w_sigma1, w_color0, w_color1, w_color2, w_spec0, w_spec1};
    const bool density = sigma != nullptr;
    if (int rc = check_field("field_forward", xyz, h1, w, density, shading)) return rc;
    const bool color = rgb != nullptr;
    N2M_REQUIRE(density || color, N2M_ENULL, "field_forward: both sigma and rgb are NULL");
    if (color) {
        N2M_REQUIRE(h2 && w[2] && w[3] && w[4], N2M_ENULL, "field_forward: colour branch needs h2 and color_net weights");
        if (shading != 0) N2M_REQUIRE(dirs && w[5] && w[6], N2M_ENULL, "field_forward: shading != 0 needs dirs and specular_net weights");
    }
    if (M == 0) return 0;
    FieldArgs a{};
    a.xyz = xyz; a.dirs = dirs; a.h1 = h1; a.h2 = (const _Float16*)h2; a.normalize_dirs = normalize_dirs & 1; a.raw_density = (normalize_dirs >> 1) & 1;
    for (int i = 0; i < 7; ++i) a.w[i] = w[i];
    a.M = M; a.shading = shading; a.sigma = sigma; a.rgb = rgb; a.specular = specular;
    a.spec_sq_partial = (color && shading != 0) ? spec_sq_partial : nullptr;
    hipStream_t s = (hipStream_t)stream;
    N2M_PROF_K(N2M_K_MLP_FWD, s, (double)M * (12 + 64 + 4 + (color ? 64 + 12 + 12 + 12 : 0)));
    const size_t smem = (size_t)FWD_HALVES * 2;
    if (color && density) N2M_LAUNCH((field_forward_kernel<true, true>), persistent_grid(M) * 2, 256, smem, s, a);
    else if (color) N2M_LAUNCH((field_forward_kernel<false, true>), persistent_grid(M) * 2, 256, smem, s, a);
    else N2M_LAUNCH((field_forward_kernel<true, false>), persistent_grid(M) * 2, 256, smem, s, a);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_field_forward(const float* xyz, const float* dirs, const float* h1, const void* h2, const float* w_sigma0,
                                 const float* w_sigma1, const float* w_color0, const float* w_color1, const float* w_color2,
                                 const float* w_spec0, const float* w_spec1, uint32_t M, int shading, int normalize_dirs, float* sigma, float* rgb,
                                 float* specular, void* stream) {
    return field_forward_impl(xyz, dirs, h1, h2, w_sigma0, w_sigma1, w_color0, w_color1, w_color2, w_spec0, w_spec1, M, shading, normalize_dirs,
                              sigma, rgb, specular, nullptr, stream);
}
extern "C" int n2m_field_forward_train(const float* xyz, const float* dirs, const float* h1, const void* h2, const float* w_sigma0,
                                       const float* w_sigma1, const float* w_color0, const float* w_color1, const float* w_color2,
                                       const float* w_spec0, const float* w_spec1, uint32_t M, int shading, int normalize_dirs, float* sigma,
                                       float* rgb, float* specular, float* spec_sq_partial, void* stream) {
    return field_forward_impl(xyz, dirs, h1, h2, w_sigma0, w_sigma1, w_color0, w_color1, w_color2, w_spec0, w_spec1, M, shading, normalize_dirs,
                              sigma, rgb, specular, spec_sq_partial, stream);
}
extern "C" uint32_t n2m_field_spec_partials(void) { return (uint32_t)kSpecPartials; }

static int field_backward_impl(const float* xyz, const float* dirs, const float* h1, const void* h2, const float* w_sigma0,
                                  const float* w_sigma1, const float* w_color0, const float* w_color1, const float* w_color2,
                                  const float* w_spec0, const float* w_spec1, uint32_t M, int shading, int normalize_dirs, const float* d_sigma,
                                  const float* d_rgb, const float* d_specular, float* d_h1, void* d_h2, float* d_w_sigma0,
                                  float* d_w_sigma1, float* d_w_color0, float* d_w_color1, float* d_w_color2, float* d_w_spec0,
                                  float* d_w_spec1, float* found_inf, float spec_reg, const float* seed, void* stream) {
    const float* w[7] = {w_sigma0, w_sigma1, w_color0, w_color1, w_color2, w_spec0, w_spec1};
    float* dw[7] = {d_w_sigma0, d_w_sigma1, d_w_color0, d_w_color1, d_w_color2, d_w_spec0, d_w_spec1};
    const bool density = d_sigma != nullptr;
    if (int rc = check_field("field_backward", xyz, h1, w, density, shading)) return rc;
    if (density) N2M_REQUIRE(d_h1 && dw[0] && dw[1], N2M_ENULL, "field_backward: density gradients are NULL");
    const bool color = d_rgb != nullptr;
    N2M_REQUIRE(density || color, N2M_ENULL, "field_backward: both d_sigma and d_rgb are NULL");
    if (color) {
        N2M_REQUIRE(h2 && d_h2 && w[2] && w[3] && w[4] && dw[2] && dw[3] && dw[4], N2M_ENULL, "field_backward: colour branch tensors are NULL");
        if (shading != 0) N2M_REQUIRE(dirs && w[5] && w[6] && dw[5] && dw[6], N2M_ENULL, "field_backward: specular branch tensors are NULL");
    }
    if (M == 0) return 0;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)field_backward_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, BWD_HALVES * 2);
        (void)hipFuncSetAttribute((const void*)field_backward_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, BWD_HALVES * 2);
        (void)hipFuncSetAttribute((const void*)field_backward_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, BWD_HALVES * 2);
        (void)hipFuncSetAttribute((const void*)field_backward_pc_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, BWD_PC_HALVES * 2);
        (void)hipFuncSetAttribute((const void*)field_backward_pc_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, BWD_PC_HALVES * 2);
        (void)hipFuncSetAttribute((const void*)field_backward_pc_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, BWD_PC_HALVES * 2);
        attr_set = true;
    }
    FieldArgs a{};
    a.xyz = xyz; a.dirs = dirs; a.h1 = h1; a.h2 = (const _Float16*)h2; a.normalize_dirs = normalize_dirs & 1; a.raw_density = (normalize_dirs >> 1) & 1;
    for (int i = 0; i < 7; ++i) { a.w[i] = w[i]; a.dw[i] = dw[i]; }
    a.M = M; a.shading = shading;
    a.spec_reg = (color && shading != 0 && seed) ? spec_reg : 0.f; a.seed = seed;
    a.d_sigma = d_sigma; a.d_rgb = d_rgb; a.d_specular = d_specular; a.d_h1 = d_h1; a.d_h2 = (_Float16*)d_h2; a.found_inf = found_inf;
    static const int dbg = getenv("N2M_FIELD_DEBUG") ? atoi(getenv("N2M_FIELD_DEBUG")) : 0;
    a.dbg = dbg;
    hipStream_t s = (hipStream_t)stream;
    N2M_PROF_K(N2M_K_MLP_BWD, s, (double)M * (12 + 64 + 4 + 64 + (color ? 64 + 12 + 24 + 64 : 0)));
    static const bool single_wave = getenv("N2M_FIELD_BWD_SINGLE") != nullptr;     // A/B switch: the one-wave-per-SIMD kernel
    if (!single_wave) {
        const size_t smem = (size_t)BWD_PC_HALVES * 2;
        const uint32_t grid = persistent_grid(M);
        float* part = dw_scratch(s);
        N2M_REQUIRE(part != nullptr, (int)hipErrorOutOfMemory, "field_backward: no memory for the weight-gradient scratch");
        a.dw_partial = part;
        if (color && density) N2M_LAUNCH((field_backward_pc_kernel<true, true>), grid, 512, smem, s, a);
        else if (color) N2M_LAUNCH((field_backward_pc_kernel<false, true>), grid, 512, smem, s, a);
        else N2M_LAUNCH((field_backward_pc_kernel<true, false>), grid, 512, smem, s, a);
        N2M_CHECK_LAUNCH();
        DwOut o;
        for (int i = 0; i < 7; ++i) o.dw[i] = dw[i];
        const uint32_t mask = (density ? 0x03u : 0u) | (color ? 0x1Cu : 0u) | (color && shading != 0 ? 0x60u : 0u);
        if (!(dbg & 2)) N2M_LAUNCH(dw_finalize_kernel, (kDwTotal + 63) / 64, 1024, 0, s, part, grid, o, mask, found_inf);
        N2M_CHECK_LAUNCH();
        return 0;
    }
    const size_t smem = (size_t)BWD_HALVES * 2;
    if (color && density) N2M_LAUNCH((field_backward_kernel<true, true>), persistent_grid(M), 256, smem, s, a);
    else if (color) N2M_LAUNCH((field_backward_kernel<false, true>), persistent_grid(M), 256, smem, s, a);
    else N2M_LAUNCH((field_backward_kernel<true, false>), persistent_grid(M), 256, smem, s, a);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_field_backward(const float* xyz, const float* dirs, const float* h1, const void* h2, const float* w_sigma0,
                                  const float* w_sigma1, const float* w_color0, const float* w_color1, const float* w_color2,
                                  const float* w_spec0, const float* w_spec1, uint32_t M, int shading, int normalize_dirs, const float* d_sigma,
                                  const float* d_rgb, const float* d_specular, float* d_h1, void* d_h2, float* d_w_sigma0,
                                  float* d_w_sigma1, float* d_w_color0, float* d_w_color1, float* d_w_color2, float* d_w_spec0,
                                  float* d_w_spec1, float* found_inf, void* stream) {
    return field_backward_impl(xyz, dirs, h1, h2, w_sigma0, w_sigma1, w_color0, w_color1, w_color2, w_spec0, w_spec1, M, shading, normalize_dirs,
                               d_sigma, d_rgb, d_specular, d_h1, d_h2, d_w_sigma0, d_w_sigma1, d_w_color0, d_w_color1, d_w_color2, d_w_spec0,
                               d_w_spec1, found_inf, 0.f, nullptr, stream);
}
extern "C" int n2m_field_backward_train(const float* xyz, const float* dirs, const float* h1, const void* h2, const float* w_sigma0,
                                        const float* w_sigma1, const float* w_color0, const float* w_color1, const float* w_color2,
                                        const float* w_spec0, const float* w_spec1, uint32_t M, int shading, int normalize_dirs,
                                        const float* d_sigma, const float* d_rgb, const float* d_specular, float* d_h1, void* d_h2,
                                        float* d_w_sigma0, float* d_w_sigma1, float* d_w_color0, float* d_w_color1, float* d_w_color2,
                                        float* d_w_spec0, float* d_w_spec1, float* found_inf, float spec_reg, const float* seed, void* stream) {
    return field_backward_impl(xyz, dirs, h1, h2, w_sigma0, w_sigma1, w_color0, w_color1, w_color2, w_spec0, w_spec1, M, shading, normalize_dirs,
                               d_sigma, d_rgb, d_specular, d_h1, d_h2, d_w_sigma0, d_w_sigma1, d_w_color0, d_w_color1, d_w_color2, d_w_spec0,
                               d_w_spec1, found_inf, spec_reg, seed, stream);
}
