// Multiresolution hash/tiled grid encoder for gfx950.  Replaces the reference's _gridencoder extension
// (gridencoder/src/gridencoder.cu); entry points are declared in include/n2m_hip.h.
//
// Layout in HBM: one table [rows, C] (fp32 or fp16) holding all levels back to back, level l at rows
// offsets[l]..offsets[l+1]; inputs [B, D] fp32 in [0,1]; features either LEVEL-major [L, B, C] (the reference
// layout) or SAMPLE-major [B, L*C] (the layout the network consumes).
//
// Per-level geometry (scale = 2^(l*S)*H - 1, resolution = ceil(scale)+1) is computed ON THE HOST with the same
// libm calls as oracle/n2m_oracle.c and handed to the kernels by value, so every lane of every kernel and the
// oracle agree on it bit for bit (a device exp2f that is 1 ulp off would move ceil() for near-integer scales).
// The file is compiled with -ffp-contract=off: interpolation weights and fp32 outputs match the oracle exactly;
// fp16 outputs round at the same points as the reference's at::Half accumulator (gridencoder.cu:163,186).
#include <math.h>
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "n2m_common.hpp"
#include "../../include/n2m_peer.h"

namespace {

constexpr uint32_t kMaxLevels = 32;

struct LevelTable {
    float scale[kMaxLevels];
    uint32_t resolution[kMaxLevels];
};

__host__ LevelTable make_levels(uint32_t L, float S, uint32_t H) {
    LevelTable t;
    for (uint32_t l = 0; l < kMaxLevels; ++l) {
        const float sc = l < L ? exp2f((float)l * S) * (float)H - 1.0f : 0.0f;
        t.scale[l] = sc;
        t.resolution[l] = (uint32_t)ceilf(sc) + 1u;
    }
    return t;
}

template <uint32_t D> struct Primes;
__device__ constexpr uint32_t kPrimes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};

// How a level maps a grid vertex to a table row; uniform per (kernel, level) so it lives in SGPRs.
template <uint32_t D>
struct Indexer {
    uint32_t stride[D];   // dense strides of the axes that take part (0 = axis dropped: table too small)
    uint32_t size;
    uint32_t mask;        // size-1 when size is a power of two
    bool hashed, pow2, wrap;

    __device__ __forceinline__ Indexer(uint32_t size_, uint32_t resolution, uint32_t gridtype, bool align_corners) : size(size_) {
        uint32_t s = 1;
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) {
            if (s <= size) {
                stride[d] = s;
                s *= align_corners ? resolution : resolution + 1u;
            } else {
                stride[d] = 0;
            }
        }
        hashed = (gridtype == 0) && s > size;
        pow2 = (size & (size - 1u)) == 0u;
        mask = size - 1u;
        // a dense index is already < size when every axis has resolution+1 entries (cell+1 <= resolution);
        // with align_corners an input of exactly 1.0 lands on entry `resolution`, so keep the reference's modulo
        wrap = hashed || s > size || align_corners;
    }

    __device__ __forceinline__ uint32_t row(const uint32_t (&v)[D]) const {
        uint32_t idx = 0;
        if (hashed) {
#pragma unroll
            for (uint32_t d = 0; d < D; ++d) idx ^= v[d] * kPrimes[d];
        } else {
#pragma unroll
            for (uint32_t d = 0; d < D; ++d) idx += v[d] * stride[d];
        }
        if (wrap) idx = pow2 ? (idx & mask) : (idx % size);
        return idx;
    }
};

// ---------------------------------------------------------------------------------------------- feature I/O
// A table row / feature vector of C channels, moved with the widest single access.
template <typename T, uint32_t C> struct Row;

template <uint32_t C>
struct Row<float, C> {
    float v[C];
    __device__ __forceinline__ static Row load(const float* p) {
        Row r;
        if constexpr (C == 1) r.v[0] = *p;
        else if constexpr (C == 2) { const float2 t = *reinterpret_cast<const float2*>(p); r.v[0] = t.x; r.v[1] = t.y; }
        else {
#pragma unroll
            for (uint32_t k = 0; k < C; k += 4) {
                const float4 t = *reinterpret_cast<const float4*>(p + k);
                r.v[k] = t.x; r.v[k + 1] = t.y; r.v[k + 2] = t.z; r.v[k + 3] = t.w;
            }
        }
        return r;
    }
    __device__ __forceinline__ void store(float* p) const {
        if constexpr (C == 1) *p = v[0];
        else if constexpr (C == 2) *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
        else {
#pragma unroll
            for (uint32_t k = 0; k < C; k += 4) *reinterpret_cast<float4*>(p + k) = make_float4(v[k], v[k + 1], v[k + 2], v[k + 3]);
        }
    }
};

template <uint32_t C>
struct Row<_Float16, C> {
    _Float16 v[C];
    __device__ __forceinline__ static Row load(const _Float16* p) {
        Row r;
        if constexpr (C == 1) r.v[0] = *p;
        else if constexpr (C == 2) { typedef _Float16 h2 __attribute__((ext_vector_type(2))); const h2 t = *reinterpret_cast<const h2*>(p); r.v[0] = t.x; r.v[1] = t.y; }
        else if constexpr (C == 4) { typedef _Float16 h4 __attribute__((ext_vector_type(4))); const h4 t = *reinterpret_cast<const h4*>(p); r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; }
        else {
            typedef _Float16 h8 __attribute__((ext_vector_type(8)));
            const h8 t = *reinterpret_cast<const h8*>(p);
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) r.v[k] = t[k];
        }
        return r;
    }
    __device__ __forceinline__ void store(_Float16* p) const {
        if constexpr (C == 1) *p = v[0];
        else if constexpr (C == 2) { typedef _Float16 h2 __attribute__((ext_vector_type(2))); h2 t; t.x = v[0]; t.y = v[1]; *reinterpret_cast<h2*>(p) = t; }
        else if constexpr (C == 4) { typedef _Float16 h4 __attribute__((ext_vector_type(4))); h4 t; t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3]; *reinterpret_cast<h4*>(p) = t; }
        else {
            typedef _Float16 h8 __attribute__((ext_vector_type(8)));
            h8 t;
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) t[k] = v[k];
            *reinterpret_cast<h8*>(p) = t;
        }
    }
};

// acc += w * g with the reference's rounding points: fp32 -> one mul + one add (no contraction); fp16 -> the
// product is rounded to half, then added in half (at::Half `+=` semantics).
__device__ __forceinline__ void accum(float& acc, float w, float g) { acc += w * g; }
__device__ __forceinline__ void accum(_Float16& acc, float w, _Float16 g) {
    const _Float16 p = (_Float16)(w * (float)g);
    acc = (_Float16)((float)acc + (float)p);
}

// (__half)(w * g) of the reference's backward (gridencoder.cu:326): the fp32 product is ROUNDED to fp32, then to half.  Without the
// barrier the compiler selects v_fma_mixlo_f16, which rounds the exact product once -- a different half in ~2^-14 of the cases.
__device__ __forceinline__ _Float16 half_product(float w, float g) {
    float p = w * g;
    asm volatile("" : "+v"(p));
    return (_Float16)p;
}

template <uint32_t D>
__device__ __forceinline__ bool outside_unit_cube(const float (&x)[D]) {
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) oob |= (x[d] < 0.0f) | (x[d] > 1.0f);
    return oob;
}

template <uint32_t D>
__device__ __forceinline__ void locate(const float (&x)[D], float scale, bool align_corners, uint32_t interp,
                                       uint32_t (&cell)[D], float (&frac)[D], float (&dfrac)[D]) {
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) {
        float p = x[d] * scale + (align_corners ? 0.0f : 0.5f);
        const float fl = floorf(p);
        cell[d] = (uint32_t)fl;
        p -= (float)cell[d];
        if (interp == 1) {
            dfrac[d] = 6 * p * (1.0f - p);
            p = p * p * (3.0f - 2.0f * p);
        } else {
            dfrac[d] = 1.0f;
        }
        frac[d] = p;
    }
}

template <uint32_t D>
__device__ __forceinline__ void load_point(const float* __restrict__ inputs, uint32_t b, float (&x)[D]) {
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) x[d] = inputs[(size_t)b * D + d];
}

// ------------------------------------------------------------------------------------------------ forward
// grid = (ceil(B/256), max_level): one level per blockIdx.y so that a block row works out of one level table.
template <typename T, uint32_t D, uint32_t C, bool SAMPLE_MAJOR>
__global__ void __launch_bounds__(256)
grid_forward_kernel(const float* __restrict__ inputs, const T* __restrict__ table, const int32_t* __restrict__ offsets,
                    T* __restrict__ outputs, uint32_t B, uint32_t L, LevelTable lv, T* __restrict__ dy_dx,
                    uint32_t gridtype, bool align_corners, uint32_t interp) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    const uint32_t level = blockIdx.y;
    if (b >= B) return;
    const uint32_t row0 = (uint32_t)offsets[level];
    const uint32_t size = (uint32_t)offsets[level + 1] - row0;
    const float scale = lv.scale[level];
    const Indexer<D> ix(size, lv.resolution[level], gridtype, align_corners);
    const T* __restrict__ tab = table + (size_t)row0 * C;
    T* out = SAMPLE_MAJOR ? outputs + ((size_t)b * L + level) * C : outputs + ((size_t)level * B + b) * C;
    T* gout = dy_dx ? dy_dx + ((size_t)b * L + level) * D * C : nullptr;

    float x[D];
    load_point<D>(inputs, b, x);
    if (outside_unit_cube<D>(x)) {
        Row<T, C> z;
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) z.v[c] = (T)0;
        z.store(out);
        if (gout) {
#pragma unroll
            for (uint32_t d = 0; d < D; ++d) z.store(gout + d * C);
        }
        return;
    }
    uint32_t cell[D];
    float frac[D], dfrac[D];
    locate<D>(x, scale, align_corners, interp, cell, frac, dfrac);

    // issue all 2^D gathers before consuming any of them
    Row<T, C> g[1u << D];
    float w[1u << D];
#pragma unroll
    for (uint32_t corner = 0; corner < (1u << D); ++corner) {
        uint32_t v[D];
        float ww = 1.0f;
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) {
            if (corner & (1u << d)) { ww *= frac[d]; v[d] = cell[d] + 1; }
            else { ww *= 1 - frac[d]; v[d] = cell[d]; }
        }
        w[corner] = ww;
        g[corner] = Row<T, C>::load(tab + (size_t)ix.row(v) * C);
    }
    Row<T, C> acc;
#pragma unroll
    for (uint32_t c = 0; c < C; ++c) acc.v[c] = (T)0;
#pragma unroll
    for (uint32_t corner = 0; corner < (1u << D); ++corner) {
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) accum(acc.v[c], w[corner], g[corner].v[c]);
    }
    acc.store(out);

    if (!gout) return;
    // d out / d x[gd]: the 2^D corner values are already in registers; pair them along axis gd
#pragma unroll
    for (uint32_t gd = 0; gd < D; ++gd) {
        Row<T, C> ga;
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) ga.v[c] = (T)0;
#pragma unroll
        for (uint32_t sub = 0; sub < (1u << (D - 1)); ++sub) {
            float ww = scale;
            uint32_t left = 0;
#pragma unroll
            for (uint32_t nd = 0; nd < D - 1; ++nd) {
                const uint32_t d = nd >= gd ? nd + 1 : nd;
                if (sub & (1u << nd)) { ww *= frac[d]; left |= 1u << d; }
                else ww *= 1 - frac[d];
            }
            const uint32_t right = left | (1u << gd);
#pragma unroll
            for (uint32_t c = 0; c < C; ++c) {
                if constexpr (sizeof(T) == 2) {
                    const T diff = (T)((float)g[right].v[c] - (float)g[left].v[c]);
                    const T p = (T)(ww * (float)diff * dfrac[gd]);
                    ga.v[c] = (T)((float)ga.v[c] + (float)p);
                } else {
                    ga.v[c] += ww * (g[right].v[c] - g[left].v[c]) * dfrac[gd];
                }
            }
        }
        ga.store(gout + gd * C);
    }
}

// ---------------------------------------------------------------------------------- forward, D = 3 fast path
// The generic kernel above issues 8 scattered row reads per (sample, level).  Measured (tools/grid_bench.py): on
// coherent samples it already runs at ~0.9 lane-requests/clk/CU, i.e. at the vector-memory request rate of a CU
// (one scattered lane per clock), so the only lever left is FEWER REQUESTS:
//   * the two corners that differ in x are neighbours in memory -- always for a dense level (row = x + y*s1 + z*s2),
//     and for a hashed level whenever x is even (x*1 ^ h differs from (x+1)*1 ^ h in bit 0 only).  One 8-byte access
//     (2 rows of C=1 fp32 or C=2 fp16; only dword alignment is required) fetches both: 4 requests instead of 8 for dense
//     levels, 6 on average for hashed ones;
//   * (experiment, N2M_GRID_FWD_XCD=1) XCD-aware level mapping: blocks numbered so that XCD k (= block id % 8, the observed
//     dispatch pattern) works on levels k and 15-k, so that each XCD's 4 MiB L2 holds exactly two 2 MiB level tables.
//     MEASURED SLOWER (2^21 coherent samples: 492 us vs 325 us level-major): the per-XCD work is unbalanced (hashed vs dense
//     levels) and a level's table is then served by ONE L2 instead of eight.  Kept off.
// Arithmetic is unchanged (same weights, same accumulation order over the 8 corners), so results stay bit-identical.
template <typename T, uint32_t C>
struct RowPair { Row<T, C> lo, hi; };

template <typename T, uint32_t C>
__device__ __forceinline__ RowPair<T, C> load_pair(const T* p) {   // rows r and r+1, contiguous
    RowPair<T, C> r;
    if constexpr (sizeof(T) * C == 4) {
        const uint2 v = *reinterpret_cast<const uint2*>(p);
        *reinterpret_cast<uint32_t*>(&r.lo) = v.x; *reinterpret_cast<uint32_t*>(&r.hi) = v.y;
    } else if constexpr (sizeof(T) * C == 8) {
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        uint2 a = make_uint2(v.x, v.y), b = make_uint2(v.z, v.w);
        *reinterpret_cast<uint2*>(&r.lo) = a; *reinterpret_cast<uint2*>(&r.hi) = b;
    } else {
        r.lo = Row<T, C>::load(p); r.hi = Row<T, C>::load(p + C);
    }
    return r;
}

template <typename T, uint32_t C, bool SAMPLE_MAJOR>
__global__ void __launch_bounds__(256)
grid_forward3_kernel(const float* __restrict__ inputs, const T* __restrict__ table, const int32_t* __restrict__ offsets,
                     T* __restrict__ outputs, uint32_t B, uint32_t L, uint32_t max_level, LevelTable lv, uint32_t gridtype,
                     bool align_corners, uint32_t interp, uint32_t n_tiles, bool xcd_map) {
    constexpr uint32_t D = 3;
    // block -> (tile, level): XCD k gets levels k, 15-k, 16+k, ... (boustrophedon so cheap coarse and expensive fine levels pair up)
    uint32_t level, tile;
    if (xcd_map) {
        const uint32_t lin = blockIdx.x, xcd = lin & 7u, j = lin >> 3;
        const uint32_t slots = (max_level + 7u) >> 3;
        const uint32_t slot = j % slots;
        tile = j / slots;
        level = (slot & 1u) ? 8u * (slot + 1u) - 1u - xcd : 8u * slot + xcd;
    } else {   // level-major block order of the generic kernel
        const uint32_t per_level = n_tiles;
        level = blockIdx.x / per_level;
        tile = blockIdx.x - level * per_level;
    }
    if (level >= max_level || tile >= n_tiles) return;
    const uint32_t b = tile * 256 + threadIdx.x;
    if (b >= B) return;
    const uint32_t row0 = (uint32_t)offsets[level];
    const uint32_t size = (uint32_t)offsets[level + 1] - row0;
    const float scale = lv.scale[level];
    const Indexer<D> ix(size, lv.resolution[level], gridtype, align_corners);
    const T* __restrict__ tab = table + (size_t)row0 * C;
    T* out = SAMPLE_MAJOR ? outputs + ((size_t)b * L + level) * C : outputs + ((size_t)level * B + b) * C;

    float x[D];
    load_point<D>(inputs, b, x);
    if (outside_unit_cube<D>(x)) {
        Row<T, C> z;
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) z.v[c] = (T)0;
        z.store(out);
        return;
    }
    uint32_t cell[D];
    float frac[D], dfrac[D];
    locate<D>(x, scale, align_corners, interp, cell, frac, dfrac);

    Row<T, C> g[8];
    const bool dense = !ix.hashed && !ix.wrap;
    if (dense) {
        const uint32_t base = cell[0] + cell[1] * ix.stride[1] + cell[2] * ix.stride[2];
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {          // q = (y bit, z bit); x pair in one access
            const uint32_t r = base + ((q & 1u) ? ix.stride[1] : 0u) + ((q & 2u) ? ix.stride[2] : 0u);
            const RowPair<T, C> pr = load_pair<T, C>(tab + (size_t)r * C);
            g[2 * q] = pr.lo; g[2 * q + 1] = pr.hi;
        }
    } else if (ix.hashed && ix.pow2) {
        const uint32_t hy0 = cell[1] * kPrimes[1], hy1 = hy0 + kPrimes[1], hz0 = cell[2] * kPrimes[2], hz1 = hz0 + kPrimes[2];
        const bool x_even = (cell[0] & 1u) == 0u;
        // 4 unconditional 8-byte reads of the ALIGNED row pair that holds corner x (rows r & ~1, r | 1); for even x the
        // other row of the pair is corner x+1 ((x+1) ^ h == (x ^ h) ^ 1).  For odd x one extra read per (y,z) fetches x+1.
        uint32_t rx[4], rx1[4];
        RowPair<T, C> pr[4];
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const uint32_t h = ((q & 1u) ? hy1 : hy0) ^ ((q & 2u) ? hz1 : hz0);
            rx[q] = (cell[0] ^ h) & ix.mask;
            rx1[q] = ((cell[0] + 1u) ^ h) & ix.mask;
            pr[q] = load_pair<T, C>(tab + (size_t)(rx[q] & ~1u) * C);
        }
        Row<T, C> extra[4];
        if (!x_even) {
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) extra[q] = Row<T, C>::load(tab + (size_t)rx1[q] * C);
        }
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const bool odd_row = (rx[q] & 1u) != 0u;
            g[2 * q] = odd_row ? pr[q].hi : pr[q].lo;
            g[2 * q + 1] = x_even ? (odd_row ? pr[q].lo : pr[q].hi) : extra[q];
        }
    } else {
#pragma unroll
        for (uint32_t corner = 0; corner < 8; ++corner) {
            const uint32_t v[D] = {cell[0] + (corner & 1u), cell[1] + ((corner >> 1) & 1u), cell[2] + (corner >> 2)};
            g[corner] = Row<T, C>::load(tab + (size_t)ix.row(v) * C);
        }
    }
    Row<T, C> acc;
#pragma unroll
    for (uint32_t c = 0; c < C; ++c) acc.v[c] = (T)0;
#pragma unroll
    for (uint32_t corner = 0; corner < 8; ++corner) {   // corner bit 0 = x, bit 1 = y, bit 2 = z: the reference's order
        float w = 1.0f;
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) w *= (corner & (1u << d)) ? frac[d] : 1 - frac[d];
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) accum(acc.v[c], w, g[corner].v[c]);
    }
    acc.store(out);
}

// zero-fill of the levels >= max_level for the sample-major layout
template <typename T>
__global__ void zero_levels_kernel(T* __restrict__ outputs, uint32_t B, uint32_t LC, uint32_t first, uint32_t count) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)B * count) return;
    const uint32_t b = (uint32_t)(t / count), k = (uint32_t)(t - (uint64_t)b * count);
    outputs[(size_t)b * LC + first + k] = (T)0;
}

// ----------------------------------------------------------------------------------------------- backward
__device__ __forceinline__ void atomic_add_row(float* dst, const float* v, uint32_t C) {
    for (uint32_t c = 0; c < C; ++c) unsafeAtomicAdd(dst + c, v[c]);
}

// TV weighting of the fused / stand-alone TV: weight (inner region) or weight_outer (|xyz|_inf > 1, nerf/utils.py:815-821),
// both times *scale_ptr when given (the GradScaler factor, so that the term can be added to still-scaled gradients).
struct TvParams {
    const float* table;       // fp32 [rows, 1]; NULL = no TV
    float weight, weight_outer, inner01;      // inner01: half extent of the inner region in [0,1] input space (>= 0.5: everything is inner)
    const float* scale_ptr;
    uint32_t stride = 1;      // floats between consecutive rows of `table`: 2 reads the density column of a packed {fp32, half2} table
    // [L, Bstride, 4] fp32 or NULL: per (level, sample) the table's values at corners 000 / 100 / 010 / 001 of the sample's interpolation cell, left
    // by the forward lookup of the SAME samples on the SAME table state (n2m_grid_encode_forward_packed_tv, hashed levels only).  They are the
    // centre and the +x / +y / +z neighbours of the TV stencil: the fill then gathers three neighbours instead of six (round 6).
    const float* corners = nullptr;
};

// (w = inner ? weight : weight_outer; w *= *scale_ptr; w /= 2 D -- the reference's operations, once per kernel: an IEEE division, ten
// instructions, per (sample, level) until round 4.)  One function for every kernel that forms TV terms: the fill, n2m_grid_tv_terms and the forward lookup
// that leaves finished terms (n2m_grid_encode_forward_packed_tvterms).
__device__ __forceinline__ void tv_weights(const TvParams& tv, float& w_in, float& w_out) {
    float wi = tv.weight, wo = tv.weight_outer;
    if (tv.table && tv.scale_ptr) { const float s = *tv.scale_ptr; wi *= s; wo *= s; }
    w_in = wi / 6.0f; w_out = wo / 6.0f;
}
__device__ __forceinline__ bool tv_inner(const TvParams& tv, const float (&x)[3]) {
    return fmaxf(fmaxf(fabsf(x[0] - 0.5f), fabsf(x[1] - 0.5f)), fabsf(x[2] - 0.5f)) <= tv.inner01;
}
// gridencoder.cu:505-609 behind the gathers: the sum over the in-grid neighbours in the reference's order (+x -x +y -y +z -z).
// rsqrtf like the reference (gridencoder.cu:606; an approximate intrinsic there too): v_rsq_f32, 1 ulp -- IEEE sqrt + IEEE division
// were ~22 instructions per (sample, level) in a kernel whose SIMDs are busy issuing VALU work more than half of the time
__device__ __forceinline__ float tv_reduce6(float centre, const float (&nb)[6], const bool (&nb_ok)[6], float w) {
    float sum = 0.f, sq = 0.f;
#pragma unroll
    for (uint32_t k = 0; k < 6; ++k)
        if (nb_ok[k]) { const float dv = centre - nb[k]; sum += dv; sq += dv * dv; }
    return w * sum * __builtin_amdgcn_rsqf(sq + 1e-9f);
}

// one thread per (sample, level): scatter w * grad into the 2^D rows
// The same forward on a PACKED table: row r = 8 bytes {fp32 density feature, 2 x fp16 colour features}.  The gathers of this
// kernel move one 64-byte line from L2 per vertex pair to use 8 bytes of it; with the two tables interleaved the same line
// serves both encoders, i.e. half the line traffic for the pair (the packed copy is kept fresh by the optimizer's update pass,
// n2m_adam_step shadow modes 2/3).  Arithmetic and outputs are identical to grid_forward3_pair_kernel.
//
// TVT (round 6, n2m_grid_encode_forward_packed_tvterms): the lookup also leaves the FINISHED total-variation term of the sample's cell,
// tv_out[level, b] = what the table backward's fill adds to vertex 000's density entry (gridencoder.cu:505-609; nerf/utils.py:800-823 applies it to the
// marched samples of the same step, on the same table state).  The TV cell floor(x * scale + 0.5) IS the interpolation cell's vertex 000, so four of
// the stencil's seven values -- centre and +x / +y / +z -- are corners 000 / 100 / 010 / 001, already in registers here; on a hashed level the -x
// neighbour of an odd cell is the other half of corner 000's 16-byte pair.  Left to gather: -y, -z and (even cells) -x.  The fill then reads 4 coalesced
// bytes per (sample, level) instead of gathering six rows inside its tile's dependent chain.  Same function of the same values as the in-place
// stencil (tv_reduce6, tv_weights): identical bits.
__device__ __forceinline__ float tv_term(const float* __restrict__ tab, const Indexer<3>& ix, uint32_t (&cell)[3], uint32_t here, uint32_t resolution,
                                         float w, uint32_t stride);
template <bool TVT>
__global__ void __launch_bounds__(256)
grid_forward3_packed_kernel(const float* __restrict__ inputs, const uint2* __restrict__ packed, const int32_t* __restrict__ offsets,
                            float* __restrict__ out1, _Float16* __restrict__ out2, uint32_t B, uint32_t max_level, LevelTable lv,
                            uint32_t gridtype, bool align_corners, uint32_t interp, uint32_t n_tiles, float in_scale, float in_offset,
                            uint32_t xcd_group, uint32_t level_begin, uint32_t n_levels, float4* __restrict__ tv4, TvParams tv,
                            float* __restrict__ tv_out) {
    __builtin_amdgcn_s_setprio(3);      // runs beside the next batch's marcher (second stream): win the issue arbitration
    constexpr uint32_t D = 3;
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    uint32_t level, tile;
    // (level_begin, n_levels): the levels this launch covers -- all of them, or one half of them when the packed rows arrive in two
    // chunks from the other ranks (n2m_grid_encode_forward_packed_levels); max_level then bounds level_begin + n_levels
    if (xcd_group == 0u) {
        level = blockIdx.x / n_tiles;
        tile = blockIdx.x - level * n_tiles;
        level += level_begin;
    } else {
        // XCD groups (max_level == 16).  Workgroup b runs on XCD b % 8 (observed dispatch rule; a wrong guess costs speed only).  The
        // kernel is bound by the L2s' request rate (tools/fwd_lab.hip: random 16-byte gathers from an L2-resident table run at 256 G
        // loads/s = 128 channels x 1 per clock, L1-resident at 800 G/s), so every level must stay spread over several L2s -- but each
        // L2 that works on a level also pulls that level's whole 4 MB table through the fabric.  g = 4 XCDs per level is the measured
        // optimum: group k = xcd / 4 owns eight levels (pairs p, 15 - p: coarse with fine) and walks them one after the other, its four
        // XCDs splitting the tiles.  67.8 -> 65.3 us stand-alone (8 XCDs per level -> 4; 2: 69.1, 1: 73.2).
        const uint32_t g = xcd_group, groups = 8u / g, per_group = n_levels / groups;
        const uint32_t xcd = blockIdx.x & 7u, k = xcd / g, j = xcd - k * g, i = blockIdx.x >> 3;
        const uint32_t tiles_per_xcd = (n_tiles + g - 1u) / g;
        const uint32_t li = i / tiles_per_xcd;
        tile = (i - li * tiles_per_xcd) * g + j;
        if (li >= per_group || tile >= n_tiles) return;
        const uint32_t pair = k + groups * (li >> 1);
        level = level_begin + ((li & 1u) ? n_levels - 1u - pair : pair);
    }
    if (level >= max_level) return;
    const uint32_t b = tile * 256 + threadIdx.x;
    if (b >= B) return;
    const uint32_t row0 = (uint32_t)offsets[level];
    const uint32_t size = (uint32_t)offsets[level + 1] - row0;
    const float scale = lv.scale[level];
    const Indexer<D> ix(size, lv.resolution[level], gridtype, align_corners);
    const uint2* __restrict__ tab = packed + (size_t)row0;
    float* o1 = out1 + (size_t)level * B + b;
    _Float16* o2 = out2 ? out2 + ((size_t)level * B + b) * 2 : nullptr;      // NULL: the density column alone (occupancy refresh)

    float x[D];
    load_point<D>(inputs, b, x);
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) x[d] = x[d] * in_scale + in_offset;
    if (outside_unit_cube<D>(x)) {
        *o1 = 0.0f;
        Row<_Float16, 2> z;
        z.v[0] = z.v[1] = (_Float16)0;
        if (o2) z.store(o2);
        if constexpr (TVT) tv_out[(size_t)level * B + b] = 0.0f;
        return;
    }
    uint32_t cell[D];
    float frac[D], dfrac[D];
    locate<D>(x, scale, align_corners, interp, cell, frac, dfrac);

    uint2 g[8];                                           // packed rows of the 8 vertices
    // TVT: the -x / -y / -z neighbours' density values (requested together with the corners), and whether the fast forms found them
    [[maybe_unused]] float tvm[3] = {0.f, 0.f, 0.f};
    [[maybe_unused]] bool tv_fast = true;
    [[maybe_unused]] const uint32_t tv_res = lv.resolution[level];
    [[maybe_unused]] const float* __restrict__ tabf = reinterpret_cast<const float*>(tab);
    const bool dense = !ix.hashed && !ix.wrap;
    if (dense) {
        const uint32_t base = cell[0] + cell[1] * ix.stride[1] + cell[2] * ix.stride[2];
        if constexpr (TVT) {
            tvm[0] = tabf[(size_t)(cell[0] > 0u ? base - 1u : base) * 2u];
            tvm[1] = tabf[(size_t)(cell[1] > 0u ? base - ix.stride[1] : base) * 2u];
            tvm[2] = tabf[(size_t)(cell[2] > 0u ? base - ix.stride[2] : base) * 2u];
        }
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const uint32_t r = base + ((q & 1u) ? ix.stride[1] : 0u) + ((q & 2u) ? ix.stride[2] : 0u);
            // rows r, r+1 are adjacent: ONE 16-byte load at an 8-byte-aligned address (gfx950 global loads need dword alignment only;
            // for odd r it used to be two 8-byte loads = two requests at the L1): 67.8 -> 63.9 us stand-alone
            const uint4 v = *reinterpret_cast<const uint4 __attribute__((aligned(8)))*>(tab + r);
            g[2 * q] = make_uint2(v.x, v.y); g[2 * q + 1] = make_uint2(v.z, v.w);
        }
    } else if (ix.hashed && ix.pow2) {
        const uint32_t hy0 = cell[1] * kPrimes[1], hy1 = hy0 + kPrimes[1], hz0 = cell[2] * kPrimes[2], hz1 = hz0 + kPrimes[2];
        const bool x_even = (cell[0] & 1u) == 0u;
        uint32_t rx[4], rx1[4];
        uint4 pr[4];
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const uint32_t h = ((q & 1u) ? hy1 : hy0) ^ ((q & 2u) ? hz1 : hz0);
            rx[q] = (cell[0] ^ h) & ix.mask;
            rx1[q] = ((cell[0] + 1u) ^ h) & ix.mask;
            pr[q] = *reinterpret_cast<const uint4*>(tab + (rx[q] & ~1u));
        }
        uint2 extra[4];
        if (!x_even) {
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) extra[q] = tab[rx1[q]];
        }
        if constexpr (TVT) {
            // the x prime is 1: the -x neighbour of an odd cell is row rx[0] ^ 1, the other half of corner 000's pair (filled in below)
            const uint32_t h00 = hy0 ^ hz0;
            if (x_even) tvm[0] = tabf[(size_t)(cell[0] > 0u ? ((cell[0] - 1u) ^ h00) & ix.mask : rx[0]) * 2u];
            tvm[1] = tabf[(size_t)(cell[1] > 0u ? (cell[0] ^ (hy0 - kPrimes[1]) ^ hz0) & ix.mask : rx[0]) * 2u];
            tvm[2] = tabf[(size_t)(cell[2] > 0u ? (cell[0] ^ hy0 ^ (hz0 - kPrimes[2])) & ix.mask : rx[0]) * 2u];
        }
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const bool odd_row = (rx[q] & 1u) != 0u;
            const uint2 lo = make_uint2(pr[q].x, pr[q].y), hi = make_uint2(pr[q].z, pr[q].w);
            g[2 * q] = odd_row ? hi : lo;
            g[2 * q + 1] = x_even ? (odd_row ? lo : hi) : extra[q];
            if constexpr (TVT) { if (q == 0u && !x_even) tvm[0] = __uint_as_float(odd_row ? lo.x : hi.x); }
        }
        // the density column of corners 000 / 100 / 010 / 001 = centre and +x / +y / +z neighbours of this sample's TV stencil (the TV cell
        // floor(x * scale + 0.5) IS the interpolation cell's vertex 000): one coalesced 16-byte record for the table backward's fill, which then
        // gathers three neighbours instead of six (n2m_grid_encode_forward_packed_tv; hashed levels only -- dense stencils hit the L1)
        if (tv4) tv4[(size_t)level * B + b] = make_float4(__uint_as_float(g[0].x), __uint_as_float(g[1].x), __uint_as_float(g[2].x), __uint_as_float(g[4].x));
    } else {
#pragma unroll
        for (uint32_t corner = 0; corner < 8; ++corner) {
            const uint32_t v[D] = {cell[0] + (corner & 1u), cell[1] + ((corner >> 1) & 1u), cell[2] + (corner >> 2)};
            g[corner] = tab[ix.row(v)];
        }
        tv_fast = false;
    }
    if constexpr (TVT) {
        float w_in, w_out;
        tv_weights(tv, w_in, w_out);
        const float w = tv_inner(tv, x) ? w_in : w_out;
        float tvv;
        if (tv_fast) {
            const float nb[6] = {__uint_as_float(g[1].x), tvm[0], __uint_as_float(g[2].x), tvm[1], __uint_as_float(g[4].x), tvm[2]};
            const bool nb_ok[6] = {cell[0] < tv_res, cell[0] > 0u, cell[1] < tv_res, cell[1] > 0u, cell[2] < tv_res, cell[2] > 0u};
            tvv = tv_reduce6(__uint_as_float(g[0].x), nb, nb_ok, w);
        } else {
            tvv = tv_term(tabf, ix, cell, ix.row(cell), tv_res, w, 2u);      // generic indexer (wrapping / non-power-of-two hash): the fill's IMODE 0
        }
        tv_out[(size_t)level * B + b] = tvv;
    }
    float a1 = 0.0f;
    _Float16 a2[2] = {(_Float16)0, (_Float16)0};
#pragma unroll
    for (uint32_t corner = 0; corner < 8; ++corner) {   // corner bit 0 = x, bit 1 = y, bit 2 = z: the reference's order
        float w = 1.0f;
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) w *= (corner & (1u << d)) ? frac[d] : 1 - frac[d];
        const h2 c2 = __builtin_bit_cast(h2, g[corner].y);
        accum(a1, w, __uint_as_float(g[corner].x));
        accum(a2[0], w, c2.x);
        accum(a2[1], w, c2.y);
    }
    *o1 = a1;
    if (!o2) return;
    Row<_Float16, 2> r2;
    r2.v[0] = a2[0]; r2.v[1] = a2[1];
    r2.store(o2);
}

// Both tables of the field in one forward: the density (fp32 C=1) and colour (fp16 C=2) encoders share geometry and inputs
// (nerf/network.py:92-108), so cell, weights and row indices are derived once and each vertex pair is fetched from both tables.
// Same arithmetic and rounding points as grid_forward3_kernel per table: outputs are bit-identical to two single calls.
// Level-major outputs out1 [L,B] f32, out2 [L,B,2] f16.
__global__ void __launch_bounds__(256)
grid_forward3_pair_kernel(const float* __restrict__ inputs, const float* __restrict__ table1, const _Float16* __restrict__ table2,
                          const int32_t* __restrict__ offsets, float* __restrict__ out1, _Float16* __restrict__ out2, uint32_t B,
                          uint32_t max_level, LevelTable lv, uint32_t gridtype, bool align_corners, uint32_t interp, uint32_t n_tiles,
                          float in_scale, float in_offset) {
    constexpr uint32_t D = 3;
    const uint32_t level = blockIdx.x / n_tiles, tile = blockIdx.x - level * n_tiles;
    if (level >= max_level) return;
    const uint32_t b = tile * 256 + threadIdx.x;
    if (b >= B) return;
    const uint32_t row0 = (uint32_t)offsets[level];
    const uint32_t size = (uint32_t)offsets[level + 1] - row0;
    const float scale = lv.scale[level];
    const Indexer<D> ix(size, lv.resolution[level], gridtype, align_corners);
    const float* __restrict__ t1 = table1 + (size_t)row0;
    const _Float16* __restrict__ t2 = table2 + (size_t)row0 * 2;
    float* o1 = out1 + (size_t)level * B + b;
    _Float16* o2 = out2 + ((size_t)level * B + b) * 2;

    float x[D];
    load_point<D>(inputs, b, x);
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) x[d] = x[d] * in_scale + in_offset;     // (1, 0) = inputs already in [0,1]; see n2m_hip.h
    if (outside_unit_cube<D>(x)) {
        *o1 = 0.0f;
        Row<_Float16, 2> z;
        z.v[0] = z.v[1] = (_Float16)0;
        z.store(o2);
        return;
    }
    uint32_t cell[D];
    float frac[D], dfrac[D];
    locate<D>(x, scale, align_corners, interp, cell, frac, dfrac);

    Row<float, 1> g1[8];
    Row<_Float16, 2> g2[8];
    const bool dense = !ix.hashed && !ix.wrap;
    if (dense) {
        const uint32_t base = cell[0] + cell[1] * ix.stride[1] + cell[2] * ix.stride[2];
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const uint32_t r = base + ((q & 1u) ? ix.stride[1] : 0u) + ((q & 2u) ? ix.stride[2] : 0u);
            const RowPair<float, 1> p1 = load_pair<float, 1>(t1 + (size_t)r);
            const RowPair<_Float16, 2> p2 = load_pair<_Float16, 2>(t2 + (size_t)r * 2);
            g1[2 * q] = p1.lo; g1[2 * q + 1] = p1.hi;
            g2[2 * q] = p2.lo; g2[2 * q + 1] = p2.hi;
        }
    } else if (ix.hashed && ix.pow2) {
        const uint32_t hy0 = cell[1] * kPrimes[1], hy1 = hy0 + kPrimes[1], hz0 = cell[2] * kPrimes[2], hz1 = hz0 + kPrimes[2];
        const bool x_even = (cell[0] & 1u) == 0u;
        uint32_t rx[4], rx1[4];
        RowPair<float, 1> p1[4];
        RowPair<_Float16, 2> p2[4];
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const uint32_t h = ((q & 1u) ? hy1 : hy0) ^ ((q & 2u) ? hz1 : hz0);
            rx[q] = (cell[0] ^ h) & ix.mask;
            rx1[q] = ((cell[0] + 1u) ^ h) & ix.mask;
            p1[q] = load_pair<float, 1>(t1 + (size_t)(rx[q] & ~1u));
            p2[q] = load_pair<_Float16, 2>(t2 + (size_t)(rx[q] & ~1u) * 2);
        }
        Row<float, 1> e1[4];
        Row<_Float16, 2> e2[4];
        if (!x_even) {
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) {
                e1[q] = Row<float, 1>::load(t1 + (size_t)rx1[q]);
                e2[q] = Row<_Float16, 2>::load(t2 + (size_t)rx1[q] * 2);
            }
        }
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const bool odd_row = (rx[q] & 1u) != 0u;
            g1[2 * q] = odd_row ? p1[q].hi : p1[q].lo;
            g1[2 * q + 1] = x_even ? (odd_row ? p1[q].lo : p1[q].hi) : e1[q];
            g2[2 * q] = odd_row ? p2[q].hi : p2[q].lo;
            g2[2 * q + 1] = x_even ? (odd_row ? p2[q].lo : p2[q].hi) : e2[q];
        }
    } else {
#pragma unroll
        for (uint32_t corner = 0; corner < 8; ++corner) {
            const uint32_t v[D] = {cell[0] + (corner & 1u), cell[1] + ((corner >> 1) & 1u), cell[2] + (corner >> 2)};
            const uint32_t r = ix.row(v);
            g1[corner] = Row<float, 1>::load(t1 + (size_t)r);
            g2[corner] = Row<_Float16, 2>::load(t2 + (size_t)r * 2);
        }
    }
    float a1 = 0.0f;
    _Float16 a2[2] = {(_Float16)0, (_Float16)0};
#pragma unroll
    for (uint32_t corner = 0; corner < 8; ++corner) {   // corner bit 0 = x, bit 1 = y, bit 2 = z: the reference's order
        float w = 1.0f;
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) w *= (corner & (1u << d)) ? frac[d] : 1 - frac[d];
        accum(a1, w, g1[corner].v[0]);
        accum(a2[0], w, g2[corner].v[0]);
        accum(a2[1], w, g2[corner].v[1]);
    }
    *o1 = a1;
    Row<_Float16, 2> r2;
    r2.v[0] = a2[0]; r2.v[1] = a2[1];
    r2.store(o2);
}

template <typename T, uint32_t D, uint32_t C, bool SAMPLE_MAJOR>
__global__ void __launch_bounds__(256)
grid_backward_kernel(const T* __restrict__ grad, const float* __restrict__ inputs, const int32_t* __restrict__ offsets,
                     T* __restrict__ grad_table, uint32_t B, uint32_t L, LevelTable lv, uint32_t gridtype,
                     bool align_corners, uint32_t interp) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    const uint32_t level = blockIdx.y;
    if (b >= B) return;
    const uint32_t row0 = (uint32_t)offsets[level];
    const uint32_t size = (uint32_t)offsets[level + 1] - row0;
    const Indexer<D> ix(size, lv.resolution[level], gridtype, align_corners);
    T* __restrict__ gtab = grad_table + (size_t)row0 * C;

    float x[D];
    load_point<D>(inputs, b, x);
    if (outside_unit_cube<D>(x)) return;
    uint32_t cell[D];
    float frac[D], dfrac[D];
    locate<D>(x, lv.scale[level], align_corners, interp, cell, frac, dfrac);
    const Row<T, C> gr = Row<T, C>::load(SAMPLE_MAJOR ? grad + ((size_t)b * L + level) * C : grad + ((size_t)level * B + b) * C);

#pragma unroll
    for (uint32_t corner = 0; corner < (1u << D); ++corner) {
        uint32_t v[D];
        float w = 1.0f;
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) {
            if (corner & (1u << d)) { w *= frac[d]; v[d] = cell[d] + 1; }
            else { w *= 1 - frac[d]; v[d] = cell[d]; }
        }
        T* dst = gtab + (size_t)ix.row(v) * C;
        if constexpr (sizeof(T) == 2) {
            // packed fp16 atomics (global_atomic_pk_add_f16), two channels per instruction (gridencoder.cu:324-330)
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (uint32_t c = 0; c < C; c += 2) {
                h2 val;
                val.x = half_product(w, (float)gr.v[c]);
                val.y = half_product(w, (float)gr.v[c + 1]);
                (void)__builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) h2*)(dst + c), val);
            }
        } else {
#pragma unroll
            for (uint32_t c = 0; c < C; ++c) unsafeAtomicAdd(dst + c, w * gr.v[c]);
        }
    }
}


// ------------------------------------------------------------------------- backward, LDS-privatised (D = 3)
// Measured on MI355X (tools/atomic_bench.hip): a float atomic costs one request per DISTINCT 64-byte line per
// wave instruction, ~21 G requests/s chip-wide whatever the scope or the XCD; 16 lanes on one line merge into a
// single request (317 G atomics/s).  The per-sample scatter of the kernel above issues 8 scattered atomics per
// (sample, level) = 33.5 M line requests for 2^18 samples -> ~2 ms.  This kernel removes them from the hot loop:
//
//   work item = (level, partition of <= P table rows, group of samples); one 1024-thread workgroup per item keeps
//   the partition's gradient in LDS as fp32 (P*C*4 = 128 KiB), sweeps its samples, recomputes the 8 vertex rows
//   and ds_add's the ones that fall into its partition, then flushes the partition with COALESCED global atomics
//   (consecutive lanes -> consecutive rows -> 16 lanes per 64 B line).
//
// Re-deriving the rows once per partition costs ALU only (the per-axis hash / stride terms are hoisted, so a corner
// is two xors/adds, a mask and a range test); scattered traffic stays inside the CU.  Consecutive samples of a ray
// share cells at the coarse levels, which would serialise in the LDS atomic unit if they sat in adjacent lanes, so
// each lane starts its 8 corners at a different one (lane % 8) there, spreading a same-cell run over its 8 rows.
// fp16 tables: products are accumulated in fp32 and rounded to half once per flush (the reference rounds every
// product to half and adds in half, gridencoder.cu:324-330 -- this is strictly more accurate, same expectation).
// Gradients are read LEVEL-major [L,B,C]; a sample-major producer is transposed first (33 MB, ~10 us).

template <typename T>
__global__ void transpose_to_level_major_kernel(const T* __restrict__ src /*[B, LC]*/, T* __restrict__ dst /*[LC/C... as [L,B,C]]*/,
                                                uint32_t B, uint32_t L, uint32_t C) {
    // thread -> (j = l*C + c, b) with b fastest: coalesced writes, line-granular (L1/L2-absorbed) reads
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t LC = L * C;
    if (t >= (uint64_t)B * LC) return;
    const uint32_t j = (uint32_t)(t / B), b = (uint32_t)(t - (uint64_t)j * B);
    const uint32_t l = j / C, c = j - l * C;
    dst[((size_t)l * B + b) * C + c] = src[(size_t)b * LC + j];
}

constexpr uint32_t kLdsBytes = 131072;   // accumulator bytes per workgroup (one workgroup per CU)

// Which table rows a work item owns.  Rows are dealt to the `parts` partitions of a level in 16-row blocks (one 64-byte
// line of fp32 C=1 / fp16 C=2 entries), round-robin: block b belongs to partition b % parts and sits at local block
// b / parts.  Interleaving keeps the partitions of the DENSE levels balanced (their row index is x + y*s1 + z*s2, so
// contiguous ranges would be z-slabs and a compact scene would pile onto one of them) while the flush still writes
// whole lines.  parts is a power of two for the hashed levels (shift/mask), otherwise a reciprocal multiply.
struct PartMap {
    uint32_t parts, part, magic, log2p;
    bool interleaved;
    // contiguous: partition = row / P (P = 2^log2p rows); used for the hashed levels, whose rows are uniform anyway.
    //   (Interleaving those was measured SLOWER: a partition's lines then share their low address bits and its flush
    //    lands on one memory channel.)
    // interleaved: 16-row blocks dealt round-robin; used for dense levels that span several partitions.
    __device__ __forceinline__ PartMap(uint32_t parts_, uint32_t part_, uint32_t log2p_, bool interleaved_)
        : parts(parts_), part(part_), log2p(log2p_), interleaved(interleaved_) {
        magic = 0xFFFFFFFFu / parts + 1u;              // exact quotient for block indices < 2^32 / parts
    }
    __device__ __forceinline__ bool mine(uint32_t row, uint32_t& rel) const {
        if (!interleaved) {
            if ((row >> log2p) != part) return false;
            rel = row & ((1u << log2p) - 1u);
            return true;
        }
        const uint32_t blk = row >> 4;
        const uint32_t q = __umulhi(blk, magic);
        if (blk - q * parts != part) return false;
        rel = (q << 4) | (row & 15u);
        return true;
    }
    // which partition a row belongs to and where it sits inside it (independent of `part`)
    __device__ __forceinline__ void split(uint32_t row, uint32_t& p, uint32_t& rel) const {
        if (!interleaved) {
            p = row >> log2p;
            rel = row & ((1u << log2p) - 1u);
        } else {
            const uint32_t blk = row >> 4;
            const uint32_t q = __umulhi(blk, magic);
            p = blk - q * parts;
            rel = (q << 4) | (row & 15u);
        }
    }
    __device__ __forceinline__ uint32_t global_row(uint32_t rel) const {
        return interleaved ? ((((rel >> 4) * parts + part) << 4) | (rel & 15u)) : ((part << log2p) + rel);
    }
};

// MODE 0: generic Indexer::row (tiled grids with a wrap, non power-of-two hashed tables)
// MODE 1: hashed level with a power-of-two table: row = (x*1 ^ y*p1 ^ z*p2) & mask
// MODE 2: dense level without wrap:               row = x + y*s1 + z*s2
template <typename T, uint32_t C, int MODE>
__device__ __forceinline__ void backward_sweep(float* acc, const T* __restrict__ glevel, const float* __restrict__ inputs,
                                               const Indexer<3>& ix, float scale, bool align_corners, uint32_t interp,
                                               uint32_t s_first, uint32_t run, uint32_t s_end, const PartMap pm) {
    constexpr uint32_t D = 3;
    const uint32_t lane = threadIdx.x & 63u;
    // coalesced assignment: at iteration `it` the block covers samples s_first + it*1024 .. +1023 (s_first = group
    // start + thread id).  Consecutive samples of a ray share a cell at the coarse (dense) levels, so adjacent lanes would
    // hit the SAME LDS word corner after corner; for those levels each lane starts at a different corner (lane % 8), which
    // spreads a same-cell run of lanes over its 8 rows and cuts the LDS atomic serialisation by ~8x at no extra traffic.
    // the loop is latency-bound (one dependent load round trip per iteration), so the NEXT iteration's point and
    // gradient row are requested before the current one is consumed
    float xn[D] = {2.f, 2.f, 2.f};
    Row<T, C> gn;
#pragma unroll
    for (uint32_t c = 0; c < C; ++c) gn.v[c] = (T)0;
    if (s_first < s_end) { load_point<D>(inputs, s_first, xn); gn = Row<T, C>::load(glevel + (size_t)s_first * C); }
    for (uint32_t it = 0; it < run; ++it) {
        const uint32_t s = s_first + it * 1024u;
        if (s >= s_end) break;
        float x[D] = {xn[0], xn[1], xn[2]};
        const Row<T, C> gr = gn;
        const uint32_t s2 = s + 1024u;
        if (it + 1 < run && s2 < s_end) { load_point<D>(inputs, s2, xn); gn = Row<T, C>::load(glevel + (size_t)s2 * C); }
        if (outside_unit_cube<D>(x)) continue;
        uint32_t cell[D];
        float frac[D], dfrac[D];
        locate<D>(x, scale, align_corners, interp, cell, frac, dfrac);
        const float wx[2] = {1 - frac[0], frac[0]}, wy[2] = {1 - frac[1], frac[1]}, wz[2] = {1 - frac[2], frac[2]};
        uint32_t tx[2], ty[2], tz[2];
        if (MODE == 1) {
            tx[0] = cell[0]; tx[1] = cell[0] + 1;
            ty[0] = cell[1] * kPrimes[1]; ty[1] = ty[0] + kPrimes[1];
            tz[0] = cell[2] * kPrimes[2]; tz[1] = tz[0] + kPrimes[2];
        } else if (MODE == 2) {
            tx[0] = cell[0]; tx[1] = cell[0] + 1;
            ty[0] = cell[1] * ix.stride[1]; ty[1] = ty[0] + ix.stride[1];
            tz[0] = cell[2] * ix.stride[2]; tz[1] = tz[0] + ix.stride[2];
        }
#pragma unroll
        for (uint32_t cc = 0; cc < 8; ++cc) {
            const uint32_t corner = MODE == 1 ? cc : ((cc + lane) & 7u);
            const uint32_t i = corner & 1u, j = (corner >> 1) & 1u, k = corner >> 2;
            uint32_t row;
            if (MODE == 1) row = (tx[i] ^ ty[j] ^ tz[k]) & ix.mask;
            else if (MODE == 2) row = (i ? tx[1] : tx[0]) + (j ? ty[1] : ty[0]) + (k ? tz[1] : tz[0]);
            else {
                const uint32_t v[D] = {cell[0] + i, cell[1] + j, cell[2] + k};
                row = ix.row(v);
            }
            uint32_t rel;
            if (pm.mine(row, rel)) {
                const float w = ((i ? wx[1] : wx[0]) * (j ? wy[1] : wy[0])) * (k ? wz[1] : wz[0]);   // forward's association
#pragma unroll
                for (uint32_t c = 0; c < C; ++c) {
                    __hip_atomic_fetch_add(&acc[rel * C + c], w * (float)gr.v[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }
}

// Total-variation sweep (gridencoder.cu:505-609) for one (level, partition) item: the cell of every sample is looked
// up, and when its row belongs to this partition the TV term  w * sum_nb (g - g_nb) / sqrt(sum_nb (g - g_nb)^2 + 1e-9)
// over the up-to-6 axis neighbours is added to the partition's LDS accumulator (instead of one scattered global
// atomic per sample and level).  fp32 tables only, like the reference's effective path.
template <uint32_t C>
__device__ __forceinline__ void tv_apply(float* acc, const float* __restrict__ tab, const float* __restrict__ inputs,
                                         const Indexer<3>& ix, float scale, bool align_corners, uint32_t resolution, float w,
                                         uint32_t s, const PartMap& pm) {
    constexpr uint32_t D = 3;
    float x[D];
    load_point<D>(inputs, s, x);
    uint32_t cell[D];
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) cell[d] = (uint32_t)floorf(x[d] * scale + (align_corners ? 0.0f : 0.5f));
    const uint32_t here = ix.row(cell);
    uint32_t rel = 0;
    (void)pm.mine(here, rel);
    // issue all 7 row reads before using any of them
    uint32_t nb_row[2 * D];
    bool nb_ok[2 * D];
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) {
        const uint32_t cur = cell[d];
        nb_ok[2 * d] = cur < resolution;
        cell[d] = cur + 1;
        nb_row[2 * d] = nb_ok[2 * d] ? ix.row(cell) : here;
        nb_ok[2 * d + 1] = cur > 0;
        cell[d] = cur - 1;
        nb_row[2 * d + 1] = nb_ok[2 * d + 1] ? ix.row(cell) : here;
        cell[d] = cur;
    }
    const Row<float, C> centre = Row<float, C>::load(tab + (size_t)here * C);
    Row<float, C> nb[2 * D];
#pragma unroll
    for (uint32_t k = 0; k < 2 * D; ++k) nb[k] = Row<float, C>::load(tab + (size_t)nb_row[k] * C);
    float sum[C], sq[C];
#pragma unroll
    for (uint32_t c = 0; c < C; ++c) { sum[c] = 0.f; sq[c] = 0.f; }
#pragma unroll
    for (uint32_t k = 0; k < 2 * D; ++k) {   // same order as the reference: +1 then -1 neighbour, axis by axis
        if (nb_ok[k]) {
#pragma unroll
            for (uint32_t c = 0; c < C; ++c) { const float dv = centre.v[c] - nb[k].v[c]; sum[c] += dv; sq[c] += dv * dv; }
        }
    }
#pragma unroll
    for (uint32_t c = 0; c < C; ++c)
        __hip_atomic_fetch_add(&acc[rel * C + c], w * sum[c] * (1.0f / sqrtf(sq[c] + 1e-9f)), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Total-variation sweep (gridencoder.cu:505-609) for one (level, partition) item.  Only ~1/partitions of the samples
// have their cell in this partition; evaluating the 7-row stencil under that divergence would make every wave pay the
// full stencil latency on every iteration.  Instead each wave COMPACTS the qualifying sample ids into a small LDS queue
// (ballot + popcount) and runs the stencil only on full batches of 64.
template <uint32_t C>
__device__ __forceinline__ void tv_sweep(float* acc, uint32_t* queue /*[128] per wave*/, const float* __restrict__ tab,
                                         const float* __restrict__ inputs, const Indexer<3>& ix, float scale, bool align_corners,
                                         uint32_t resolution, float w, uint32_t s_first, uint32_t run, uint32_t s_end,
                                         const PartMap pm) {
    constexpr uint32_t D = 3;
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t qn = 0;                                   // wave-uniform queue length
    for (uint32_t it = 0; it < run; ++it) {
        const uint32_t s = s_first + it * 1024u;
        bool mine = false;
        if (s < s_end) {
            float x[D];
            load_point<D>(inputs, s, x);
            if (!outside_unit_cube<D>(x)) {
                uint32_t cell[D];
#pragma unroll
                for (uint32_t d = 0; d < D; ++d) cell[d] = (uint32_t)floorf(x[d] * scale + (align_corners ? 0.0f : 0.5f));
                uint32_t rel;
                mine = pm.mine(ix.row(cell), rel);
            }
        }
        const unsigned long long m = __ballot(mine);
        if (mine) queue[qn + (uint32_t)__popcll(m & below)] = s;
        qn += (uint32_t)__popcll(m);
        if (qn >= 64u) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const uint32_t sq = queue[lane];
            const uint32_t tail = lane + 64u < qn ? queue[lane + 64u] : 0u;
            tv_apply<C>(acc, tab, inputs, ix, scale, align_corners, resolution, w, sq, pm);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (lane + 64u < qn) queue[lane] = tail;
            qn -= 64u;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane < qn) tv_apply<C>(acc, tab, inputs, ix, scale, align_corners, resolution, w, queue[lane], pm);
}

template <typename T, uint32_t C, bool TV>
__global__ void __launch_bounds__(1024)
grid_backward_lds_kernel(const T* __restrict__ grad /*[L,B,C]*/, const float* __restrict__ inputs,
                         const int32_t* __restrict__ offsets, T* __restrict__ grad_table, uint32_t B, uint32_t max_level,
                         LevelTable lv, uint32_t gridtype, bool align_corners, uint32_t interp, uint32_t G,
                         const float* __restrict__ tv_table, float tv_weight) {
    constexpr uint32_t D = 3;
    constexpr uint32_t P = kLdsBytes / (4 * C);          // table rows per partition
    extern __shared__ __attribute__((aligned(16))) float acc[];
    __shared__ uint32_t item_prefix[kMaxLevels + 1];
    const uint32_t tid = threadIdx.x;

    // LDS float atomics sustain only ~0.8 G/s per CU (measured), so an item's cost is its number of LDS updates:
    // 8 * samples / partitions(level).  A level with few partitions (the small dense tables: ONE partition receives
    // every update) therefore gets proportionally more sample groups, so that all items carry about `target` updates.
    __shared__ uint32_t level_groups[kMaxLevels];
    __shared__ uint32_t tv_queue[TV ? 16 * 128 : 1];       // per-wave compaction queues of the TV sweep
    if (tid == 0) {
        const uint32_t target = 24576u / (C > 2 ? 2 : C) * 2u;     // LDS updates per item (C float atomics each)
        uint32_t run = 0;
        for (uint32_t l = 0; l < max_level; ++l) {
            const uint32_t size = (uint32_t)(offsets[l + 1] - offsets[l]);
            const uint32_t parts = (((size + 15u) >> 4) + (P / 16u) - 1) / (P / 16u);   // partitions hold whole 16-row blocks
            const uint64_t updates = (uint64_t)(TV ? 1 : 8) * B;   // LDS updates of the whole level
            uint32_t g = (uint32_t)((updates + (uint64_t)parts * target - 1) / ((uint64_t)parts * target));
            g = g < G ? G : (g > 256u ? 256u : g);
            level_groups[l] = g;
            item_prefix[l] = run;
            run += parts * g;
        }
        item_prefix[max_level] = run;
    }
    __syncthreads();
    const uint32_t total_items = item_prefix[max_level];

    for (uint32_t item = blockIdx.x; item < total_items; item += gridDim.x) {
        uint32_t level = 0;
        while (item >= item_prefix[level + 1]) ++level;
        const uint32_t Gl = level_groups[level];
        const uint32_t chunk = (((B + Gl - 1) / Gl) + 1023u) & ~1023u;   // samples per group, a multiple of the block size
        const uint32_t run = chunk / 1024u;                              // iterations per lane
        const uint32_t local = item - item_prefix[level];
        const uint32_t part = local / Gl, grp = local - part * Gl;
        const uint32_t row0 = (uint32_t)offsets[level];
        const uint32_t size = (uint32_t)offsets[level + 1] - row0;
        const uint32_t parts = (((size + 15u) >> 4) + (P / 16u) - 1) / (P / 16u);
        const float scale = lv.scale[level];
        const Indexer<D> ix(size, lv.resolution[level], gridtype, align_corners);
        constexpr uint32_t kLog2P = 31u - __builtin_clz(P);            // P is a power of two for C in {1,2,4,8}
        const bool interleaved = !ix.hashed && parts > 1u;
        const PartMap pm(parts, part, kLog2P, interleaved);
        const uint32_t n_blocks = (size + 15u) >> 4;
        const uint32_t my_blocks = interleaved ? (part < n_blocks ? (n_blocks - part + parts - 1) / parts : 0u)
                                               : min(P / 16u, n_blocks - min(n_blocks, part * (P / 16u)));
        const uint32_t rows_here = my_blocks << 4;                      // <= P by construction

        for (uint32_t i = tid; i < rows_here * C; i += 1024) acc[i] = 0.0f;
        __syncthreads();

        const uint32_t s_begin = grp * chunk;
        const uint32_t s_end = min(B, s_begin + chunk);
        const uint32_t s_first = s_begin + tid;
        const T* __restrict__ glevel = grad + (size_t)level * B * C;
        if constexpr (TV) {
            if constexpr (sizeof(T) == 4)
                tv_sweep<C>(acc, &tv_queue[(tid >> 6) * 128u], tv_table + (size_t)row0 * C, inputs, ix, scale, align_corners,
                            lv.resolution[level], tv_weight / (float)(2 * D), s_first, run, s_end, pm);
        } else if (ix.hashed && ix.pow2)
            backward_sweep<T, C, 1>(acc, glevel, inputs, ix, scale, align_corners, interp, s_first, run, s_end, pm);
        else if (!ix.hashed && !ix.wrap)
            backward_sweep<T, C, 2>(acc, glevel, inputs, ix, scale, align_corners, interp, s_first, run, s_end, pm);
        else
            backward_sweep<T, C, 0>(acc, glevel, inputs, ix, scale, align_corners, interp, s_first, run, s_end, pm);
        __syncthreads();

        // flush: 16 consecutive lanes -> the 16 rows of one block = one 64-byte line; untouched entries are skipped
        T* __restrict__ gtab = grad_table + (size_t)row0 * C;
        for (uint32_t rel = tid; rel < rows_here; rel += 1024) {
            const uint32_t row = pm.global_row(rel);
            if (row >= size) continue;
            if constexpr (sizeof(T) == 2) {
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (uint32_t c = 0; c < C; c += 2) {
                    const float a = acc[rel * C + c], b2 = acc[rel * C + c + 1];
                    if (a != 0.f || b2 != 0.f) {
                        h2 val;
                        val.x = (_Float16)a;
                        val.y = (_Float16)b2;
                        (void)__builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) h2*)(gtab + (size_t)row * C + c), val);
                    }
                }
            } else {
#pragma unroll
                for (uint32_t c = 0; c < C; ++c) {
                    const float a = acc[rel * C + c];
                    if (a != 0.f) unsafeAtomicAdd(gtab + (size_t)row * C + c, a);
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------- binned backward / TV
// The partition kernel above visits every sample once per PARTITION of a level (16-32 times for a 2^19-row level) and
// is bound by that redundant ALU work (measured: dropping its LDS atomics altogether saves only 15 %).  The binned
// path visits every (sample, level) ONCE:
//
//   bin_fill_kernel        one workgroup per (tile of 1024 samples, level): each thread derives its sample's 8 vertex
//                          updates (partition, row-in-partition, w*g), the tile is counting-sorted by partition in LDS
//                          (ds_add_rtn_u32 slots + a scan) and written as ONE contiguous, fully coalesced segment of the
//                          update log; a small directory records where each partition's run starts inside the tile.
//   bin_accumulate_kernel  one workgroup per (level, partition[, tile group]): walks the directory, streams its runs
//                          (coalesced 8-byte entries) into an LDS accumulator and flushes it with coalesced stores.
//
// The accumulator is 64-bit FIXED POINT: LDS integer atomics run ~9x faster than float ones on gfx950 (measured 7.1 vs
// 0.78 G/s per CU, tools/lds_atomic_bench.hip), and an integer sum does not depend on the order of its terms.  The unit
// is 2^-38 of the level's largest |gradient| (found by bin_fill_kernel), so every product w*g is represented to at least
// 14 bits below fp32's own resolution of the largest term and the sum is EXACT from there on; it is rounded to the
// table's type once, at the flush.  Partitions owned by a single workgroup are flushed with plain read-modify-writes,
// which makes the result bit-reproducible run to run (the reference's atomicAdd order is not, gridencoder.cu:324-334);
// only partitions split over several tile groups (the small dense levels) end in float atomics.
// fp16 tables: each product is rounded to half as in the reference (:326), then summed exactly.
// Layout of the caller-provided workspace: [level_max u32[32]] [directory u32] [log u64], see make_bin_plan.

constexpr uint32_t kBinAccBytes = 65536;          // LDS accumulator per accumulate workgroup (two workgroups per CU)
constexpr uint32_t kTileEntries = 8192;           // 1024 threads x 8 entries per tile and level
constexpr uint32_t kMaxPartsPerLevel = 2048;      // LDS counters of bin_fill_kernel
constexpr uint32_t kBinChunk = 1u << 20;          // samples per pass over the workspace (a stage-1 frame shades ~0.65 M pixels)

struct BinPlan {
    uint32_t row0[kMaxLevels], size[kMaxLevels], parts[kMaxLevels], groups[kMaxLevels];
    uint32_t dir_base[kMaxLevels];                // first directory word of the level, laid out [tile][parts + 1]
    uint32_t item_prefix[kMaxLevels + 1];         // accumulate work items: level -> parts * groups
    uint32_t tiles, levels;
};

template <uint32_t C> struct BinGeom {
    static constexpr uint32_t P = kBinAccBytes / (8u * C);          // table rows per partition (u64 per channel)
    static constexpr uint32_t kLog2P = 31u - __builtin_clz(P);
};

// value -> fixed point.  x = v * 2^ex is an exact power-of-two scaling with |x| < 2^38; split it into a 22-bit high part
// and a 16-bit low part, both exactly representable, and recombine in 64-bit integers.
__device__ __forceinline__ long long to_fixed(float v, float scale) {
    const float x = v * scale;
    const float hi = truncf(x * (1.0f / 65536.0f));
    const float lo = x - hi * 65536.0f;
    return ((long long)(int32_t)hi << 16) + (long long)(int32_t)rintf(lo);
}

// Total-variation term of one cell (gridencoder.cu:505-609): w * sum_nb (g - g_nb) / sqrt(sum_nb (g - g_nb)^2 + 1e-9) over the
// up-to-6 axis neighbours, in the reference's order (+1 then -1 neighbour, axis by axis).  C = 1 tables.
__device__ __forceinline__ float tv_term(const float* __restrict__ tab, const Indexer<3>& ix, uint32_t (&cell)[3], uint32_t here,
                                         uint32_t resolution, float w, uint32_t stride = 1) {
    constexpr uint32_t D = 3;
    uint32_t nb_row[2 * D];
    bool nb_ok[2 * D];
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) {
        const uint32_t cur = cell[d];
        nb_ok[2 * d] = cur < resolution;
        cell[d] = cur + 1;
        nb_row[2 * d] = nb_ok[2 * d] ? ix.row(cell) : here;
        nb_ok[2 * d + 1] = cur > 0;
        cell[d] = cur - 1;
        nb_row[2 * d + 1] = nb_ok[2 * d + 1] ? ix.row(cell) : here;
        cell[d] = cur;
    }
    const float centre = tab[(size_t)here * stride];
    float nb[2 * D];
#pragma unroll
    for (uint32_t k = 0; k < 2 * D; ++k) nb[k] = tab[(size_t)nb_row[k] * stride];
    float sum = 0.f, sq = 0.f;
#pragma unroll
    for (uint32_t k = 0; k < 2 * D; ++k)
        if (nb_ok[k]) { const float dv = centre - nb[k]; sum += dv; sq += dv * dv; }
    return w * sum * (1.0f / sqrtf(sq + 1e-9f));
}

// MODE 0: backward entries; MODE 1: TV entries only (8 samples per thread); MODE 2: backward + TV folded into vertex 000's
// entry (the TV cell floor(x*scale+0.5) IS that vertex) -- fp32 C=1 tables.
template <typename T, uint32_t C, int MODE>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8)))      // <= 64 VGPRs: two workgroups per CU
bin_fill_kernel(const T* __restrict__ grad /*[L,Bstride,C], first sample of this pass*/, const float* __restrict__ inputs, TvParams tv,
                uint32_t B, uint32_t Bstride, BinPlan plan, LevelTable lv, uint32_t gridtype, bool align_corners, uint32_t interp, uint32_t* __restrict__ level_max,
                uint32_t* __restrict__ directory, uint64_t* __restrict__ log, float* __restrict__ found_inf) {
    constexpr bool TV = MODE == 1;
    constexpr uint32_t D = 3;
    extern __shared__ __attribute__((aligned(16))) uint64_t bin_stage[];       // kTileEntries entries, grouped by partition
    __shared__ uint32_t cnt[kMaxPartsPerLevel];
    __shared__ uint32_t wave_tot[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    const uint32_t tile = blockIdx.x, level = blockIdx.y;
    const uint32_t parts = plan.parts[level], size = plan.size[level];
    for (uint32_t i = tid; i < parts; i += 1024) cnt[i] = 0;
    __syncthreads();

    const float scale = lv.scale[level];
    const Indexer<D> ix(size, lv.resolution[level], gridtype, align_corners);
    const PartMap pm(parts, 0, BinGeom<C>::kLog2P, !ix.hashed && parts > 1u);
    uint32_t e_part[8], e_rel[8], e_val[8], e_slot[8];
    uint32_t vmask = 0;
    float vmax = 0.0f;        // largest finite |value source| seen by this lane

    if constexpr (!TV) {
        const uint32_t s = tile * 1024u + tid;
        float x[D] = {2.f, 2.f, 2.f};
        if (s < B) load_point<D>(inputs, s, x);
        if (!outside_unit_cube<D>(x)) {
            const Row<T, C> gr = Row<T, C>::load(grad + ((size_t)level * Bstride + s) * C);
#pragma unroll
            for (uint32_t c = 0; c < C; ++c) {
                const float a = fabsf((float)gr.v[c]);
                vmax = fmaxf(vmax, a <= 3.0e38f ? a : 1.0f);             // inf / nan: keep the level alive, they bypass the fixed point
                if (!(a <= 3.0e38f) && found_inf) *found_inf = 1.0f;     // GradScaler's non-finite check, done where the value is read anyway
            }
            uint32_t cell[D];
            float tvv = 0.0f;
            float frac[D], dfrac[D];
            locate<D>(x, scale, align_corners, interp, cell, frac, dfrac);
            const float wx[2] = {1 - frac[0], frac[0]}, wy[2] = {1 - frac[1], frac[1]}, wz[2] = {1 - frac[2], frac[2]};
            if constexpr (MODE == 2) {
                const bool inner = fmaxf(fmaxf(fabsf(x[0] - 0.5f), fabsf(x[1] - 0.5f)), fabsf(x[2] - 0.5f)) <= tv.inner01;
                float w = (inner ? tv.weight : tv.weight_outer);
                if (tv.scale_ptr) w *= *tv.scale_ptr;
                tvv = tv_term(tv.table + (size_t)plan.row0[level] * tv.stride, ix, cell, ix.row(cell), lv.resolution[level], w / (float)(2 * D), tv.stride);
                const float a = fabsf(tvv);
                vmax += a <= 3.0e38f ? a : 1.0f;                         // |w*g + tv| <= |g| + |tv|
            }
#pragma unroll
            for (uint32_t corner = 0; corner < 8; ++corner) {
                const uint32_t i = corner & 1u, j = (corner >> 1) & 1u, k = corner >> 2;
                const uint32_t v[D] = {cell[0] + i, cell[1] + j, cell[2] + k};
                const uint32_t row = ix.row(v);
                const float w = (wx[i] * wy[j]) * wz[k];                 // forward's association
                uint32_t bits;
                bool nz;
                if constexpr (sizeof(T) == 4) {
                    float p = w * (float)gr.v[0];
                    if (MODE == 2 && corner == 0) p += tvv;
                    bits = __float_as_uint(p);
                    nz = (bits << 1) != 0u;
                } else {
                    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                    h2 p;
                    p.x = half_product(w, (float)gr.v[0]);
                    p.y = half_product(w, (float)gr.v[1]);
                    bits = __builtin_bit_cast(uint32_t, p);
                    nz = (bits & 0x7FFF7FFFu) != 0u;
                }
                pm.split(row, e_part[corner], e_rel[corner]);
                e_val[corner] = bits;
                if (nz) vmask |= 1u << corner;
            }
        }
    } else {
        // total variation (gridencoder.cu:505-609): one entry per sample, 8 samples per thread
        const uint32_t resolution = lv.resolution[level];
        const float sc = tv.scale_ptr ? *tv.scale_ptr : 1.0f;
#pragma unroll
        for (uint32_t c8 = 0; c8 < 8; ++c8) {
            const uint32_t s = tile * kTileEntries + c8 * 1024u + tid;
            float x[D] = {2.f, 2.f, 2.f};
            if (s < B) load_point<D>(inputs, s, x);
            if (outside_unit_cube<D>(x)) continue;
            uint32_t cell[D];
#pragma unroll
            for (uint32_t d = 0; d < D; ++d) cell[d] = (uint32_t)floorf(x[d] * scale + (align_corners ? 0.0f : 0.5f));
            const uint32_t here = ix.row(cell);
            const bool inner = fmaxf(fmaxf(fabsf(x[0] - 0.5f), fabsf(x[1] - 0.5f)), fabsf(x[2] - 0.5f)) <= tv.inner01;
            const float w = tv.scale_ptr ? (inner ? tv.weight : tv.weight_outer) * sc : (inner ? tv.weight : tv.weight_outer);
            const float p = tv_term(tv.table + (size_t)plan.row0[level] * tv.stride, ix, cell, here, resolution, w / (float)(2 * D), tv.stride);
            const uint32_t bits = __float_as_uint(p);
            pm.split(here, e_part[c8], e_rel[c8]);
            e_val[c8] = bits;
            if ((bits << 1) != 0u) {
                vmask |= 1u << c8;
                const float a = fabsf(p);
                vmax = fmaxf(vmax, a <= 3.0e38f ? a : 1.0f);
            }
        }
    }

    // the level's largest finite magnitude -> the fixed-point unit of bin_accumulate_kernel.  Same-address global atomics
    // retire one every ~11 ns (measured: 65 k of them cost 0.75 ms), so the workgroup reduces first and only issues one
    // when it would actually raise the value (a stale read merely costs a redundant atomic).
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    if (lane == 0) wave_tot[wid] = __float_as_uint(vmax);
    __syncthreads();
    if (tid == 0) {
        uint32_t m = 0;
#pragma unroll
        for (uint32_t w = 0; w < 16; ++w) m = max(m, wave_tot[w]);
        if (m > __hip_atomic_load(&level_max[level], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&level_max[level], m);
    }

    // slot of every entry inside its partition's run of this tile
    if (parts == 1u) {
        const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
        for (uint32_t c = 0; c < 8; ++c) {
            const bool v = (vmask >> c) & 1u;
            const unsigned long long m = __ballot(v);
            uint32_t base = 0;
            if (lane == 0 && m) base = atomicAdd(&cnt[0], (uint32_t)__popcll(m));
            base = __shfl(base, 0, 64);
            e_slot[c] = base + (uint32_t)__popcll(m & below);
        }
    } else {
#pragma unroll
        for (uint32_t c = 0; c < 8; ++c)
            if ((vmask >> c) & 1u) e_slot[c] = atomicAdd(&cnt[e_part[c]], 1u);
    }
    __syncthreads();

    // exclusive scan of the counters in place (two per thread), run starts into the directory
    const uint32_t i0 = 2u * tid, i1 = i0 + 1u;
    const uint32_t a0 = i0 < parts ? cnt[i0] : 0u, a1 = i1 < parts ? cnt[i1] : 0u;
    const uint32_t incl = n2m_wave_scan_add_u32(a0 + a1, (int)lane);
    if (lane == 63u) wave_tot[wid] = incl;
    __syncthreads();
    uint32_t woff = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < 16; ++w) {
        const uint32_t t = wave_tot[w];
        if (w < wid) woff += t;
        total += t;
    }
    const uint32_t excl = woff + incl - (a0 + a1);
    uint32_t* __restrict__ dir = directory + plan.dir_base[level] + (size_t)tile * (parts + 1u);
    if (i0 < parts) { cnt[i0] = excl; dir[i0] = excl; }
    if (i1 < parts) { cnt[i1] = excl + a0; dir[i1] = excl + a0; }
    if (tid == 0) dir[parts] = total;
    __syncthreads();

#pragma unroll
    for (uint32_t c = 0; c < 8; ++c)
        if ((vmask >> c) & 1u) bin_stage[cnt[e_part[c]] + e_slot[c]] = ((uint64_t)e_rel[c] << 32) | e_val[c];
    __syncthreads();

    uint64_t* __restrict__ seg = log + ((size_t)level * plan.tiles + tile) * kTileEntries;
    for (uint32_t i = tid; i < total; i += 1024) seg[i] = bin_stage[i];
}

// Both encoders of the field in ONE fill.  nerf2mesh's density (fp32, C=1) and colour (fp16, C=2) tables share their
// geometry (levels, resolutions, hash, offsets) and are queried at the same points, so cell, weights, rows, partition sort and
// directory are computed once and two logs are written (same structure, different values).  Measured before this: the
// front end (loads, index arithmetic, sort) is ~80 % of a fill, the log traffic the rest.  Partitions hold kPairP rows for
// both tables (C=1 then uses half of its LDS accumulator).  TV (template flag) rides on vertex 000 of the fp32 log.
constexpr uint32_t kPairP = kBinAccBytes / 16u;      // 4096 rows
constexpr uint32_t kPairTilesPerWg = 4;              // tiles one fill workgroup walks (next tile's inputs prefetched)
constexpr uint32_t kPairMergeLevels = 9;             // levels (from the coarsest) whose same-cell runs of consecutive samples are merged

// Entries of one sample in one level for bin_fill_pair_kernel.  IMODE 1: hashed power-of-two table, 2: dense table without wrap,
// 0: generic Indexer::row.  ILV: interleaved partition map (dense levels spanning several partitions); IMODE 0 asks the PartMap.
// In the two fast modes the per-axis terms of the row index are hoisted, and the six TV neighbours reuse them: +1 neighbours ARE
// corners 100 / 010 / 001, -1 neighbours cost one subtraction each (was: seven generic Indexer::row calls per sample and level).
struct PairCtx {
    TvParams tv; const float* tv_tab; float scale; uint32_t resolution; bool align_corners; uint32_t interp;
    float tv_w_in, tv_w_out;      // the TV weight of an inner / outer sample: (weight [* *scale_ptr]) / 6, the reference's operations once per kernel
};
// (w = inner ? weight : weight_outer; w *= *scale_ptr; w /= 2 D -- an IEEE division, ten instructions, per (sample, level) until round 4)
__device__ __forceinline__ PairCtx make_pair_ctx(const TvParams& tv, const float* tv_tab, float scale, uint32_t resolution, bool align_corners, uint32_t interp) {
    float w_in, w_out;
    tv_weights(tv, w_in, w_out);
    return PairCtx{tv, tv_tab, scale, resolution, align_corners, interp, w_in, w_out};
}

// TV term of one (sample, level): gridencoder.cu:505-609 on the cell floor(x * scale + 0.5) -- vertex 000 of the interpolation cell.  One
// function for the fill that computes it in place (TV mode 1) and for the stand-alone pre-pass n2m_grid_tv_terms (whose result the fill
// of TV mode 2 reads back): identical bits either way.
#ifndef N2M_TV_ABLATE
#define N2M_TV_ABLATE 0
#endif
template <int IMODE>
__device__ __forceinline__ float pair_tv_value(const PairCtx& cx, const Indexer<3>& ix, const float (&x)[3], uint32_t (&cell)[3],
                                               const uint32_t (&rows)[8], const uint32_t (&tx)[2], const uint32_t (&ty)[2],
                                               const uint32_t (&tz)[2], uint32_t sy, uint32_t sz, const float4* corners = nullptr) {
    constexpr uint32_t D = 3;
    auto comb = [&](uint32_t a, uint32_t b, uint32_t c) { return IMODE == 1 ? ((a ^ b ^ c) & ix.mask) : (a + b + c); };
    float tvv = 0.0f;
        const float w = tv_inner(cx.tv, x) ? cx.tv_w_in : cx.tv_w_out;
        if constexpr (IMODE == 0) tvv = tv_term(cx.tv_tab, ix, cell, rows[0], cx.resolution, w, cx.tv.stride);
        else {
            // gridencoder.cu:505-609, neighbours in the reference's order: +x -x +y -y +z -z; out-of-grid ones are skipped
            const float* __restrict__ tab = cx.tv_tab;
            const uint32_t nb_row[6] = {rows[1], comb(tx[0] - 1u, ty[0], tz[0]), rows[2], comb(tx[0], ty[0] - sy, tz[0]),
                                        rows[4], comb(tx[0], ty[0], tz[0] - sz)};
            const bool nb_ok[6] = {cell[0] < cx.resolution, cell[0] > 0u, cell[1] < cx.resolution, cell[1] > 0u,
                                   cell[2] < cx.resolution, cell[2] > 0u};
            const uint32_t st = cx.tv.stride;
            float centre, nb[6];
            if (IMODE == 1 && corners != nullptr) {
                // Round 6.  The fine levels of the fill are bound by their XCD's L2 REQUEST rate, and six of the ~8 scattered requests a (sample,
                // level) costs were this stencil (profiles/r06_fill_split.txt).  Four of its seven values are corners 000 / 100 / 010 / 001 of the
                // sample's interpolation cell, which the forward lookup has just gathered: it leaves them as ONE coalesced 16-byte record per
                // (sample, level), and only the -x / -y / -z neighbours are gathered here.  Same table, same state, same bits.
                const float4 c4 = *corners;
                centre = c4.x; nb[0] = c4.y; nb[2] = c4.z; nb[4] = c4.w;
                nb[1] = tab[(size_t)(nb_ok[1] ? nb_row[1] : rows[0]) * st];
                nb[3] = tab[(size_t)(nb_ok[3] ? nb_row[3] : rows[0]) * st];
                nb[5] = tab[(size_t)(nb_ok[5] ? nb_row[5] : rows[0]) * st];
            } else if (st == 1u) {
                // Seven scattered 4-byte reads per (sample, level) pace the fine levels (a fully divergent wave load costs the CU 64 address
                // cycles).  One of them is free: on a hashed level the x prime is 1, so the +x (cell even) or -x (cell odd) neighbour is row
                // r ^ 1 -- the other half of the centre's aligned 8 bytes; on a dense level +x is row r + 1.  Six reads instead of seven.
                typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
#if N2M_TV_ABLATE      // (measurement build, WRONG results: what the fill would gain if the forward lookup handed over the centre and the +x / +y / +z
                       //  neighbours -- corners 000 / 100 / 010 / 001 of its interpolation cell -- and only -x / -y / -z were gathered here)
                if constexpr (IMODE == 1) {
                    centre = __uint_as_float(rows[0] & 0x3F800000u);
                    nb[0] = centre; nb[2] = centre; nb[4] = centre;
                    nb[1] = tab[nb_ok[1] ? nb_row[1] : rows[0]];
                    nb[3] = tab[nb_ok[3] ? nb_row[3] : rows[0]];
                    nb[5] = tab[nb_ok[5] ? nb_row[5] : rows[0]];
                } else
#endif
                if constexpr (IMODE == 1) {
                    const float2 pr = *reinterpret_cast<const float2*>(tab + (rows[0] & ~1u));
                    const bool odd_row = (rows[0] & 1u) != 0u, even_cell = (cell[0] & 1u) == 0u;
                    centre = odd_row ? pr.y : pr.x;
                    const float other = odd_row ? pr.x : pr.y;                       // +x neighbour of an even cell, -x of an odd one
                    const uint32_t k_far = even_cell ? 1u : 0u;                      // the x neighbour the pair does not hold
                    const float far = tab[(even_cell ? nb_ok[1] : nb_ok[0]) ? (even_cell ? nb_row[1] : nb_row[0]) : rows[0]];
                    (void)k_far;
                    nb[0] = even_cell ? other : far;
                    nb[1] = even_cell ? far : other;
                } else {
                    const f2u pr = *reinterpret_cast<const f2u*>(tab + rows[0]);
                    centre = pr.x;
                    nb[0] = pr.y;
                    nb[1] = tab[nb_ok[1] ? nb_row[1] : rows[0]];
                }
#if N2M_TV_ABLATE
                if constexpr (IMODE != 1)
#endif
                {
#pragma unroll
                    for (uint32_t k = 2; k < 6; ++k) nb[k] = tab[nb_ok[k] ? nb_row[k] : rows[0]];
                }
            } else {
                centre = tab[(size_t)rows[0] * st];
#pragma unroll
                for (uint32_t k = 0; k < 6; ++k) nb[k] = tab[(size_t)(nb_ok[k] ? nb_row[k] : rows[0]) * st];
            }
            tvv = tv_reduce6(centre, nb, nb_ok, w);
        }
    return tvv;
}

template <int TV, int IMODE, bool ILV>
__device__ __forceinline__ void pair_entries(const PairCtx& cx, const Indexer<3>& ix, const PartMap& pm, const float (&x)[3], float g1,
                                             float g2x, float g2y, float a1, float& vmax1, uint32_t (&e_pr)[8], float (&f1)[8],
                                             float (&f2x)[8], float (&f2y)[8], uint32_t (&cell)[3], float tv_given = 0.0f) {
    constexpr uint32_t D = 3;
    float frac[D], dfrac[D];
    locate<D>(x, cx.scale, cx.align_corners, cx.interp, cell, frac, dfrac);
    const uint32_t sy = IMODE == 1 ? kPrimes[1] : ix.stride[1], sz = IMODE == 1 ? kPrimes[2] : ix.stride[2];
    const uint32_t tx[2] = {cell[0], cell[0] + 1u};
    const uint32_t ty0 = cell[1] * sy, tz0 = cell[2] * sz;
    const uint32_t ty[2] = {ty0, ty0 + sy}, tz[2] = {tz0, tz0 + sz};
    auto comb = [&](uint32_t a, uint32_t b, uint32_t c) { return IMODE == 1 ? ((a ^ b ^ c) & ix.mask) : (a + b + c); };
    uint32_t rows[8];
#pragma unroll
    for (uint32_t corner = 0; corner < 8; ++corner) {
        const uint32_t i = corner & 1u, j = (corner >> 1) & 1u, k = corner >> 2;
        if constexpr (IMODE != 0) rows[corner] = comb(tx[i], ty[j], tz[k]);
        else {
            const uint32_t v[D] = {cell[0] + i, cell[1] + j, cell[2] + k};
            rows[corner] = ix.row(v);
        }
    }
    float tvv = 0.0f;
    if constexpr (TV != 0) {
        if constexpr (TV == 1) tvv = pair_tv_value<IMODE>(cx, ix, x, cell, rows, tx, ty, tz, sy, sz);
        else tvv = tv_given;                       // precomputed by n2m_grid_tv_terms (same function, same bits)
        const float a = fabsf(tvv);
        vmax1 = fmaxf(vmax1, (a1 <= 3.0e38f ? a1 : 1.0f) + (a <= 3.0e38f ? a : 1.0f));       // |w*g + tv| <= |g| + |tv|
    }
    const float wx[2] = {1 - frac[0], frac[0]}, wy[2] = {1 - frac[1], frac[1]}, wz[2] = {1 - frac[2], frac[2]};
#pragma unroll
    for (uint32_t corner = 0; corner < 8; ++corner) {
        const uint32_t i = corner & 1u, j = (corner >> 1) & 1u, k = corner >> 2;
        const float w = (wx[i] * wy[j]) * wz[k];                     // forward's association
        float p1 = w * g1;
        if (TV != 0 && corner == 0) p1 += tvv;
        f1[corner] = p1;
        f2x[corner] = (float)half_product(w, g2x);                   // each product rounded to half like the reference (:326)
        f2y[corner] = (float)half_product(w, g2y);
        uint32_t part_, rel_;
        const uint32_t row = rows[corner];
        if constexpr (IMODE == 0) pm.split(row, part_, rel_);
        else if constexpr (ILV) {
            const uint32_t blk = row >> 4;
            const uint32_t q = __umulhi(blk, pm.magic);
            part_ = blk - q * pm.parts;
            rel_ = (q << 4) | (row & 15u);
        } else {
            part_ = row >> pm.log2p;
            rel_ = row & ((1u << pm.log2p) - 1u);
        }
        e_pr[corner] = (part_ << 16) | rel_;
    }
}

// Samples that follow each other along a ray fall into the SAME cell of a coarse level (a 2^18-sample batch of the lego scene:
// ~16 per cell at level 0, still ~2 at level 8), i.e. they update the same eight rows.  Lanes are consecutive samples, so such a
// run of lanes is reduced to ONE entry per vertex before anything is sorted, logged or accumulated: a segmented inclusive scan
// over the 16-lane DPP rows (row_shr 1, 2, 4, 8), the last lane of a run keeps the sums.  Halves the update log of the whole
// batch.  The partial sums are fp32 (table 2: of the half-rounded products), rounded once more when the entry is packed.
__device__ __forceinline__ float dpp_row_shr(float v, int d) {      // value of lane - d inside the 16-lane row, 0 beyond its start
    int r;
    const int i = __float_as_int(v);
    switch (d) {
        case 1: r = __builtin_amdgcn_update_dpp(0, i, 0x111, 0xF, 0xF, true); break;
        case 2: r = __builtin_amdgcn_update_dpp(0, i, 0x112, 0xF, 0xF, true); break;
        case 4: r = __builtin_amdgcn_update_dpp(0, i, 0x114, 0xF, 0xF, true); break;
        default: r = __builtin_amdgcn_update_dpp(0, i, 0x118, 0xF, 0xF, true); break;
    }
    return __int_as_float(r);
}
// returns false for lanes whose entries were handed to a later lane of their run
__device__ __forceinline__ bool merge_runs(bool inside, const uint32_t (&cell)[3], float (&f1)[8], float (&f2x)[8], float (&f2y)[8], uint32_t lane) {
    const uint32_t key0 = inside ? cell[0] : 0xFFFFFFFFu;           // an outside lane never matches (and carries zeros)
    const uint32_t p0 = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)key0, 0x111, 0xF, 0xF, false);
    const uint32_t p1 = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)cell[1], 0x111, 0xF, 0xF, false);
    const uint32_t p2 = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)cell[2], 0x111, 0xF, 0xF, false);
    const bool same = inside && (lane & 15u) != 0u && p0 == key0 && p1 == cell[1] && p2 == cell[2];
    const unsigned long long heads = __ballot(!same);               // bit i: lane i starts a run
    const unsigned long long upto = heads & (~0ull >> (63u - lane));
    const uint32_t dist = lane - (63u - (uint32_t)__builtin_clzll(upto));   // lanes between this one and the head of its run
    const bool last = lane == 63u || ((heads >> (lane + 1u)) & 1ull);
    if (__builtin_popcountll(heads) == 64) return true;              // no run anywhere in this wave (wave-uniform)
#pragma unroll
    for (int d = 1; d <= 8; d <<= 1) {
        const bool take = dist >= (uint32_t)d;
        if (__ballot(take) == 0ull) break;                           // no run of this wave is longer than d lanes: the later steps add nothing
#pragma unroll
        for (uint32_t c = 0; c < 8; ++c) {
            const float a = dpp_row_shr(f1[c], d), b = dpp_row_shr(f2x[c], d), e = dpp_row_shr(f2y[c], d);
            if (take) { f1[c] += a; f2x[c] += b; f2y[c] += e; }
        }
    }
    return last;
}

// The TV terms of a batch on their own (n2m_grid_tv_terms): tv_out[level, s] = pair_tv_value of (sample s, level) -- what the shared fill adds
// to vertex 000's entry.  Reads the samples and the density table only, i.e. nothing the step produces after its forward lookup: the step
// executor runs it on a side stream beside the field kernels (MFMA / latency bound, they leave the L2 request path idle) and hands the fill
// the finished terms.  Grid (tiles of 256 samples, level).
__global__ void __launch_bounds__(256)
tv_terms_kernel(const float* __restrict__ inputs, TvParams tv, uint32_t B, uint32_t Bstride, BinPlan plan, LevelTable lv, uint32_t gridtype,
                bool align_corners, uint32_t interp, float in_scale, float in_offset, float* __restrict__ tv_out) {
    constexpr uint32_t D = 3;
    const uint32_t level = blockIdx.y, s = blockIdx.x * 256u + threadIdx.x;
    if (s >= B) return;
    float x[D];
    load_point<D>(inputs, s, x);
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) x[d] = x[d] * in_scale + in_offset;
    float tvv = 0.0f;
    if (!outside_unit_cube<D>(x)) {
        const uint32_t size = plan.size[level];
        const Indexer<D> ix(size, lv.resolution[level], gridtype, align_corners);
        const PairCtx cx = make_pair_ctx(tv, tv.table + (size_t)plan.row0[level] * tv.stride, lv.scale[level], lv.resolution[level], align_corners, interp);
        const bool fast_hash = ix.hashed && ix.pow2, fast_dense = !ix.hashed && !ix.wrap;
        uint32_t cell[D];
        float frac[D], dfrac[D];
        locate<D>(x, cx.scale, align_corners, interp, cell, frac, dfrac);
        auto run = [&](auto mode) {
            constexpr int IMODE = decltype(mode)::value;
            const uint32_t sy = IMODE == 1 ? kPrimes[1] : ix.stride[1], sz = IMODE == 1 ? kPrimes[2] : ix.stride[2];
            const uint32_t tx[2] = {cell[0], cell[0] + 1u};
            const uint32_t ty0 = cell[1] * sy, tz0 = cell[2] * sz;
            const uint32_t ty[2] = {ty0, ty0 + sy}, tz[2] = {tz0, tz0 + sz};
            auto comb = [&](uint32_t a, uint32_t b, uint32_t c) { return IMODE == 1 ? ((a ^ b ^ c) & ix.mask) : (a + b + c); };
            uint32_t rows[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
            if constexpr (IMODE != 0) { rows[0] = comb(tx[0], ty[0], tz[0]); rows[1] = comb(tx[1], ty[0], tz[0]); rows[2] = comb(tx[0], ty[1], tz[0]); rows[4] = comb(tx[0], ty[0], tz[1]); }
            else { const uint32_t v[D] = {cell[0], cell[1], cell[2]}; rows[0] = ix.row(v); }
            return pair_tv_value<IMODE>(cx, ix, x, cell, rows, tx, ty, tz, sy, sz);
        };
        if (fast_hash) tvv = run(std::integral_constant<int, 1>{});
        else if (fast_dense) tvv = run(std::integral_constant<int, 2>{});
        else tvv = run(std::integral_constant<int, 0>{});
    }
    tv_out[(size_t)level * Bstride + s] = tvv;
}

// Measurement aid (n2m_debug_fill_times): shader-clock stamps of two workgroups' first 8 tile iterations, 6 per iteration --
// loop top, entries ready, after barrier 1 (slots counted), after barrier 3 (run starts), after barrier 4 (tile staged), log stores issued.
__device__ unsigned int g_fill_timing_on;
__device__ unsigned long long g_fill_t[2][8][6];     // workgroup 3 (a fine, hashed level) and workgroup gridDim/2 + 3 (a coarse, dense one)
#define N2M_FILL_STAMP(i) do { if (stamp && it < 8u) g_fill_t[stamp_w][it][(i)] = __builtin_readcyclecounter(); } while (0)

// ---- SDF recipe: the six finite-difference copies of a sample (x +- eps e_a, nerf/network.py:143-154) folded into the sample's own entries.
// At the end of the schedule eps = 1e-4 is a tenth of the finest cell: on a given level a copy nearly always lies in the cell of its centre
// sample (99.9 % on level 0 ... 90 % on level 15), i.e. it updates the SAME eight rows with slightly different weights.  Instead of going
// through the fill as six more samples (the stacked pass: ~800 us per step), such a (copy, level) pair adds w_copy(corner) * g_copy to the
// centre's eight values here; the pairs that leave the cell (n2m_sdf_fold_plan lists them per level, 2.4 % of all) take a density-only call
// over those lists.  Positions are recomputed exactly as n2m_sdf_offsets writes them (clamp(x + off, -bound, bound), then
// (p + bound) / (2 bound)), so a folded copy's weights are the ones the stacked pass would have used, bit for bit; what changes is the
// association of the fp32 sum (centre + copies here, a lane scan there).
struct FoldArgs {
    const uint8_t* flags;       // [L, B] bit c: copy c (= 2 axis + (0: +eps, 1: -eps)) shares the centre's cell on this level
    const float* grad6;         // [L, 6 B] gradient of the stacked copies' features, sample-major inside a level
    const float* raw;           // [B, 3] the samples as the caller has them (before the in_scale / in_offset map)
    uint32_t stride6;           // 6 B
    float eps, bound;
};

__device__ __forceinline__ void fold_copies(const FoldArgs& fo, uint32_t s, uint32_t level, const float (&x01)[3], float scale, bool align_corners,
                                            uint32_t interp, uint32_t fl, float (&f1)[8], float& vmax1, float* found_inf) {
    uint32_t cell[3];
    float frac[3], dfrac[3];
    locate<3>(x01, scale, align_corners, interp, cell, frac, dfrac);
    const float2* gp = reinterpret_cast<const float2*>(fo.grad6 + (size_t)level * fo.stride6 + (size_t)s * 6u);
    const float2 g01 = gp[0], g23 = gp[1], g45 = gp[2];
    const float g6[6] = {g01.x, g01.y, g23.x, g23.y, g45.x, g45.y};
    const float xw[3] = {fo.raw[(size_t)s * 3u], fo.raw[(size_t)s * 3u + 1u], fo.raw[(size_t)s * 3u + 2u]};
#pragma unroll
    for (uint32_t c = 0; c < 6; ++c) {
        if (!((fl >> c) & 1u)) continue;
        const uint32_t a = c >> 1;
        const float g = g6[c];
        if (!(fabsf(g) <= 3.0e38f)) { if (found_inf) *found_inf = 1.0f; continue; }
        const float pw = fminf(fmaxf(xw[a] + ((c & 1u) ? -fo.eps : fo.eps), -fo.bound), fo.bound);
        const float p01 = (pw + fo.bound) / (2.0f * fo.bound);
        float fr = p01 * scale + (align_corners ? 0.0f : 0.5f);
        fr -= (float)cell[a];                                           // (same cell: the plan has checked floor() of this very value)
        if (interp == 1) fr = fr * fr * (3.0f - 2.0f * fr);
        float w2[3][2];
#pragma unroll
        for (uint32_t d = 0; d < 3; ++d) { const float f = d == a ? fr : frac[d]; w2[d][0] = 1 - f; w2[d][1] = f; }
#pragma unroll
        for (uint32_t corner = 0; corner < 8; ++corner) {
            const float w = (w2[0][corner & 1u] * w2[1][(corner >> 1) & 1u]) * w2[2][corner >> 2];
            f1[corner] += w * g;
        }
    }
    float m = 0.0f;
#pragma unroll
    for (uint32_t corner = 0; corner < 8; ++corner) m = fmaxf(m, fabsf(f1[corner]));
    vmax1 = fmaxf(vmax1, m <= 3.0e38f ? m : 1.0f);
    if (!(m <= 3.0e38f) && found_inf) *found_inf = 1.0f;
}

// TV: 0 none, 1 computed in place (pair_tv_value), 2 read from `tv_terms` [L, Bstride] (n2m_grid_tv_terms wrote it earlier, beside the
// field kernels: the stencil's scattered gathers -- 45 us of this kernel -- then run where nobody waits for the L2 request path)
template <int TV, bool FOLD = false>
__global__ void __launch_bounds__(1024)
bin_fill_pair_kernel(const float* __restrict__ grad1 /*[L,Bstride]*/, const _Float16* __restrict__ grad2 /*[L,Bstride,2]*/,
                     const float* __restrict__ inputs, TvParams tv, const float* __restrict__ tv_terms, uint32_t B, uint32_t Bstride, BinPlan plan, LevelTable lv,
                     uint32_t gridtype, bool align_corners, uint32_t interp, uint32_t* __restrict__ level_max /*[2][32]*/,
                     uint32_t* __restrict__ directory, uint16_t* __restrict__ log_rel, uint32_t* __restrict__ log_v1,
                     uint32_t* __restrict__ log_v2, float* __restrict__ found_inf, float in_scale, float in_offset,
                     float* __restrict__ clear1, _Float16* __restrict__ clear2, uint32_t clear_mask1, uint32_t clear_mask2,
                     uint32_t merge_levels, uint32_t groups_x, uint32_t slot_begin, unsigned long long* __restrict__ lm_ready,
                     unsigned long long lm_token, uint32_t in_level_stride = 0u /* floats between the levels' own point lists; 0: one list */,
                     FoldArgs fold = FoldArgs{}) {
    __builtin_amdgcn_s_setprio(3);
    constexpr uint32_t D = 3;
    constexpr uint32_t kLog2P = 31u - __builtin_clz(kPairP);
    // The level maxima start from zero.  Workgroup 0 clears them itself and then publishes this launch's token; every workgroup waits for
    // the token before its ONE atomicMax at the very end (a whole tile walk later).  That removes a 256-byte memset launch from the
    // stream -- a kernel boundary costs 6-7 us here whatever the kernel does (timeline).
    // (ONE workgroup: on the plain 2-D grid of a call with fewer than 16 levels blockIdx.x == 0 names one workgroup PER LEVEL -- each of them
    // used to clear the maxima again, whenever it happened to start, wiping out what faster workgroups had already published: levels then
    // reached the accumulate kernels with a maximum of zero (skipped) or too small a unit.  Found by the folded-copies test against the
    // scatter backward; max_level == 16 runs the 1-D XCD grid and was never affected.)
    if (blockIdx.x == 0u && blockIdx.y == 0u) {
        if (threadIdx.x < 2u * kMaxLevels) level_max[threadIdx.x] = 0u;
        __syncthreads();
        if (threadIdx.x == 0u) __hip_atomic_store(lm_ready, lm_token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    // The logs are structure-of-arrays: row-in-partition as u16 (shared by both tables) + one 4-byte value array per table =
    // 10 bytes per update pair instead of 2 x 8; staged through LDS so that every array is written as one contiguous run.
    // SQ counters showed the waves of this kernel parked on barriers / s_waitcnt 55 % of the time (one workgroup per CU, seven
    // barriers per tile): a workgroup therefore walks several tiles of its level, requests the next tile's inputs before it
    // works on the current one, stages all three arrays at once and double-buffers the partition counters -- four barriers
    // per tile.
    extern __shared__ __attribute__((aligned(16))) uint64_t bin_stage[];
    uint32_t* stage_v1 = reinterpret_cast<uint32_t*>(bin_stage);                      // kTileEntries x u32
    uint32_t* stage_v2 = stage_v1 + kTileEntries;                                    // kTileEntries x u32
    uint16_t* stage_rel = reinterpret_cast<uint16_t*>(stage_v2 + kTileEntries);       // kTileEntries x u16
    __shared__ uint32_t cnt2[2][kMaxPartsPerLevel];
    __shared__ uint32_t wave_tot[16], wave_max[2][16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    // (tile group, level) of this workgroup.  Plain: blockIdx = (group, level).  XCD-aware (groups_x != 0, 1-D grid): workgroups are
    // dealt round-robin to the 8 XCDs; XCD x works through the levels x, 15-x, ... one after the other (coarse paired with fine:
    // equal work per XCD), so the table a level's TV stencil reads from (2 MB) stays in that XCD's 4 MB L2.
    uint32_t level = blockIdx.y, group = blockIdx.x, n_groups = gridDim.x;
    if (groups_x != 0u) {
        const uint32_t xcd = blockIdx.x & 7u, k = blockIdx.x >> 3;
        const uint32_t ls = k / groups_x;
        const uint32_t slot = slot_begin + ls;                       // which of this XCD's levels (a launch may cover a range of slots)
        group = k - ls * groups_x;
        n_groups = groups_x;
        const uint32_t pairi = xcd + 8u * (slot >> 1);               // levels come in pairs (p, L-1-p)
        level = (slot & 1u) ? pairi : plan.levels - 1u - pairi;
        if (level >= plan.levels || pairi > plan.levels - 1u - pairi || ((slot & 1u) && pairi == plan.levels - 1u - pairi)) return;
    }
    const uint32_t parts = plan.parts[level], size = plan.size[level];
    for (uint32_t i = tid; i < parts; i += 1024) { cnt2[0][i] = 0; cnt2[1][i] = 0; }
    // overwrite mode: levels whose partitions are split over several accumulate groups receive atomics and must start from zero;
    // this kernel is ordered before the accumulates, so its workgroups clear them (a slice each) instead of two extra memset launches
    if ((clear_mask1 | clear_mask2) >> level & 1u) {
        const uint32_t per = (size + n_groups - 1) / n_groups, lo = min(size, group * per), hi = min(size, lo + per);
        const size_t r0 = plan.row0[level];
        if (clear_mask1 >> level & 1u)
            for (uint32_t i = lo + tid; i < hi; i += 1024) clear1[r0 + i] = 0.0f;
        if (clear_mask2 >> level & 1u)
            for (uint32_t i = lo + tid; i < hi; i += 1024) reinterpret_cast<uint32_t*>(clear2)[r0 + i] = 0u;
    }

    const float scale = lv.scale[level];
    const Indexer<D> ix(size, lv.resolution[level], gridtype, align_corners);
    const PartMap pm(parts, 0, kLog2P, !ix.hashed && parts > 1u);
    const bool fast_hash = ix.hashed && ix.pow2, fast_dense = !ix.hashed && !ix.wrap;
    float vmax1 = 0.0f, vmax2 = 0.0f;

    // inputs of the first tile
    uint32_t tile = group;
    float nx[D] = {2.f, 2.f, 2.f}, ng1 = 0.0f, ntv = 0.0f;
    h2 ng2 = {(_Float16)0, (_Float16)0};
    auto request = [&](uint32_t t) {
        const uint32_t s = t * 1024u + tid;
        nx[0] = nx[1] = nx[2] = 2.f;
        if (t < plan.tiles && s < B) {
            load_point<D>(inputs + (size_t)level * in_level_stride, s, nx);
#pragma unroll
            for (uint32_t d = 0; d < D; ++d) nx[d] = nx[d] * in_scale + in_offset;
            ng1 = grad1 ? grad1[(size_t)level * Bstride + s] : 0.0f;        // grad1 == NULL: colour table only (stage 1: the density branch is idle)
            if (grad2) ng2 = *reinterpret_cast<const h2*>(grad2 + ((size_t)level * Bstride + s) * 2);      // grad2 == NULL: density table only
            else ng2 = h2{(_Float16)0, (_Float16)0};
            if (TV == 2) ntv = tv_terms[(size_t)level * Bstride + s];
        }
    };
    request(tile);
    __syncthreads();

    const bool stamp = (g_fill_timing_on & 1u) != 0u && tid == 0u && (blockIdx.x == 3u || blockIdx.x == gridDim.x / 2u + 3u);
    const uint32_t stamp_w = blockIdx.x == 3u ? 0u : 1u;
    for (uint32_t it = 0; tile < plan.tiles; tile += n_groups, ++it) {
        N2M_FILL_STAMP(0);
        uint32_t* cnt = cnt2[it & 1u];
        uint32_t* cnt_next = cnt2[(it & 1u) ^ 1u];
        float x[D] = {nx[0], nx[1], nx[2]};
        const float g1 = ng1, tvg = ntv;
        const h2 g2 = ng2;
        request(tile + n_groups);                        // next tile's inputs are in flight while this one is processed

        uint32_t e_pr[8], e_v1[8], e_v2[8], e_slot[8];    // e_pr = partition << 16 | row in partition (rel < 4096, parts <= 2048)
        float f1[8], f2x[8], f2y[8];
        uint32_t cell[D] = {0u, 0u, 0u};
        uint32_t vmask = 0;
        const bool inside = !outside_unit_cube<D>(x);
#pragma unroll
        for (uint32_t c = 0; c < 8; ++c) { f1[c] = 0.f; f2x[c] = 0.f; f2y[c] = 0.f; e_pr[c] = 0u; }
        if (inside) {
            const float g2x = (float)g2.x, g2y = (float)g2.y;
            const float a1 = fabsf(g1), a2 = fmaxf(fabsf(g2x), fabsf(g2y));
            vmax1 = fmaxf(vmax1, a1 <= 3.0e38f ? a1 : 1.0f);
            vmax2 = fmaxf(vmax2, a2 <= 3.0e38f ? a2 : 1.0f);           // nan: fmaxf drops it, the check below catches it
            const bool bad2 = !(fabsf(g2x) <= 3.0e38f) || !(fabsf(g2y) <= 3.0e38f);
            if ((!(a1 <= 3.0e38f) || bad2) && found_inf) *found_inf = 1.0f;
            if (bad2) vmax2 = fmaxf(vmax2, 1.0f);
            const PairCtx cx = make_pair_ctx(tv, tv.table ? tv.table + (size_t)plan.row0[level] * tv.stride : nullptr, scale, lv.resolution[level], align_corners, interp);
            // one straight-line body per index mode (wave-uniform per level) instead of three-way branches around every row
            if (fast_hash) pair_entries<TV, 1, false>(cx, ix, pm, x, g1, g2x, g2y, a1, vmax1, e_pr, f1, f2x, f2y, cell, tvg);
            else if (fast_dense && parts > 1u) pair_entries<TV, 2, true>(cx, ix, pm, x, g1, g2x, g2y, a1, vmax1, e_pr, f1, f2x, f2y, cell, tvg);
            else if (fast_dense) pair_entries<TV, 2, false>(cx, ix, pm, x, g1, g2x, g2y, a1, vmax1, e_pr, f1, f2x, f2y, cell, tvg);
            else pair_entries<TV, 0, false>(cx, ix, pm, x, g1, g2x, g2y, a1, vmax1, e_pr, f1, f2x, f2y, cell, tvg);
            if constexpr (FOLD) {
                const uint32_t sidx = tile * 1024u + tid;
                const uint32_t fl = fold.flags[(size_t)level * B + sidx];
                if (fl != 0u) fold_copies(fold, sidx, level, x, scale, align_corners, interp, fl, f1, vmax1, found_inf);
            }
        }
        bool keep = inside;
        if (level < merge_levels) {                                    // block-uniform
            keep = merge_runs(inside, cell, f1, f2x, f2y, lane) && inside;
            if (keep) {                                                  // a run's sum can exceed every one of its terms
                float m1 = 0.f, m2 = 0.f;
#pragma unroll
                for (uint32_t c = 0; c < 8; ++c) { m1 = fmaxf(m1, fabsf(f1[c])); m2 = fmaxf(m2, fmaxf(fabsf(f2x[c]), fabsf(f2y[c]))); }
                vmax1 = fmaxf(vmax1, m1 <= 3.0e38f ? m1 : 1.0f);
                vmax2 = fmaxf(vmax2, m2 <= 65504.0f ? m2 : 65504.0f);
                // a sum of finite terms that leaves the range of its type: the reference's half atomicAdd overflows on it just the same
                if ((!(m1 <= 3.0e38f) || !(m2 <= 65504.0f)) && found_inf) *found_inf = 1.0f;
            }
        }
#pragma unroll
        for (uint32_t c = 0; c < 8; ++c) {
            h2 p2;
            p2.x = (_Float16)f2x[c];
            p2.y = (_Float16)f2y[c];
            e_v1[c] = __float_as_uint(f1[c]);
            e_v2[c] = __builtin_bit_cast(uint32_t, p2);
            if (keep && ((e_v1[c] << 1) | (e_v2[c] & 0x7FFF7FFFu)) != 0u) vmask |= 1u << c;
        }

        N2M_FILL_STAMP(1);
        // slot of every entry inside its partition's run of this tile
        if (parts == 1u) {
            const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
            for (uint32_t c = 0; c < 8; ++c) {
                const bool v = (vmask >> c) & 1u;
                const unsigned long long m = __ballot(v);
                uint32_t base = 0;
                if (lane == 0 && m) base = atomicAdd(&cnt[0], (uint32_t)__popcll(m));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                e_slot[c] = base + (uint32_t)__popcll(m & below);
            }
        } else {
#pragma unroll
            for (uint32_t c = 0; c < 8; ++c)
                if ((vmask >> c) & 1u) e_slot[c] = atomicAdd(&cnt[e_pr[c] >> 16], 1u);
        }
        __syncthreads();                                                     // (1) counters complete
        N2M_FILL_STAMP(2);

        const uint32_t i0 = 2u * tid, i1 = i0 + 1u;
        const uint32_t a0 = i0 < parts ? cnt[i0] : 0u, a1c = i1 < parts ? cnt[i1] : 0u;
        const uint32_t incl = n2m_wave_scan_add_u32(a0 + a1c, (int)lane);
        if (lane == 63u) wave_tot[wid] = incl;
        __syncthreads();                                                     // (2) wave totals
        uint32_t woff = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < 16; ++w) {
            const uint32_t t = wave_tot[w];
            if (w < wid) woff += t;
            total += t;
        }
        const uint32_t excl = woff + incl - (a0 + a1c);
        uint32_t* __restrict__ dir = directory + plan.dir_base[level] + (size_t)tile * (parts + 1u);
        if (i0 < parts) { cnt[i0] = excl; dir[i0] = excl; }
        if (i1 < parts) { cnt[i1] = excl + a0; dir[i1] = excl + a0; }
        if (tid == 0) dir[parts] = total;
        __syncthreads();                                                     // (3) run starts in place
        N2M_FILL_STAMP(3);

#pragma unroll
        for (uint32_t c = 0; c < 8; ++c)
            if ((vmask >> c) & 1u) {
                const uint32_t pos = cnt[e_pr[c] >> 16] + e_slot[c];         // position in the sorted tile
                if (log_v1) stage_v1[pos] = e_v1[c];
                if (log_v2) stage_v2[pos] = e_v2[c];
                stage_rel[pos] = (uint16_t)(e_pr[c] & 0xFFFFu);
            }
        for (uint32_t i = tid; i < parts; i += 1024) cnt_next[i] = 0;        // the other counter set, for the next tile
        __syncthreads();                                                     // (4) tile staged (and next counters clean)
        N2M_FILL_STAMP(4);

        const size_t seg = ((size_t)level * plan.tiles + tile) * kTileEntries;
        // streaming stores: the log is read back by another kernel, it need not displace the level's table lines in L2
        if (g_fill_timing_on & 2u) {                 // measurement switch: plain (cache-allocating) log stores
            for (uint32_t i = tid; i < total; i += 1024) { if (log_v1) log_v1[seg + i] = stage_v1[i]; if (log_v2) log_v2[seg + i] = stage_v2[i]; }
            const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(stage_rel);
            uint32_t* __restrict__ dst = reinterpret_cast<uint32_t*>(log_rel + seg);
            for (uint32_t i = tid; i < (total + 1u) / 2u; i += 1024) dst[i] = src[i];
        } else {
        for (uint32_t i = tid; i < total; i += 1024) {
            if (log_v1) __builtin_nontemporal_store(stage_v1[i], &log_v1[seg + i]);
            if (log_v2) __builtin_nontemporal_store(stage_v2[i], &log_v2[seg + i]);
        }
        {   // rows as u32 pairs (seg is even; a trailing odd entry drags one stale u16 along: never read back)
            const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(stage_rel);
            uint32_t* __restrict__ dst = reinterpret_cast<uint32_t*>(log_rel + seg);
            for (uint32_t i = tid; i < (total + 1u) / 2u; i += 1024) __builtin_nontemporal_store(src[i], &dst[i]);
        }
        }
        // no barrier here: the next tile writes the stage only after its barrier (3), which every thread reaches after this copy
        N2M_FILL_STAMP(5);
    }

    // level maxima of both tables: workgroup reduction, one conditional atomicMax each (see bin_fill_kernel)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        vmax1 = fmaxf(vmax1, __shfl_xor(vmax1, o, 64));
        vmax2 = fmaxf(vmax2, __shfl_xor(vmax2, o, 64));
    }
    if (lane == 0) { wave_max[0][wid] = __float_as_uint(vmax1); wave_max[1][wid] = __float_as_uint(vmax2); }
    __syncthreads();
    if (tid < 2u) {
        uint32_t m = 0;
#pragma unroll
        for (uint32_t w = 0; w < 16; ++w) m = max(m, wave_max[tid][w]);
        uint32_t* dst = level_max + tid * kMaxLevels + level;
        while (__hip_atomic_load(lm_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != lm_token) __builtin_amdgcn_s_sleep(8);
        if (m > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, m);
    }
}

// Adam inside the accumulate's flush (n2m_grid_encode_backward_binned_pair_adam): on the levels whose partitions ONE work item owns (every
// hashed level of a 2^18-sample batch) the flush has the final gradient row in registers -- instead of storing it for an optimizer pass that
// reads it back together with parameter and moments, the item performs that pass for its rows: reads p / m / v from the LIVE buffers, writes
// the updated ones to the OTHER buffers (double-buffered: GradScaler's skip is all-or-nothing and the verdict is only known when the last
// item has flushed, so the update is made beside the old state and a restore pass copies the old state over it on a skipped step) and
// refreshes its column of the packed table the lookup reads.  Arithmetic = adam_kernel's (step_helpers.hip), element for element.
#ifndef N2M_FUSE_BATCH
#define N2M_FUSE_BATCH 2      // (measured: 2 -> backward 401 us, 4 -> 433 us, 8 -> 472 us: more loads in flight only add spills)
#endif
struct AdamFuse {
    const float* p_in; const float* m_in; const float* v_in;      // live buffers (whole tables: element row * C + c)
    float* p_out; float* m_out; float* v_out;                    // the other buffers
    uint32_t* packed;                                            // packed table, 8-byte rows {fp32 density, half2 colour}
    uint32_t first_level;                                        // first fused level
    float lr, beta1, beta2, omb1, omb2, eps;
    const float* scale; const float* bias; uint32_t slot;
};

// P = table rows per partition; SUB = consecutive partitions one work item accumulates (their runs are adjacent in the
// partition-sorted tiles, so they stream as one run): LDS accumulator SUB * P * C * 8 bytes.  SUB = 2 lets the fp32 table, whose
// rows are half as wide, keep 8192-row items on the 4096-row partition structure it shares with the fp16 table.
// SOA: entries come as (u16 row-in-partition, u32 value) arrays (the shared-fill logs) instead of packed u64.
// Measurement aid (n2m_debug_acc_times): shader-clock stamps of one work item per accumulate kernel (workgroup gridDim/2 + 1 and
// workgroup 1): item start, accumulator cleared + directory in (barrier), runs walked, barrier, rows flushed.
__device__ unsigned long long g_acc_t[2][2][5];
#define N2M_ACC_STAMP(i) do { if (stamp) g_acc_t[sizeof(T) == 4 ? 0 : 1][stamp_w][(i)] = __builtin_readcyclecounter(); } while (0)

template <typename T, uint32_t C, uint32_t P, uint32_t SUB, bool SOA = false, bool FUSE = false>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8)))
bin_accumulate_kernel(T* __restrict__ grad_table, BinPlan plan, LevelTable lv, uint32_t gridtype, bool align_corners,
                      const uint32_t* __restrict__ level_max, const uint32_t* __restrict__ directory,
                      const uint64_t* __restrict__ log, float* __restrict__ found_inf, const uint16_t* __restrict__ log_rel = nullptr,
                      const uint32_t* __restrict__ log_val = nullptr, bool overwrite = false, uint32_t dbg = 0, float inf_bound = 0.0f,
                      AdamFuse af = AdamFuse{}) {
    // overwrite: the gradient table holds no earlier sums.  Partitions owned by one workgroup (Gl == 1) are then STORED in full,
    // zeros included -- no read-modify-write round trips in the flush (measured: eight dependent load-add-store steps per item were
    // a third of this kernel) and no zero-fill of the table before the call; levels split over several groups still add
    // atomically onto rows the launcher has cleared.
    constexpr uint32_t kLog2P = 31u - __builtin_clz(P);
    extern __shared__ __attribute__((aligned(16))) unsigned long long bin_acc[];   // P * C
    __shared__ uint32_t nonfinite_seen;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    const uint32_t total_items = plan.item_prefix[plan.levels];

    const bool stamp = (g_fill_timing_on & 1u) != 0u && tid == 0u && (blockIdx.x == 1u || blockIdx.x == gridDim.x / 2u + 1u);
    const uint32_t stamp_w = blockIdx.x == 1u ? 0u : 1u;
    for (uint32_t item = blockIdx.x; item < total_items; item += gridDim.x) {
        N2M_ACC_STAMP(0);
        uint32_t level = 0;
        while (item >= plan.item_prefix[level + 1]) ++level;
        // the level's largest magnitude through the VECTOR memory path: as a scalar load its ~1 us would be waited for (in-order scalar
        // counter) before the directory below could even be requested
        const uint32_t vm_v = __hip_atomic_load(level_max + level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t Gl = plan.groups[level];
        const bool store_all = overwrite && Gl == 1u;
        if (!store_all && (uint32_t)__builtin_amdgcn_readfirstlane((int)vm_v) == 0u) continue;     // no non-zero finite update in this level
        const uint32_t local = item - plan.item_prefix[level];
        const uint32_t part0 = (local / Gl) * SUB, grp = local - (local / Gl) * Gl;
        const uint32_t parts = plan.parts[level], size = plan.size[level];
        const uint32_t part_end = min(part0 + SUB, parts);                   // this item owns partitions part0 .. part_end-1
        const Indexer<3> ix(size, lv.resolution[level], gridtype, align_corners);
        const bool interleaved = !ix.hashed && parts > 1u;                  // same rule as bin_fill_kernel
        const uint32_t row0 = plan.row0[level];
        const PartMap pm(parts, part0, kLog2P, interleaved);
        const uint32_t n_blocks = (size + 15u) >> 4;
        auto rows_of = [&](uint32_t part) {
            const uint32_t my_blocks = interleaved ? (part < n_blocks ? (n_blocks - part + parts - 1) / parts : 0u)
                                                   : min(P / 16u, n_blocks - min(n_blocks, part * (P / 16u)));
            return my_blocks << 4;
        };
        auto global_row = [&](uint32_t u, uint32_t rel) {                    // table row of entry `rel` of partition part0 + u
            return interleaved ? ((((rel >> 4) * parts + part0 + u) << 4) | (rel & 15u)) : (((part0 + u) << kLog2P) + rel);
        };

        T* __restrict__ gtab = grad_table + (size_t)row0 * C;
        const uint32_t* __restrict__ dir_l = directory + plan.dir_base[level];
        // one wave per tile run.  The wave first fetches the directory entries of ALL its tiles in one parallel load (lane j <-
        // its j-th tile), then walks them with readlane: one dependent global-load round trip per item instead of one per tile.
        // The loads are issued BEFORE the accumulator is cleared, so that their latency hides behind the LDS stores and the barrier
        // (the per-item skeleton -- clear, directory, barriers -- is 25-30 us of each accumulate launch, tools/acc_lab.sh).
        const uint32_t my_tiles = (plan.tiles > grp + wid * Gl) ? (plan.tiles - (grp + wid * Gl) + 16u * Gl - 1u) / (16u * Gl) : 0u;   // <= 64
        uint32_t d_off = 0, d_mid = 0, d_end = 0;
        if (lane < my_tiles) {
            const uint32_t t = grp + wid * Gl + lane * 16u * Gl;
            const uint32_t* __restrict__ dir = dir_l + (size_t)t * (parts + 1u);
            d_off = dir[part0];
            d_end = dir[part_end];
            d_mid = SUB > 1u && part0 + 1u < part_end ? dir[part0 + 1u] : d_end;
        }
        for (uint32_t i = tid; i < SUB * P * C; i += 1024) bin_acc[i] = 0ull;
        if (tid == 0) nonfinite_seen = 0u;
        __syncthreads();
        N2M_ACC_STAMP(1);
        // unit = 2^-ex with |v| * 2^ex < 2^38 for every finite v of the level
        const uint32_t vm = (uint32_t)__builtin_amdgcn_readfirstlane((int)vm_v);
        int ex = 37 - ((int)((vm >> 23) & 255u) - 127);
        ex = ex < -126 ? -126 : (ex > 126 ? 126 : ex);
        const float scale = __uint_as_float((uint32_t)(ex + 127) << 23);
        const float inv = __uint_as_float((uint32_t)(127 - ex) << 23);
        // walk(body): body(rel0, u, bits) for every entry of this item's runs
        auto walk = [&](auto&& body) {
            const uint32_t n_t = vm != 0u ? my_tiles : 0u;
            if constexpr (SOA) {
                // A wave's runs sit in different tiles: short pieces (64-128 entries) far apart in the log, each a DRAM round trip of
                // ~1 us under load (shader-clock stamps: 16 serial trips = the whole walk).  The first 64 entries of kWalkAhead runs are
                // therefore requested together before any of them is accumulated; what a long run has beyond that follows serially.
#ifndef N2M_WALK_AHEAD
#define N2M_WALK_AHEAD 4
#endif
                constexpr uint32_t kWalkAhead = N2M_WALK_AHEAD;
                for (uint32_t j0 = 0; j0 < n_t; j0 += kWalkAhead) {
                    uint32_t r_rel[kWalkAhead], r_val[kWalkAhead], r_off[kWalkAhead], r_end[kWalkAhead], r_mid[kWalkAhead];
                    size_t r_seg[kWalkAhead];
#pragma unroll
                    for (uint32_t u = 0; u < kWalkAhead; ++u) {
                        const uint32_t j = j0 + u < n_t ? j0 + u : n_t - 1u;       // (clamped: the copy of the last run is not accumulated)
                        r_off[u] = (uint32_t)__builtin_amdgcn_readlane((int)d_off, j);
                        r_end[u] = j0 + u < n_t ? (uint32_t)__builtin_amdgcn_readlane((int)d_end, j) : 0u;
                        r_mid[u] = (uint32_t)__builtin_amdgcn_readlane((int)d_mid, j);
                        r_seg[u] = ((size_t)level * plan.tiles + (grp + wid * Gl + j * 16u * Gl)) * kTileEntries;
                        const uint32_t i = r_off[u] + lane;
                        r_rel[u] = 0u; r_val[u] = 0u;
                        if (i < r_end[u]) {
                            if (dbg & 4u) { r_rel[u] = i & (P - 1u); r_val[u] = 0x3f800000u; }
                            else { r_rel[u] = log_rel[r_seg[u] + i]; r_val[u] = log_val[r_seg[u] + i]; }
                        }
                    }
#pragma unroll
                    for (uint32_t u = 0; u < kWalkAhead; ++u) {
                        const uint32_t i = r_off[u] + lane;
                        if (i < r_end[u]) body(r_rel[u], (SUB > 1u && i >= r_mid[u]) ? 1u : 0u, r_val[u]);
                    }
#pragma unroll
                    for (uint32_t u = 0; u < kWalkAhead; ++u)
                        for (uint32_t i = r_off[u] + lane + 64u; i < r_end[u]; i += 64u) {
                            uint32_t rel0, bits;
                            if (dbg & 4u) { rel0 = i & (P - 1u); bits = 0x3f800000u; }
                            else { rel0 = log_rel[r_seg[u] + i]; bits = log_val[r_seg[u] + i]; }
                            body(rel0, (SUB > 1u && i >= r_mid[u]) ? 1u : 0u, bits);
                        }
                }
                return;
            }
            for (uint32_t j = 0; j < n_t; ++j) {
                const uint32_t t = grp + wid * Gl + j * 16u * Gl;
                const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)d_off, j), end = (uint32_t)__builtin_amdgcn_readlane((int)d_end, j);
                const uint32_t mid = (uint32_t)__builtin_amdgcn_readlane((int)d_mid, j);      // first entry of the second partition
                const size_t seg0 = ((size_t)level * plan.tiles + t) * kTileEntries;
                for (uint32_t i = off + lane; i < end; i += 64u) {
                    uint32_t rel0, bits;
                    if constexpr (SOA) {
                        if (dbg & 4u) { rel0 = i & (P - 1u); bits = 0x3f800000u; }
                        else {
                        rel0 = log_rel[seg0 + i];        // (streaming-load hints here measured 5 us slower)
                        bits = log_val[seg0 + i];
                        }
                    } else {
                        const uint64_t e = log[seg0 + i];
                        rel0 = (uint32_t)(e >> 32);
                        bits = (uint32_t)e;
                    }
                    body(rel0, (SUB > 1u && i >= mid) ? 1u : 0u, bits);
                }
            }
        };
        // inf / nan entries bypass the fixed-point sum and go to the table as they are.  When the flush STORES its rows (store_all)
        // that has to happen after the flush: the first walk only notes that there are any, a second walk (never taken in a healthy
        // run) adds them onto the stored sums.
        auto bypass = [&](uint32_t rel0, uint32_t u, uint32_t bits) {
            if constexpr (sizeof(T) == 4) unsafeAtomicAdd(gtab + global_row(u, rel0), __uint_as_float(bits));
            else {
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                (void)__builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) h2*)(gtab + (size_t)global_row(u, rel0) * 2u),
                                                                __builtin_bit_cast(h2, bits));
            }
        };
        auto finite = [&](uint32_t bits) {
            if constexpr (sizeof(T) == 4) return fabsf(__uint_as_float(bits)) <= 3.0e38f;
            else {
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const h2 p = __builtin_bit_cast(h2, bits);
                return fabsf((float)p.x) <= 3.0e38f && fabsf((float)p.y) <= 3.0e38f;
            }
        };
        walk([&](uint32_t rel0, uint32_t u, uint32_t bits) {
            const uint32_t rel = rel0 + u * P;                            // slot in this item's accumulator
            if (dbg & 1u) { if (bits == 0x12345u) bin_acc[rel] = 1; return; }
            if (!finite(bits)) {
                if (store_all) nonfinite_seen = 1u;
                else bypass(rel0, u, bits);
                return;
            }
            if constexpr (sizeof(T) == 4) {
                __hip_atomic_fetch_add(&bin_acc[rel], (unsigned long long)to_fixed(__uint_as_float(bits), scale), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const h2 p = __builtin_bit_cast(h2, bits);
                const float v0 = (float)p.x, v1 = (float)p.y;
                if (v0 != 0.f) __hip_atomic_fetch_add(&bin_acc[rel * 2u], (unsigned long long)to_fixed(v0, scale), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (v1 != 0.f) __hip_atomic_fetch_add(&bin_acc[rel * 2u + 1u], (unsigned long long)to_fixed(v1, scale), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        });
        N2M_ACC_STAMP(2);
        __syncthreads();
        N2M_ACC_STAMP(3);
        const bool second_walk = store_all && nonfinite_seen != 0u;       // read here: the next item resets the flag before ITS first barrier

        const bool fused_level = FUSE && level >= af.first_level;         // (store_all holds: these levels are owned by one work item)
        if (fused_level && !(dbg & 2u)) {
            // the optimizer pass for this item's rows: parameter and moments in (streaming, like adam_kernel's), the sums from LDS, the
            // update, the stores -- N2M_FUSE_BATCH values per thread requested together
            constexpr uint32_t IT = SUB * P / 1024u, IB = N2M_FUSE_BATCH / C;      // rows per thread, rows per batch of loads (registers: 64 per lane)
            typedef float vC __attribute__((ext_vector_type(C == 1 ? 1 : 2)));
            const float bc1 = af.bias[2u * af.slot], bc2_sqrt = af.bias[2u * af.slot + 1u];
            const float step_size = af.lr / bc1, inv_scale = af.scale ? 1.0f / *af.scale : 1.0f;
            const float bound = inf_bound > 0.0f ? inf_bound : (sizeof(T) == 4 ? 3.0e38f : 65504.0f);
            for (uint32_t ib = 0; ib < IT; ib += IB) {
                vC pi[IB], mi[IB], vi[IB];
                uint32_t grow[IB];
#pragma unroll
                for (uint32_t it = 0; it < IB; ++it) {
                    const uint32_t rel = tid + (ib + it) * 1024u, u = rel >> kLog2P, rel0 = rel & (P - 1u);
                    grow[it] = 0xffffffffu;
                    if (part0 + u >= part_end || rel0 >= rows_of(part0 + u)) continue;
                    const uint32_t row = global_row(u, rel0);
                    if (row >= size) continue;
                    grow[it] = row0 + row;
                    const size_t e = (size_t)grow[it] * C;
                    pi[it] = __builtin_nontemporal_load(reinterpret_cast<const vC*>(af.p_in + e));
                    mi[it] = __builtin_nontemporal_load(reinterpret_cast<const vC*>(af.m_in + e));
                    vi[it] = __builtin_nontemporal_load(reinterpret_cast<const vC*>(af.v_in + e));
                }
#pragma unroll
                for (uint32_t it = 0; it < IB; ++it) {
                    if (grow[it] == 0xffffffffu) continue;
                    const uint32_t rel = tid + (ib + it) * 1024u;
                    vC po, mo, vo;
#pragma unroll
                    for (uint32_t c = 0; c < C; ++c) {
                        const float f = (float)(long long)bin_acc[rel * C + c] * inv;
                        if (!(fabsf(f) <= bound) && found_inf) *found_inf = 1.0f;
                        // the gradient as the separate pass would read it back from the table: fp32, or the sum rounded to fp16 and widened again
                        const float gr = (sizeof(T) == 4 ? f : (float)(_Float16)f) * inv_scale;
                        const float m = af.beta1 * mi[it][c] + af.omb1 * gr;
                        const float v = af.beta2 * vi[it][c] + af.omb2 * gr * gr;
                        const float denom = sqrtf(v) / bc2_sqrt + af.eps;
                        po[c] = pi[it][c] - step_size * m / denom;
                        mo[c] = m; vo[c] = v;
                    }
                    const size_t e = (size_t)grow[it] * C;
                    __builtin_nontemporal_store(po, reinterpret_cast<vC*>(af.p_out + e));
                    __builtin_nontemporal_store(mo, reinterpret_cast<vC*>(af.m_out + e));
                    __builtin_nontemporal_store(vo, reinterpret_cast<vC*>(af.v_out + e));
                    if constexpr (C == 1) af.packed[(size_t)grow[it] * 2u] = __float_as_uint(po[0]);
                    else {
                        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                        h2 c16; c16.x = (_Float16)po[0]; c16.y = (_Float16)po[1];
                        af.packed[(size_t)grow[it] * 2u + 1u] = __builtin_bit_cast(uint32_t, c16);
                    }
                }
            }
        }
        for (uint32_t rel = tid; rel < ((dbg & 2u) || fused_level ? 0u : SUB * P); rel += 1024) {
            const uint32_t u = rel >> kLog2P, rel0 = rel & (P - 1u);
            if (part0 + u >= part_end || rel0 >= rows_of(part0 + u)) continue;
            const uint32_t row = global_row(u, rel0);
            if (row >= size) continue;
            if constexpr (sizeof(T) == 4) {
                const long long a = (long long)bin_acc[rel];
                if (a != 0 || store_all) {
                    const float f = (float)a * inv;
                    if (!(fabsf(f) <= (inf_bound > 0.0f ? inf_bound : 3.0e38f)) && found_inf) *found_inf = 1.0f;
                    if (store_all) gtab[row] = f;
                    else if (Gl == 1u) gtab[row] += f;
                    else unsafeAtomicAdd(gtab + row, f);
                }
            } else {
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const long long a0 = (long long)bin_acc[rel * 2u], a1 = (long long)bin_acc[rel * 2u + 1u];
                if ((a0 | a1) != 0 || store_all) {
                    const float f0 = (float)a0 * inv, f1 = (float)a1 * inv;
                    const float b16 = inf_bound > 0.0f ? inf_bound : 65504.0f;
                    if (!(fabsf(f0) <= b16 && fabsf(f1) <= b16) && found_inf) *found_inf = 1.0f;       // rounds to inf in fp16 (or could, summed over ranks)
                    h2* dst = reinterpret_cast<h2*>(gtab + (size_t)row * 2u);
                    if (store_all) {
                        h2 o;
                        o.x = (_Float16)f0;          // == (half)(0 + f): the sum below starts from +0
                        o.y = (_Float16)f1;
                        *dst = o;
                    } else if (Gl == 1u) {
                        h2 o = *dst;
                        o.x = (_Float16)((float)o.x + f0);
                        o.y = (_Float16)((float)o.y + f1);
                        *dst = o;
                    } else {
                        h2 val;
                        val.x = (_Float16)f0;
                        val.y = (_Float16)f1;
                        (void)__builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) h2*)dst, val);
                    }
                }
            }
        }
        __syncthreads();
        N2M_ACC_STAMP(4);
        if (second_walk) {
            __threadfence();                                               // the stores above before the atomics below
            walk([&](uint32_t rel0, uint32_t u, uint32_t bits) { if (!finite(bits)) bypass(rel0, u, bits); });
            __syncthreads();
        }
    }
}

// =====================================================================================================================================
// Partition-major update log (round 4).  The tile-major log above keeps every tile's entries together (one contiguous segment per tile,
// sorted by partition) and leaves the gathering to the accumulate: a work item walks one SHORT run per tile (64-130 entries = a 128 B +
// a 256 B piece somewhere in a 170 MB log) -- DRAM efficiency on short scattered READS, which a wave has to wait for, is what that
// kernel costs, and it forces 1024-sample tiles (shorter runs lost more in the accumulate than they won in the fill).  Here the
// scatter is moved to the WRITE side, which nobody waits for: every (level, partition) owns one contiguous region of the log, a fill
// workgroup reserves room for its tile's run with ONE returning atomic per partition (cursor[level][partition] += n) and stores the
// run there; the accumulate then streams ONE contiguous array per partition -- no directory, no per-tile round trips -- and tiles can
// be as small as the fill likes: 512-sample tiles = two independent fill workgroups per CU whose phases (stencil gathers | LDS sort |
// stores) overlap each other, which is what the barrier-serialised 1024-thread workgroup could not do.  The order of the runs inside
// a region depends on the atomics' order; the sum is 64-bit fixed point, i.e. exact and order-independent as before.
//   Region capacity is 2 x the partition's expected share (+ slack): a run that does not fit goes to ONE shared overflow log
// (bump-allocated records {level | partition | row, value, value}); a work item whose cursor passed its capacity scans that log for
// its records.  Never taken by marched samples (hashed rows are uniform, dense levels deal 16-row blocks round-robin); a batch with
// every sample in one cell takes it and stays exact.  The overflow log holds every entry of the pass, so nothing can be dropped.
struct PmPlan {
    uint32_t cur_base[kMaxLevels];      // first cursor word of the level (one u32 per partition)
    uint32_t home_cap[kMaxLevels];      // entries one partition's region holds
    uint32_t home_base[kMaxLevels];     // first entry of the level's regions: partition p at home_base + p * home_cap
    uint32_t max_parts, cursors;        // largest partition count of a level (LDS sizing), cursor words in all
    uint32_t ovf_cap;                   // records the overflow log holds (= every entry of the pass)
};

// Lean form of pair_entries for pm_fill_pair_kernel (the fill's SIMDs issue VALU work more than half of the time: ~815 instructions per
// (sample, level) before this): the colour products leave as PACKED halves (no f32 round trip on the levels that do not merge runs),
// smoothstep only when asked for (a uniform branch instead of a select over both forms), the partition key in two instructions.
// DEAD: every sample of the wave has a zero feature gradient on this level (wave-uniform, decided by the caller): the only entry such a sample
// can deliver is its TV term on vertex 000 -- cell, the rows the stencil reads and that one partition key are all that is computed.
template <int TV, int IMODE, bool ILV, bool DEAD = false>
__device__ __forceinline__ void pm_entries(const PairCtx& cx, const Indexer<3>& ix, const PartMap& pm, const float (&x)[3], float g1,
                                           float g2x, float g2y, float a1, float& vmax1, uint32_t (&e_pr)[8], float (&f1)[8],
                                           uint32_t (&p2)[8], uint32_t (&cell)[3], float tv_given, float& tv_out, const float4* corners = nullptr) {
    constexpr uint32_t D = 3;
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    float frac[D];
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) {
        float p = x[d] * cx.scale + (cx.align_corners ? 0.0f : 0.5f);
        cell[d] = (uint32_t)floorf(p);
        frac[d] = p - (float)cell[d];
    }
    if (!DEAD && cx.interp == 1) {
        asm volatile("" ::: "memory");                                   // (keeps this a branch: linear interpolation is what nerf2mesh runs)
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) frac[d] = frac[d] * frac[d] * (3.0f - 2.0f * frac[d]);
    }
    const uint32_t sy = IMODE == 1 ? kPrimes[1] : ix.stride[1], sz = IMODE == 1 ? kPrimes[2] : ix.stride[2];
    const uint32_t tx[2] = {cell[0], cell[0] + 1u};
    const uint32_t ty0 = cell[1] * sy, tz0 = cell[2] * sz;
    const uint32_t ty[2] = {ty0, ty0 + sy}, tz[2] = {tz0, tz0 + sz};
    auto comb = [&](uint32_t a, uint32_t b, uint32_t c) { return IMODE == 1 ? ((a ^ b ^ c) & ix.mask) : (a + b + c); };
    uint32_t rows[8];
#pragma unroll
    for (uint32_t corner = 0; corner < 8; ++corner) {
        const uint32_t i = corner & 1u, j = (corner >> 1) & 1u, k = corner >> 2;
        if constexpr (IMODE != 0) rows[corner] = comb(tx[i], ty[j], tz[k]);
        else {
            const uint32_t v[D] = {cell[0] + i, cell[1] + j, cell[2] + k};
            rows[corner] = ix.row(v);
        }
    }
    float tvv = 0.0f;
    if constexpr (TV != 0) {
        if constexpr (TV == 1) tvv = pair_tv_value<IMODE>(cx, ix, x, cell, rows, tx, ty, tz, sy, sz, corners);
        else tvv = tv_given;
        const float a = fabsf(tvv);
        vmax1 = fmaxf(vmax1, (a1 <= 3.0e38f ? a1 : 1.0f) + (a <= 3.0e38f ? a : 1.0f));
    }
    tv_out = tvv;
    const float wx[2] = {1 - frac[0], frac[0]}, wy[2] = {1 - frac[1], frac[1]}, wz[2] = {1 - frac[2], frac[2]};
#pragma unroll
    for (uint32_t corner = 0; corner < (DEAD ? 1u : 8u); ++corner) {
        const uint32_t i = corner & 1u, j = (corner >> 1) & 1u, k = corner >> 2;
        if constexpr (DEAD) {
            f1[0] = tvv;                                             // (w * g1 is +-0 here: the live path's sum is the TV term, or a zero that is dropped)
            p2[0] = 0u;
        } else {
            const float w = (wx[i] * wy[j]) * wz[k];                     // forward's association
            float p1 = w * g1;
            if (TV != 0 && corner == 0) p1 += tvv;
            f1[corner] = p1;
            float pa = w * g2x, pb = w * g2y;                            // rounded to fp32, THEN to half (gridencoder.cu:326; see half_product)
            asm volatile("" : "+v"(pa), "+v"(pb));
            h2 hp;
            hp.x = (_Float16)pa;
            hp.y = (_Float16)pb;
            p2[corner] = __builtin_bit_cast(uint32_t, hp);
        }
        const uint32_t row = rows[corner];
        if constexpr (IMODE == 0) {
            uint32_t part_, rel_;
            pm.split(row, part_, rel_);
            e_pr[corner] = (part_ << 16) | rel_;
        } else if constexpr (ILV) {
            const uint32_t blk = row >> 4;
            const uint32_t q = __umulhi(blk, pm.magic);
            e_pr[corner] = ((blk - q * pm.parts) << 16) | (q << 4) | (row & 15u);
        } else {
            // (row >> log2p) << 16 | row & (2^log2p - 1)  ==  row + (row >> log2p) * (65536 - 2^log2p)
            e_pr[corner] = row + (row >> pm.log2p) * (65536u - (1u << pm.log2p));
        }
    }
}

// merge_runs for a wave whose samples deliver at most their TV term: one value per lane instead of twenty-four.
__device__ __forceinline__ bool merge_runs_tv(bool inside, const uint32_t (&cell)[3], float& f, uint32_t lane) {
    const uint32_t key0 = inside ? cell[0] : 0xFFFFFFFFu;
    const uint32_t p0 = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)key0, 0x111, 0xF, 0xF, false);
    const uint32_t p1 = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)cell[1], 0x111, 0xF, 0xF, false);
    const uint32_t p2 = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)cell[2], 0x111, 0xF, 0xF, false);
    const bool same = inside && (lane & 15u) != 0u && p0 == key0 && p1 == cell[1] && p2 == cell[2];
    const unsigned long long heads = __ballot(!same);
    const unsigned long long upto = heads & (~0ull >> (63u - lane));
    const uint32_t dist = lane - (63u - (uint32_t)__builtin_clzll(upto));
    const bool last = lane == 63u || ((heads >> (lane + 1u)) & 1ull);
    if (__builtin_popcountll(heads) == 64) return true;
#pragma unroll
    for (int d = 1; d <= 8; d <<= 1) {
        const bool take = dist >= (uint32_t)d;
        if (__ballot(take) == 0ull) break;
        const float a = dpp_row_shr(f, d);
        if (take) f += a;
    }
    return last;
}

// TS = samples per tile = threads per workgroup.  LDS (dynamic): three u32 staging arrays of 8 TS entries, then
// cnt[2][MP] start[MP] delta[MP] ovfb[MP] with MP = pm.max_parts rounded up to 128.
#ifndef N2M_PM_WAVES
#define N2M_PM_WAVES 4
#endif
#ifndef N2M_PM_FINE_WAVES
#define N2M_PM_FINE_WAVES 6
#endif
#ifndef N2M_PM_PREFETCH_LATE
#define N2M_PM_PREFETCH_LATE 0
#endif
// EX (template): the round-5 additions of the fill -- a caller-given sample order, the TV-only path of waves of dead samples, phase stamps.
// They cost the plain kernel 6.5 us per launch even when unused (measured, ABBA on one box: 213.9 vs 207.2 us), so they are a second
// instantiation the launcher picks only when an order is set or a measurement switch is on.
template <int TV, bool FOLD, uint32_t TS, bool EX = false>
__global__ void __launch_bounds__(TS) __attribute__((amdgpu_waves_per_eu(N2M_PM_WAVES, N2M_PM_WAVES)))
pm_fill_pair_kernel(const float* __restrict__ grad1 /*[L,Bstride]*/, const _Float16* __restrict__ grad2 /*[L,Bstride,2]*/,
                    const float* __restrict__ inputs, TvParams tv, const float* __restrict__ tv_terms, uint32_t B, uint32_t Bstride, BinPlan plan,
                    PmPlan pm, LevelTable lv, uint32_t gridtype, bool align_corners, uint32_t interp, uint32_t* __restrict__ level_max /*[2][32]*/,
                    uint32_t* __restrict__ cursors, uint32_t* __restrict__ ovf_cursor, uint16_t* __restrict__ log_rel, uint32_t* __restrict__ log_v1,
                    uint32_t* __restrict__ log_v2, uint32_t* __restrict__ ovf_key, uint32_t* __restrict__ ovf_v1, uint32_t* __restrict__ ovf_v2,
                    float* __restrict__ found_inf, float in_scale, float in_offset, float* __restrict__ clear1, _Float16* __restrict__ clear2,
                    uint32_t clear_mask1, uint32_t clear_mask2, uint32_t merge_levels, uint32_t groups_x, uint32_t slot_begin,
                    unsigned long long* __restrict__ lm_ready, unsigned long long lm_token, uint32_t in_level_stride, FoldArgs fold,
                    const uint32_t* __restrict__ perm /*[B] or NULL: the order the samples are visited in (n2m_grid_backward_sample_order)*/) {
    constexpr uint32_t D = 3, kWaves = TS / 64u, kEntries = TS * 8u;
    constexpr uint32_t kLog2P = 31u - __builtin_clz(kPairP);
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    // Workgroup 0 (always resident first) clears the level maxima, the cursors and the overflow cursor, then publishes this launch's
    // token; a workgroup waits for the token before its first reservation (a whole entries phase later: never seen to spin).
    // (bit 31 of slot_begin: this launch CONTINUES a fill -- pm_fill_fine_kernel has cleared and published the token for the fine levels,
    //  whose cursors and maxima are live -- so workgroup 0 clears nothing and every token wait passes at once)
    const bool continues = (slot_begin >> 31) != 0u;
    slot_begin &= 0x7FFFFFFFu;
    if (blockIdx.x == 0u && blockIdx.y == 0u && !continues) {
        if (threadIdx.x < 2u * kMaxLevels) level_max[threadIdx.x] = 0u;
        for (uint32_t i = threadIdx.x; i < pm.cursors; i += TS) cursors[i] = 0u;
        if (threadIdx.x == 0u) *ovf_cursor = 0u;
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0u) __hip_atomic_store(lm_ready, lm_token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    extern __shared__ __attribute__((aligned(16))) uint32_t pm_lds[];
    const uint32_t MP = (pm.max_parts + 127u) & ~127u;
    uint32_t* stage_v1 = pm_lds;
    uint32_t* stage_v2 = stage_v1 + kEntries;
    uint32_t* stage_e = stage_v2 + kEntries;             // partition << 16 | row in partition
    uint32_t* cnt2 = stage_e + kEntries;                 // [2][MP]
    uint32_t* start = cnt2 + 2u * MP;                    // first position of the partition's run in the sorted tile
    uint32_t* delta = start + MP;                        // log index of an entry = delta[partition] + its position in the tile
    uint32_t* ovfb = delta + MP;                         // overflowing runs only: first record of the run's tail in the overflow log
    __shared__ uint32_t wave_max[2][kWaves], tile_ovf[2];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    uint32_t level = blockIdx.y, group = blockIdx.x, n_groups = gridDim.x;
    if (groups_x != 0u) {                                // XCD-aware 1-D grid: see bin_fill_pair_kernel
        const uint32_t xcd = blockIdx.x & 7u, k = blockIdx.x >> 3;
        const uint32_t ls = k / groups_x;
        const uint32_t slot = slot_begin + ls;
        group = k - ls * groups_x;
        n_groups = groups_x;
        const uint32_t pairi = xcd + 8u * (slot >> 1);
        level = (slot & 1u) ? pairi : plan.levels - 1u - pairi;
        if (level >= plan.levels || pairi > plan.levels - 1u - pairi || ((slot & 1u) && pairi == plan.levels - 1u - pairi)) return;
    }
    const uint32_t parts = plan.parts[level], size = plan.size[level];
    for (uint32_t i = tid; i < 2u * MP; i += TS) cnt2[i] = 0u;
    if (tid < 2u) tile_ovf[tid] = 0u;
    if ((clear_mask1 | clear_mask2) >> level & 1u) {     // overwrite mode: levels that receive atomics start from zero
        const uint32_t per = (size + n_groups - 1) / n_groups, lo = min(size, group * per), hi = min(size, lo + per);
        const size_t r0 = plan.row0[level];
        if (clear_mask1 >> level & 1u)
            for (uint32_t i = lo + tid; i < hi; i += TS) clear1[r0 + i] = 0.0f;
        if (clear_mask2 >> level & 1u)
            for (uint32_t i = lo + tid; i < hi; i += TS) reinterpret_cast<uint32_t*>(clear2)[r0 + i] = 0u;
    }
    const float scale = lv.scale[level];
    const Indexer<D> ix(size, lv.resolution[level], gridtype, align_corners);
    const PartMap pmap(parts, 0, kLog2P, !ix.hashed && parts > 1u);
    const bool fast_hash = ix.hashed && ix.pow2, fast_dense = !ix.hashed && !ix.wrap;
    const uint32_t cap = pm.home_cap[level], region0 = pm.home_base[level];
    uint32_t* __restrict__ cur_l = cursors + pm.cur_base[level];
    float vmax1 = 0.0f, vmax2 = 0.0f;
    bool token_seen = false;
    const PairCtx cx = make_pair_ctx(tv, tv.table ? tv.table + (size_t)plan.row0[level] * tv.stride : nullptr, scale, lv.resolution[level], align_corners, interp);
    const uint32_t dbg = g_fill_timing_on;               // measurement switches (tools/pm_lab.sh; wrong results when set)
    if ((dbg >> 8) != 0u && level != (dbg >> 8) - 1u) return;          // (measurement: one level alone, tools/pm_levels.sh)

    // (measurement, n2m_debug_fill_times bit 0: shader-clock stamps of two workgroups' first 8 tiles -- lane 0 of wave 0, and of the LAST wave in
    //  the second half of the record: loop top | entries ready | barrier 1 passed | staged | barrier 2 passed | log stores issued)
    const bool stamp = EX && (dbg & 1u) != 0u && (tid == 0u || tid == TS - 64u) && (blockIdx.x == 3u || blockIdx.x == gridDim.x / 2u + 3u);
    const uint32_t stamp_w = blockIdx.x == 3u ? 0u : 1u;
#define N2M_PM_STAMP(i) do { if (stamp && it < 4u) g_fill_t[stamp_w][it + (tid == 0u ? 0u : 4u)][(i)] = __builtin_readcyclecounter(); } while (0)
    uint32_t tile = group;
    float nx[D] = {2.f, 2.f, 2.f}, ng1 = 0.0f, ntv = 0.0f;
    h2 ng2 = {(_Float16)0, (_Float16)0};
    // the next tile's inputs, RAW: the affine input map is applied when they are consumed, a tile later.  (Until round 4 the map sat here,
    // right behind the loads -- the compiler put s_waitcnt vmcnt(0) between them and the "prefetch" was a full DRAM round trip at the top of
    // every tile, draining the previous tile's log stores with it.)
    bool nvalid = false, pvalid = false;
    uint32_t pidx = 0u;
    // (the forward lookup's corner records, tv.corners: read where they are used.  Requested a tile ahead with the inputs they cost four more
    //  registers across both barriers: 122 VGPRs + 32 bytes of scratch, measured slower -- round 6)
    const bool use_corners = TV == 1 && !EX && !FOLD && tv.corners != nullptr && fast_hash;
    // which sample this thread visits in tile t: loaded TWO tiles ahead when the caller hands an order (the inputs of tile t+1 are requested
    // at the top of tile t and need the index then -- a load issued only there would be a dependent round trip in front of every request)
    auto fetch_index = [&](uint32_t t) {
        const uint32_t s0 = t * TS + tid;
        pvalid = t < plan.tiles && s0 < B;
        pidx = pvalid ? ((EX && perm) ? perm[s0] : s0) : 0u;
    };
    auto request = [&]() {
        nvalid = pvalid;
        if (nvalid) {
            const uint32_t s = pidx;
            load_point<D>(inputs + (size_t)level * in_level_stride, s, nx);
            ng1 = grad1 ? grad1[(size_t)level * Bstride + s] : 0.0f;
            if (grad2) ng2 = *reinterpret_cast<const h2*>(grad2 + ((size_t)level * Bstride + s) * 2);
            else ng2 = h2{(_Float16)0, (_Float16)0};
            if (TV == 2) ntv = tv_terms[(size_t)level * Bstride + s];
        }
    };
    fetch_index(tile);
    request();
    fetch_index(tile + n_groups);
    __syncthreads();

    for (uint32_t it = 0; tile < plan.tiles; tile += n_groups, ++it) {
        N2M_PM_STAMP(0);
        uint32_t* cnt = cnt2 + (it & 1u) * MP;
        uint32_t* cnt_next = cnt2 + ((it & 1u) ^ 1u) * MP;
        float x[D];
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) x[d] = nvalid ? nx[d] * in_scale + in_offset : 2.0f;
        const float g1 = ng1, tvg = ntv;
        const h2 g2 = ng2;
#if !N2M_PM_PREFETCH_LATE
        request();                                       // tile + n_groups
        fetch_index(tile + 2u * n_groups);
#endif

        uint32_t e_pr[8], e_v1[8], e_v2[8], e_slot[8];
        float f1[8];
        uint32_t cell[D] = {0u, 0u, 0u};
        uint32_t vmask = 0;
        const bool inside = !outside_unit_cube<D>(x);
        bool gnz = false, tvnz = false;                  // any gradient / a TV term to deliver for this sample on this level
#pragma unroll
        for (uint32_t c = 0; c < 8; ++c) { f1[c] = 0.f; e_v2[c] = 0u; e_pr[c] = 0u; }
        if (inside) {
            const float g2x = (float)g2.x, g2y = (float)g2.y;
            const float a1 = fabsf(g1), a2 = fmaxf(fabsf(g2x), fabsf(g2y));
            vmax1 = fmaxf(vmax1, a1 <= 3.0e38f ? a1 : 1.0f);
            vmax2 = fmaxf(vmax2, a2 <= 3.0e38f ? a2 : 1.0f);
            const bool bad2 = !(fabsf(g2x) <= 3.0e38f) || !(fabsf(g2y) <= 3.0e38f);
            if ((!(a1 <= 3.0e38f) || bad2) && found_inf) *found_inf = 1.0f;
            if (bad2) vmax2 = fmaxf(vmax2, 1.0f);
            gnz = (g1 != 0.0f) | ((__builtin_bit_cast(uint32_t, g2) & 0x7FFF7FFFu) != 0u);
        }
        // A wave none of whose samples carries a gradient on this level (the dead tails of the rays, visited together when the caller
        // hands the live-first order: n2m_grid_backward_sample_order) delivers TV terms only: one entry per sample instead of eight, one
        // value through the run merge instead of twenty-four.  Wave-uniform; the folded-copies form keeps the full path.
        const bool wave_live = !EX || FOLD || (dbg & 128u) || __ballot(inside && gnz) != 0ull;       // (dbg 128: measurement switch, the full path for every wave)
        // the forward lookup's record of this (level, sample): only where the launcher handed one over (TV computed here, samples in input order)
        const float4* corner_rec = use_corners ? reinterpret_cast<const float4*>(tv.corners) + ((size_t)level * Bstride + tile * TS + tid) : nullptr;
        if (inside) {
            const float g2x = (float)g2.x, g2y = (float)g2.y;
            const float a1 = fabsf(g1);
            float tvv = 0.0f;
            if (wave_live) {
                if (fast_hash) pm_entries<TV, 1, false>(cx, ix, pmap, x, g1, g2x, g2y, a1, vmax1, e_pr, f1, e_v2, cell, tvg, tvv, corner_rec);
                else if (fast_dense && parts > 1u) pm_entries<TV, 2, true>(cx, ix, pmap, x, g1, g2x, g2y, a1, vmax1, e_pr, f1, e_v2, cell, tvg, tvv);
                else if (fast_dense) pm_entries<TV, 2, false>(cx, ix, pmap, x, g1, g2x, g2y, a1, vmax1, e_pr, f1, e_v2, cell, tvg, tvv);
                else pm_entries<TV, 0, false>(cx, ix, pmap, x, g1, g2x, g2y, a1, vmax1, e_pr, f1, e_v2, cell, tvg, tvv);
            } else if (TV != 0) {
                if (fast_hash) pm_entries<TV, 1, false, true>(cx, ix, pmap, x, g1, g2x, g2y, a1, vmax1, e_pr, f1, e_v2, cell, tvg, tvv, corner_rec);
                else if (fast_dense && parts > 1u) pm_entries<TV, 2, true, true>(cx, ix, pmap, x, g1, g2x, g2y, a1, vmax1, e_pr, f1, e_v2, cell, tvg, tvv);
                else if (fast_dense) pm_entries<TV, 2, false, true>(cx, ix, pmap, x, g1, g2x, g2y, a1, vmax1, e_pr, f1, e_v2, cell, tvg, tvv);
                else pm_entries<TV, 0, false, true>(cx, ix, pmap, x, g1, g2x, g2y, a1, vmax1, e_pr, f1, e_v2, cell, tvg, tvv);
            }
            tvnz = tvv != 0.0f;
            if constexpr (FOLD) {
                const uint32_t sidx = tile * TS + tid;
                const uint32_t fl = fold.flags[(size_t)level * B + sidx];
                if (fl != 0u) { fold_copies(fold, sidx, level, x, scale, align_corners, interp, fl, f1, vmax1, found_inf); gnz = true; }
            }
        }
#if N2M_PM_PREFETCH_LATE
        request();                                       // behind the stencil gathers: the vector memory counter is in order, a wait for a gather also waits for every older load
        fetch_index(tile + 2u * n_groups);
#endif
        if (level < merge_levels && !(dbg & 64u) && !wave_live) {
            // a wave of TV-only samples: one value per lane through the run merge
            const bool keep = merge_runs_tv(inside, cell, f1[0], lane) && inside;
            if (keep) {
                const float m1 = fabsf(f1[0]);
                vmax1 = fmaxf(vmax1, m1 <= 3.0e38f ? m1 : 1.0f);
                if (!(m1 <= 3.0e38f) && found_inf) *found_inf = 1.0f;
            }
#pragma unroll
            for (uint32_t c = 0; c < 8; ++c) e_v1[c] = 0u;
            e_v1[0] = __float_as_uint(f1[0]);
            if (keep && (e_v1[0] << 1) != 0u) vmask = 1u;
        } else
        if (level < merge_levels && !(dbg & 64u)) {      // block-uniform: same-cell runs of consecutive samples become one entry per vertex
            float f2x[8], f2y[8];
#pragma unroll
            for (uint32_t c = 0; c < 8; ++c) { const h2 p = __builtin_bit_cast(h2, e_v2[c]); f2x[c] = (float)p.x; f2y[c] = (float)p.y; }
            const bool keep = merge_runs(inside, cell, f1, f2x, f2y, lane) && inside;
            if (keep) {                                  // a run's sum can exceed every one of its terms
                float m1 = 0.f, m2 = 0.f;
#pragma unroll
                for (uint32_t c = 0; c < 8; ++c) { m1 = fmaxf(m1, fabsf(f1[c])); m2 = fmaxf(m2, fmaxf(fabsf(f2x[c]), fabsf(f2y[c]))); }
                vmax1 = fmaxf(vmax1, m1 <= 3.0e38f ? m1 : 1.0f);
                vmax2 = fmaxf(vmax2, m2 <= 65504.0f ? m2 : 65504.0f);
                if ((!(m1 <= 3.0e38f) || !(m2 <= 65504.0f)) && found_inf) *found_inf = 1.0f;
            }
#pragma unroll
            for (uint32_t c = 0; c < 8; ++c) {
                h2 p2;
                p2.x = (_Float16)f2x[c];
                p2.y = (_Float16)f2y[c];
                e_v1[c] = __float_as_uint(f1[c]);
                e_v2[c] = __builtin_bit_cast(uint32_t, p2);
                if (keep && ((e_v1[c] << 1) | (e_v2[c] & 0x7FFF7FFFu)) != 0u) vmask |= 1u << c;
            }
        } else {
            // no merging on this level: a sample with a gradient delivers its eight entries (a vertex of weight zero: an explicit zero the
            // accumulate skips), a sample without one at most its TV term on vertex 000
#pragma unroll
            for (uint32_t c = 0; c < 8; ++c) e_v1[c] = __float_as_uint(f1[c]);
            vmask = !inside ? 0u : (gnz ? 0xFFu : (tvnz ? 1u : 0u));
        }
        N2M_PM_STAMP(1);
        // slot of every entry inside its partition's run of this tile
        if (dbg & 16u) {
#pragma unroll
            for (uint32_t c = 0; c < 8; ++c) e_slot[c] = c;
        } else if (parts == 1u) {
            const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
            for (uint32_t c = 0; c < 8; ++c) {
                if (c != 0u && !wave_live) break;                        // (a TV-only wave holds entries on vertex 000 only)
                const bool v = (vmask >> c) & 1u;
                const unsigned long long m = __ballot(v);
                uint32_t base = 0;
                if (lane == 0 && m) base = atomicAdd(&cnt[0], (uint32_t)__popcll(m));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                e_slot[c] = base + (uint32_t)__popcll(m & below);
            }
        } else {
#pragma unroll
            for (uint32_t c = 0; c < 8; ++c) {
                if (c != 0u && !wave_live) break;
                if ((vmask >> c) & 1u) e_slot[c] = atomicAdd(&cnt[e_pr[c] >> 16], 1u);
            }
        }
        __syncthreads();                                                     // (1) counters complete
        N2M_PM_STAMP(2);

        // Run starts: EVERY wave scans the counters itself (128 per step, two per lane) and writes the same values -- a wave reads
        // back what it has written itself (LDS operations of a wave complete in order), so no barrier separates scan and staging.
        // Wave 0 also reserves the runs' room in the partitions' regions.
        uint32_t total = 0;
        uint32_t q0 = 0u, q1 = 0u, ra0 = 0u, ra1 = 0u, rs0 = 0u, rs1 = 0u;     // wave 0: the first 128 partitions' reservation (in flight over the staging)
        for (uint32_t c0 = 0; c0 < parts; c0 += 128u) {
            const uint32_t i0 = c0 + 2u * lane, i1 = i0 + 1u;
            const uint32_t a0 = i0 < parts ? cnt[i0] : 0u, a1c = i1 < parts ? cnt[i1] : 0u;
            const uint32_t incl = n2m_wave_scan_add_u32(a0 + a1c, (int)lane);
            const uint32_t s0 = total + incl - (a0 + a1c), s1 = s0 + a0;
            if (i0 < parts) start[i0] = s0;
            if (i1 < parts) start[i1] = s1;
            total += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (wid == 0u && c0 == 0u) {
                if (!token_seen) {
                    while (__hip_atomic_load(lm_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != lm_token) __builtin_amdgcn_s_sleep(8);
                    token_seen = true;
                }
                ra0 = a0; ra1 = a1c; rs0 = s0; rs1 = s1;
                if (dbg & 32u) { q0 = min(tile * 36u, cap - min(cap, a0)); q1 = min(tile * 36u, cap - min(cap, a1c)); }    // (measurement: a reservation without its round trip)
                else {
                    if (a0) q0 = atomicAdd(cur_l + i0, a0);
                    if (a1c) q1 = atomicAdd(cur_l + i1, a1c);
                }
            }
        }
        if (!(dbg & 8u)) {
            uint32_t pos[8];
#pragma unroll
            for (uint32_t c = 0; c < 8; ++c) {
                if (c != 0u && !wave_live) break;
                pos[c] = start[e_pr[c] >> 16];          // (all eight lookups in flight; a dropped entry reads partition 0's)
            }
#pragma unroll
            for (uint32_t c = 0; c < 8; ++c) {
                if (c != 0u && !wave_live) break;
                if ((vmask >> c) & 1u) {
                    const uint32_t at = pos[c] + e_slot[c];
                    stage_v1[at] = e_v1[c];
                    stage_v2[at] = e_v2[c];
                    stage_e[at] = e_pr[c];
                }
            }
        }
        for (uint32_t i = tid; i < parts; i += TS) cnt_next[i] = 0;
        if (tid == 0u) tile_ovf[(it & 1u) ^ 1u] = 0u;
        if (wid == 0u) {
            bool ovf_here = false;
            auto finish = [&](uint32_t i, uint32_t a, uint32_t s, uint32_t q) {
                if (a == 0u) return;
                delta[i] = region0 + i * cap + q - s;
                if (q + a > cap) {                       // (part of) the run does not fit its region: the shared overflow log takes the rest
                    ovfb[i] = atomicAdd(ovf_cursor, min(a, q + a - cap));
                    ovf_here = true;
                }
            };
            finish(2u * lane, ra0, rs0, q0);
            finish(2u * lane + 1u, ra1, rs1, q1);
            for (uint32_t c0 = 128u; c0 < parts; c0 += 128u) {             // tables with more than 128 partitions per level
                const uint32_t i0 = c0 + 2u * lane, i1 = i0 + 1u;
                const uint32_t a0 = i0 < parts ? cnt[i0] : 0u, a1c = i1 < parts ? cnt[i1] : 0u;
                uint32_t p0 = 0u, p1 = 0u;
                if (a0) p0 = atomicAdd(cur_l + i0, a0);
                if (a1c) p1 = atomicAdd(cur_l + i1, a1c);
                finish(i0, a0, a0 ? start[i0] : 0u, p0);
                finish(i1, a1c, a1c ? start[i1] : 0u, p1);
            }
            if (__ballot(ovf_here) != 0ull && lane == 0u) tile_ovf[it & 1u] = 1u;
        }
        N2M_PM_STAMP(3);
        __syncthreads();                                                     // (2) tile staged, regions reserved
        N2M_PM_STAMP(4);

        if (dbg & 12u) {
        } else if (tile_ovf[it & 1u] == 0u) {
            // A thread moves up to eight entries.  All their LDS reads are issued together, then the eight lookups of the runs' places, then
            // the stores: as a loop (one entry per trip: read -> wait -> lookup -> wait -> store) the copy was 2 dependent LDS round trips
            // per entry, serial -- the "cost of the log stores" of the earlier ablations was this latency chain, not the stores.
            // Plain (cache-allocating) stores: a run is a 64-256 byte piece at an arbitrary offset of its region, neighbouring runs complete
            // each other's lines in L2 (streaming stores: 190 us per fill instead of 172).  Byte offsets in 32 bits (the host checks that the
            // regions end below 2^30 entries): the stores take the arrays' bases from scalar registers.
            uint32_t ce[8], cv1[8], cv2[8], cd[8];
#pragma unroll
            for (uint32_t q = 0; q < 8; ++q) {
                const uint32_t i = tid + q * TS;
                ce[q] = 0u; cv1[q] = 0u; cv2[q] = 0u;
                if (i < total) { ce[q] = stage_e[i]; cv1[q] = stage_v1[i]; cv2[q] = stage_v2[i]; }
            }
#pragma unroll
            for (uint32_t q = 0; q < 8; ++q) cd[q] = delta[ce[q] >> 16] + tid + q * TS;
#pragma unroll
            for (uint32_t q = 0; q < 8; ++q)
                if (tid + q * TS < total) {
                    if (dbg & 2u) {                  // (measurement switch: streaming stores)
                        if (log_v1) __builtin_nontemporal_store(cv1[q], &log_v1[cd[q]]);
                        if (log_v2) __builtin_nontemporal_store(cv2[q], &log_v2[cd[q]]);
                        __builtin_nontemporal_store((uint16_t)ce[q], &log_rel[cd[q]]);
                    } else {
                        if (log_v1) *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(log_v1) + (cd[q] << 2)) = cv1[q];
                        if (log_v2) *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(log_v2) + (cd[q] << 2)) = cv2[q];
                        *reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(log_rel) + (cd[q] << 1)) = (uint16_t)ce[q];
                    }
                }
        } else {
            for (uint32_t i = tid; i < total; i += TS) {
                const uint32_t e = stage_e[i], p = e >> 16;
                const uint32_t st = start[p];
                const uint32_t qr = delta[p] + st - (region0 + p * cap);           // where the run starts in its region
                const uint32_t qi = qr + (i - st);                                  // this entry's place in the partition's stream
                if (qi < cap) {
                    const uint32_t d = delta[p] + i;
                    if (log_v1) log_v1[d] = stage_v1[i];
                    if (log_v2) log_v2[d] = stage_v2[i];
                    log_rel[d] = (uint16_t)e;
                } else {
                    const uint32_t o = ovfb[p] + qi - max(qr, cap);
                    if (o < pm.ovf_cap) {
                        ovf_key[o] = (level << 27) | e;
                        if (ovf_v1) ovf_v1[o] = stage_v1[i];
                        if (ovf_v2) ovf_v2[o] = stage_v2[i];
                    }
                }
            }
        }
        N2M_PM_STAMP(5);
        // no barrier here: the next tile writes the stage / start / delta only after its barrier (1)
    }

#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        vmax1 = fmaxf(vmax1, __shfl_xor(vmax1, o, 64));
        vmax2 = fmaxf(vmax2, __shfl_xor(vmax2, o, 64));
    }
    if (lane == 0) { wave_max[0][wid] = __float_as_uint(vmax1); wave_max[1][wid] = __float_as_uint(vmax2); }
    __syncthreads();
    if (tid < 2u) {
        uint32_t m = 0;
#pragma unroll
        for (uint32_t w = 0; w < kWaves; ++w) m = max(m, wave_max[tid][w]);
        uint32_t* dst = level_max + tid * kMaxLevels + level;
        while (__hip_atomic_load(lm_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != lm_token) __builtin_amdgcn_s_sleep(8);
        if (m > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, m);
    }
}

// ---- The fill of the FINE levels (round 6): levels 8..15 of the standard table -- hashed, power-of-two, 2..128 partitions, no run merging --
// deliver 4/5 of the log's entries, and in pm_fill_pair_kernel they carry the register budget of everything the coarse levels need (the run
// merge's f2x/f2y, three indexers, the folded copies): 124 VGPRs = two workgroups per CU for a kernel whose waves wait half of their time.
// This instantiation serves exactly those levels and is written for a THIRD resident workgroup (<= 80 VGPRs, 6 waves per SIMD, 3 x 50 KB
// of LDS): what crosses barrier 1 is not the sample's 24 finished values but what they are made from -- the three interpolation
// fractions, the two gradients, the TV term -- plus ONE word per entry {slot in the partition's run : 13 | table row : 19}; products,
// half rounding and partition keys are formed when the entry is staged (a few dozen VALU instructions per sample in a kernel bound by
// its waits).  Same arithmetic in the same association as pm_entries: the log holds the same bits, the level maxima are the same, the
// accumulate does not know which kernel filled a region.  One level per XCD (level = first_level + (blockIdx.x & 7)): the level's 2 MB
// table slice (TV stencil) and its regions stay in that XCD's L2, as in the paired mapping of the general kernel.
template <int TV>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(N2M_PM_FINE_WAVES, N2M_PM_FINE_WAVES)))
pm_fill_fine_kernel(const float* __restrict__ grad1 /*[L,Bstride]*/, const _Float16* __restrict__ grad2 /*[L,Bstride,2]*/,
                    const float* __restrict__ inputs, TvParams tv, const float* __restrict__ tv_terms, uint32_t B, uint32_t Bstride, BinPlan plan,
                    PmPlan pm, LevelTable lv, bool align_corners, uint32_t interp, uint32_t* __restrict__ level_max /*[2][32]*/,
                    uint32_t* __restrict__ cursors, uint32_t* __restrict__ ovf_cursor, uint16_t* __restrict__ log_rel, uint32_t* __restrict__ log_v1,
                    uint32_t* __restrict__ log_v2, uint32_t* __restrict__ ovf_key, uint32_t* __restrict__ ovf_v1, uint32_t* __restrict__ ovf_v2,
                    float* __restrict__ found_inf, float in_scale, float in_offset, uint32_t groups_x, uint32_t first_level,
                    unsigned long long* __restrict__ lm_ready, unsigned long long lm_token, uint32_t in_level_stride) {
    constexpr uint32_t D = 3, TS = 512u, kWaves = TS / 64u, kEntries = TS * 8u, MP = 128u;
    constexpr uint32_t kLog2P = 31u - __builtin_clz(kPairP);
    constexpr uint32_t kRowBits = 19u, kRowMask = (1u << kRowBits) - 1u;          // rows of a level < 2^19, slots of a tile's run < 2^13
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    if (blockIdx.x == 0u) {                              // (see pm_fill_pair_kernel: workgroup 0 clears, then publishes this launch's token)
        if (threadIdx.x < 2u * kMaxLevels) level_max[threadIdx.x] = 0u;
        for (uint32_t i = threadIdx.x; i < pm.cursors; i += TS) cursors[i] = 0u;
        if (threadIdx.x == 0u) *ovf_cursor = 0u;
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0u) __hip_atomic_store(lm_ready, lm_token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    extern __shared__ __attribute__((aligned(16))) uint32_t pm_lds[];
    uint32_t* stage_v1 = pm_lds;
    uint32_t* stage_v2 = stage_v1 + kEntries;
    uint32_t* stage_e = stage_v2 + kEntries;             // partition << 16 | row in partition
    uint32_t* cnt2 = stage_e + kEntries;                 // [2][MP]
    uint32_t* start = cnt2 + 2u * MP;
    uint32_t* delta = start + MP;
    uint32_t* ovfb = delta + MP;
    __shared__ uint32_t wave_max[2][kWaves], tile_ovf[2];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    const uint32_t level = first_level + (blockIdx.x & 7u), group = blockIdx.x >> 3, n_groups = groups_x;
    if (level >= plan.levels) return;
    const uint32_t parts = plan.parts[level], size = plan.size[level];
    for (uint32_t i = tid; i < 2u * MP; i += TS) cnt2[i] = 0u;
    if (tid < 2u) tile_ovf[tid] = 0u;
    const float scale = lv.scale[level];
    const Indexer<D> ix(size, lv.resolution[level], 0u, align_corners);
    const uint32_t cap = pm.home_cap[level], region0 = pm.home_base[level];
    uint32_t* __restrict__ cur_l = cursors + pm.cur_base[level];
    float vmax1 = 0.0f, vmax2 = 0.0f;
    bool token_seen = false;
    const PairCtx cx = make_pair_ctx(tv, tv.table ? tv.table + (size_t)plan.row0[level] * tv.stride : nullptr, scale, lv.resolution[level], align_corners, interp);

    uint32_t tile = group;
    float nx[D] = {2.f, 2.f, 2.f}, ng1 = 0.0f, ntv = 0.0f;
    h2 ng2 = {(_Float16)0, (_Float16)0};
    bool nvalid = false;
    auto request = [&](uint32_t t) {                     // the inputs of tile t, RAW (the affine map is applied when they are consumed)
        const uint32_t s = t * TS + tid;
        nvalid = t < plan.tiles && s < B;
        if (nvalid) {
            load_point<D>(inputs + (size_t)level * in_level_stride, s, nx);
            ng1 = grad1 ? grad1[(size_t)level * Bstride + s] : 0.0f;
            if (grad2) ng2 = *reinterpret_cast<const h2*>(grad2 + ((size_t)level * Bstride + s) * 2);
            else ng2 = h2{(_Float16)0, (_Float16)0};
            if (TV == 2) ntv = tv_terms[(size_t)level * Bstride + s];
        }
    };
    request(tile);
    __syncthreads();

    for (uint32_t it = 0; tile < plan.tiles; tile += n_groups, ++it) {
        uint32_t* cnt = cnt2 + (it & 1u) * MP;
        uint32_t* cnt_next = cnt2 + ((it & 1u) ^ 1u) * MP;
        float x[D];
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) x[d] = nvalid ? nx[d] * in_scale + in_offset : 2.0f;
        const float g1 = ng1, tvg = ntv;
        const h2 g2 = ng2;
        request(tile + n_groups);

        // ---- what the sample's entries are made from
        uint32_t e[8];                                   // slot << 19 | row (valid where vmask has the bit)
        float frac[D] = {0.f, 0.f, 0.f}, tvv = 0.0f;
        uint32_t vmask = 0u;
        const float g2x = (float)g2.x, g2y = (float)g2.y;
#pragma unroll
        for (uint32_t c = 0; c < 8; ++c) e[c] = 0u;
        if (!outside_unit_cube<D>(x)) {
            const float a1 = fabsf(g1), a2 = fmaxf(fabsf(g2x), fabsf(g2y));
            vmax1 = fmaxf(vmax1, a1 <= 3.0e38f ? a1 : 1.0f);
            vmax2 = fmaxf(vmax2, a2 <= 3.0e38f ? a2 : 1.0f);
            const bool bad2 = !(fabsf(g2x) <= 3.0e38f) || !(fabsf(g2y) <= 3.0e38f);
            if ((!(a1 <= 3.0e38f) || bad2) && found_inf) *found_inf = 1.0f;
            if (bad2) vmax2 = fmaxf(vmax2, 1.0f);
            const bool gnz = (g1 != 0.0f) | ((__builtin_bit_cast(uint32_t, g2) & 0x7FFF7FFFu) != 0u);
            uint32_t cell[D];
#pragma unroll
            for (uint32_t d = 0; d < D; ++d) {
                const float p = x[d] * scale + (align_corners ? 0.0f : 0.5f);
                cell[d] = (uint32_t)floorf(p);
                frac[d] = p - (float)cell[d];
            }
            if (interp == 1) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (uint32_t d = 0; d < D; ++d) frac[d] = frac[d] * frac[d] * (3.0f - 2.0f * frac[d]);
            }
            const uint32_t sy = kPrimes[1], sz = kPrimes[2];
            const uint32_t tx[2] = {cell[0], cell[0] + 1u};
            const uint32_t ty0 = cell[1] * sy, tz0 = cell[2] * sz;
            const uint32_t ty[2] = {ty0, ty0 + sy}, tz[2] = {tz0, tz0 + sz};
#pragma unroll
            for (uint32_t c = 0; c < 8; ++c) e[c] = (tx[c & 1u] ^ ty[(c >> 1) & 1u] ^ tz[c >> 2]) & ix.mask;
            if constexpr (TV != 0) {
                if constexpr (TV == 1) tvv = pair_tv_value<1>(cx, ix, x, cell, e, tx, ty, tz, sy, sz);
                else tvv = tvg;
                const float a = fabsf(tvv);
                vmax1 = fmaxf(vmax1, (a1 <= 3.0e38f ? a1 : 1.0f) + (a <= 3.0e38f ? a : 1.0f));
            }
            // a sample with a gradient delivers its eight entries, one without at most its TV term on vertex 000
            vmask = gnz ? 0xFFu : (tvv != 0.0f ? 1u : 0u);
        }
        // slot of every entry inside its partition's run of this tile
#pragma unroll
        for (uint32_t c = 0; c < 8; ++c)
            if ((vmask >> c) & 1u) e[c] |= atomicAdd(&cnt[e[c] >> kLog2P], 1u) << kRowBits;
        __syncthreads();                                                     // (1) counters complete

        // run starts: every wave scans the <= 128 counters itself; wave 0 also reserves the runs' room in the partitions' regions
        uint32_t total = 0, q0 = 0u, q1 = 0u, ra0 = 0u, ra1 = 0u, rs0 = 0u, rs1 = 0u;
        {
            const uint32_t i0 = 2u * lane, i1 = i0 + 1u;
            const uint32_t a0 = i0 < parts ? cnt[i0] : 0u, a1c = i1 < parts ? cnt[i1] : 0u;
            const uint32_t incl = n2m_wave_scan_add_u32(a0 + a1c, (int)lane);
            const uint32_t s0 = incl - (a0 + a1c), s1 = s0 + a0;
            if (i0 < parts) start[i0] = s0;
            if (i1 < parts) start[i1] = s1;
            total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (wid == 0u) {
                if (!token_seen) {
                    while (__hip_atomic_load(lm_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != lm_token) __builtin_amdgcn_s_sleep(8);
                    token_seen = true;
                }
                ra0 = a0; ra1 = a1c; rs0 = s0; rs1 = s1;
                if (a0) q0 = atomicAdd(cur_l + i0, a0);
                if (a1c) q1 = atomicAdd(cur_l + i1, a1c);
            }
        }
        {   // staging: the entries are FORMED here (pm_entries' arithmetic, association and rounding points)
            uint32_t pos[8];
#pragma unroll
            for (uint32_t c = 0; c < 8; ++c) pos[c] = start[(e[c] & kRowMask) >> kLog2P];      // (all eight lookups in flight)
            const float wx[2] = {1 - frac[0], frac[0]}, wy[2] = {1 - frac[1], frac[1]}, wz[2] = {1 - frac[2], frac[2]};
#pragma unroll
            for (uint32_t c = 0; c < 8; ++c) {
                if ((vmask >> c) & 1u) {
                    const uint32_t row = e[c] & kRowMask, at = pos[c] + (e[c] >> kRowBits);
                    const float w = (wx[c & 1u] * wy[(c >> 1) & 1u]) * wz[c >> 2];                  // forward's association
                    float p1 = w * g1;
                    if (TV != 0 && c == 0) p1 += tvv;
                    float pa = w * g2x, pb = w * g2y;                                               // rounded to fp32, THEN to half (gridencoder.cu:326)
                    asm volatile("" : "+v"(pa), "+v"(pb));
                    h2 hp;
                    hp.x = (_Float16)pa;
                    hp.y = (_Float16)pb;
                    stage_v1[at] = __float_as_uint(p1);
                    stage_v2[at] = __builtin_bit_cast(uint32_t, hp);
                    stage_e[at] = row + (row >> kLog2P) * (65536u - (1u << kLog2P));                // (row >> 12) << 16 | row & 4095
                }
            }
        }
        if (tid < parts) cnt_next[tid] = 0u;
        if (tid == 0u) tile_ovf[(it & 1u) ^ 1u] = 0u;
        if (wid == 0u) {
            bool ovf_here = false;
            auto finish = [&](uint32_t i, uint32_t a, uint32_t s, uint32_t q) {
                if (a == 0u) return;
                delta[i] = region0 + i * cap + q - s;
                if (q + a > cap) {
                    ovfb[i] = atomicAdd(ovf_cursor, min(a, q + a - cap));
                    ovf_here = true;
                }
            };
            finish(2u * lane, ra0, rs0, q0);
            finish(2u * lane + 1u, ra1, rs1, q1);
            if (__ballot(ovf_here) != 0ull && lane == 0u) tile_ovf[it & 1u] = 1u;
        }
        __syncthreads();                                                     // (2) tile staged, regions reserved

        if (tile_ovf[it & 1u] == 0u) {
            uint32_t ce[8], cv1[8], cv2[8], cd[8];
#pragma unroll
            for (uint32_t q = 0; q < 8; ++q) {
                const uint32_t i = tid + q * TS;
                ce[q] = 0u; cv1[q] = 0u; cv2[q] = 0u;
                if (i < total) { ce[q] = stage_e[i]; cv1[q] = stage_v1[i]; cv2[q] = stage_v2[i]; }
            }
#pragma unroll
            for (uint32_t q = 0; q < 8; ++q) cd[q] = delta[ce[q] >> 16] + tid + q * TS;
#pragma unroll
            for (uint32_t q = 0; q < 8; ++q)
                if (tid + q * TS < total) {
                    if (log_v1) *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(log_v1) + (cd[q] << 2)) = cv1[q];
                    if (log_v2) *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(log_v2) + (cd[q] << 2)) = cv2[q];
                    *reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(log_rel) + (cd[q] << 1)) = (uint16_t)ce[q];
                }
        } else {
            for (uint32_t i = tid; i < total; i += TS) {
                const uint32_t ee = stage_e[i], p = ee >> 16;
                const uint32_t st = start[p];
                const uint32_t qr = delta[p] + st - (region0 + p * cap);
                const uint32_t qi = qr + (i - st);
                if (qi < cap) {
                    const uint32_t d = delta[p] + i;
                    if (log_v1) log_v1[d] = stage_v1[i];
                    if (log_v2) log_v2[d] = stage_v2[i];
                    log_rel[d] = (uint16_t)ee;
                } else {
                    const uint32_t o = ovfb[p] + qi - max(qr, cap);
                    if (o < pm.ovf_cap) {
                        ovf_key[o] = (level << 27) | ee;
                        if (ovf_v1) ovf_v1[o] = stage_v1[i];
                        if (ovf_v2) ovf_v2[o] = stage_v2[i];
                    }
                }
            }
        }
        // no barrier here: the next tile writes the stage / start / delta only after its barrier (1)
    }

#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        vmax1 = fmaxf(vmax1, __shfl_xor(vmax1, o, 64));
        vmax2 = fmaxf(vmax2, __shfl_xor(vmax2, o, 64));
    }
    if (lane == 0) { wave_max[0][wid] = __float_as_uint(vmax1); wave_max[1][wid] = __float_as_uint(vmax2); }
    __syncthreads();
    if (tid < 2u) {
        uint32_t m = 0;
#pragma unroll
        for (uint32_t w = 0; w < kWaves; ++w) m = max(m, wave_max[tid][w]);
        uint32_t* dst = level_max + tid * kMaxLevels + level;
        while (__hip_atomic_load(lm_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != lm_token) __builtin_amdgcn_s_sleep(8);
        if (m > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, m);
    }
}

// One work item = (level, SUB adjacent partitions[, slice grp of Gl]): streams the partitions' regions (contiguous arrays) into the
// 64-bit fixed-point LDS accumulator, then flushes like bin_accumulate_kernel.
// Levels whose partitions are split over Gl > 1 work items (the small dense levels; every level of a batch above 2^19 samples): the
// slices of a region are cut by entry index, i.e. by the order in which the fill's reservations happened to arrive, so their partial sums
// differ from run to run.  They are therefore never added as floats: every slice leaves its accumulator -- integers -- in a scratch
// slot, the last of the Gl items to arrive (a ticket per partition group) adds the slots and flushes the rows with plain stores.  Integer
// sums are exact whatever the cut: the whole table backward is bit-reproducible run to run (the tile-major path ends these levels in
// float atomics), and the fill no longer clears rows for atomics to land on.
// Peer-store routing of the flush (include/n2m_peer.h): the row's owner keeps a staging slot for this rank; world == 0: off.
struct PeerRouteK { uint32_t world, split, rows_c, rows_f; };
struct PeerRouteT { void* base[2][N2M_PEER_MAX]; };
template <typename T, uint32_t C>
__device__ __forceinline__ T* peer_row(const PeerRouteK& k, const PeerRouteT& t, uint32_t abs_row) {
    const uint32_t half = abs_row >= k.split ? 1u : 0u, rel = abs_row - (half ? k.split : 0u), n = half ? k.rows_f : k.rows_c;
    const uint32_t owner = rel / n;
    return reinterpret_cast<T*>(t.base[half][owner]) + (size_t)(rel - owner * n) * C;
}

struct PmSplit {
    uint32_t slot0[kMaxLevels];         // first scratch slot of the level's work items (slot = SUB * P * C u64; levels with Gl > 1 only)
    uint32_t tick0[kMaxLevels];         // first ticket of the level's partition groups
    uint32_t slice_log2;                // entries per slice: 2^16 (C = 2: 2^15) when the launch has plenty of partition groups, 2^13 when it has few
};

template <typename T, uint32_t C, uint32_t P, uint32_t SUB>
__device__ __forceinline__ void pm_accumulate_items(uint32_t first_item, uint32_t item_stride, T* __restrict__ grad_table, const BinPlan& plan,
                     const PmPlan& pm, const PmSplit& sp, const LevelTable& lv, uint32_t gridtype, bool align_corners,
                     const uint32_t* __restrict__ level_max, const uint32_t* __restrict__ cursors, const uint32_t* __restrict__ ovf_cursor,
                     const uint16_t* __restrict__ log_rel, const uint32_t* __restrict__ log_val, const uint32_t* __restrict__ ovf_key,
                     const uint32_t* __restrict__ ovf_val, unsigned long long* __restrict__ slots, uint32_t* __restrict__ tickets,
                     float* __restrict__ found_inf, bool overwrite, uint32_t dbg, float inf_bound, const PeerRouteK& prk, const PeerRouteT& prt) {
    constexpr uint32_t kLog2P = 31u - __builtin_clz(P);
    constexpr uint32_t kSlot = SUB * P * C;
    extern __shared__ __attribute__((aligned(16))) unsigned long long bin_acc[];   // SUB * P * C
    __shared__ uint32_t nonfinite_seen, arrival;
    const uint32_t tid = threadIdx.x;
    const uint32_t total_items = plan.item_prefix[plan.levels];
    for (uint32_t item = first_item; item < total_items; item += item_stride) {
        uint32_t level = 0;
        while (item >= plan.item_prefix[level + 1]) ++level;
        const uint32_t vm_v = __hip_atomic_load(level_max + level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t Gplan = plan.groups[level];          // slices the launch provides per partition group (sized for unmerged entries)
        if (!overwrite && (uint32_t)__builtin_amdgcn_readfirstlane((int)vm_v) == 0u) continue;     // no non-zero finite update in this level
        const uint32_t local = item - plan.item_prefix[level];
        const uint32_t ip = local / Gplan, part0 = ip * SUB, grp = local - ip * Gplan;
        const uint32_t parts = plan.parts[level], size = plan.size[level];
        const uint32_t part_end = min(part0 + SUB, parts);
        const Indexer<3> ix(size, lv.resolution[level], gridtype, align_corners);
        const bool interleaved = !ix.hashed && parts > 1u;
        const uint32_t row0 = plan.row0[level];
        const uint32_t n_blocks = (size + 15u) >> 4;
        auto rows_of = [&](uint32_t part) {
            const uint32_t my_blocks = interleaved ? (part < n_blocks ? (n_blocks - part + parts - 1) / parts : 0u)
                                                   : min(P / 16u, n_blocks - min(n_blocks, part * (P / 16u)));
            return my_blocks << 4;
        };
        auto global_row = [&](uint32_t u, uint32_t rel) {
            return interleaved ? ((((rel >> 4) * parts + part0 + u) << 4) | (rel & 15u)) : (((part0 + u) << kLog2P) + rel);
        };
        T* __restrict__ gtab = grad_table + (size_t)row0 * C;
        // the cursors through the vector memory path, requested before the accumulator is cleared
        uint32_t n_all[SUB];
#pragma unroll
        for (uint32_t u = 0; u < SUB; ++u)
            n_all[u] = part0 + u < part_end ? __hip_atomic_load(cursors + pm.cur_base[level] + part0 + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        // Slices actually used: one per 2^16 (C = 2: 2^15) entries the regions HOLD.  Same-cell runs merged by the fill leave the coarse levels of a
        // marched batch with a sixteenth of the entries the launch had to provide for; every slice costs a 64 KB slot round trip, and
        // 135 of them took this kernel from 37 to 93 us.  The items beyond Gl find nothing to do.
        uint32_t n_tot = 0;
#pragma unroll
        for (uint32_t u = 0; u < SUB; ++u) n_tot += min((uint32_t)__builtin_amdgcn_readfirstlane((int)n_all[u]), pm.home_cap[level]);
        // (a slice = at most twice what an item of the hashed levels holds at 2^18 samples; a call over a few dense levels -- the SDF recipe's
        //  early schedule: 4 levels, 18 partition groups, 1.5 M stacked samples -- gets 2^13-entry slices, or 200 items would serve 256 CUs)
        const uint32_t kSliceLog2 = sp.slice_log2;
        const uint32_t Gl = min(Gplan, max(1u, (n_tot + (1u << kSliceLog2) - 1u) >> kSliceLog2));
        if (grp >= Gl) continue;
        for (uint32_t i = tid; i < kSlot; i += 1024) bin_acc[i] = 0ull;
        if (tid == 0) nonfinite_seen = 0u;
        __syncthreads();
        const uint32_t vm = (uint32_t)__builtin_amdgcn_readfirstlane((int)vm_v);
        int ex = 37 - ((int)((vm >> 23) & 255u) - 127);
        ex = ex < -126 ? -126 : (ex > 126 ? 126 : ex);
        const float scale = __uint_as_float((uint32_t)(ex + 127) << 23);
        const float inv = __uint_as_float((uint32_t)(127 - ex) << 23);
        const uint32_t cap = pm.home_cap[level];
        // walk(body, whole): body(rel0, u, bits) for every entry of this item's slice (whole: of the complete regions) + overflow records
        auto walk = [&](auto&& body, bool whole) {
            if (vm == 0u) return;
#pragma unroll
            for (uint32_t u = 0; u < SUB; ++u) {
                const uint32_t nu = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_all[u]);
                const uint32_t n = min(nu, cap);
                // slice of this group: boundaries on multiples of 4 entries (the last slice ends at n)
                const uint32_t lo = (Gl == 1u || whole) ? 0u : (uint32_t)(((uint64_t)n * grp / Gl) & ~3ull);
                const uint32_t hi = (Gl == 1u || whole || grp + 1u == Gl) ? n : (uint32_t)(((uint64_t)n * (grp + 1u) / Gl) & ~3ull);
                const size_t base = (size_t)pm.home_base[level] + (size_t)(part0 + u) * cap;
                const uint16_t* __restrict__ rp = log_rel + base;
                const uint32_t* __restrict__ vp = log_val + base;
                for (uint32_t i0 = lo + tid; i0 < hi; i0 += 4096u) {
                    uint32_t r[4], v[4];
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) {
                        const uint32_t i = i0 + k * 1024u;
                        r[k] = 0u; v[k] = 0u;
                        if (i < hi) { r[k] = rp[i]; v[k] = vp[i]; }
                    }
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k)
                        if (i0 + k * 1024u < hi) body(r[k], u, v[k]);
                }
                if (nu > cap && (grp == 0u || whole)) {  // records of this partition in the shared overflow log
                    const uint32_t n_ovf = min(__hip_atomic_load(ovf_cursor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), pm.ovf_cap);
                    const uint32_t want = (level << 11) | (part0 + u);
                    for (uint32_t i = tid; i < n_ovf; i += 1024u) {
                        const uint32_t k = ovf_key[i];
                        if ((k >> 16) == want) body(k & 0xFFFFu, u, ovf_val[i]);
                    }
                }
            }
        };
        // inf / nan entries bypass the fixed-point sum and go to the table as they are -- AFTER the rows have been stored: the first walk
        // only notes that there are any, a second walk (never taken in a healthy run) adds them onto the stored sums
        auto bypass = [&](uint32_t rel0, uint32_t u, uint32_t bits) {
            if constexpr (sizeof(T) == 4) unsafeAtomicAdd(gtab + global_row(u, rel0), __uint_as_float(bits));
            else {
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                (void)__builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) h2*)(gtab + (size_t)global_row(u, rel0) * 2u),
                                                                __builtin_bit_cast(h2, bits));
            }
        };
        auto finite = [&](uint32_t bits) {
            if constexpr (sizeof(T) == 4) return fabsf(__uint_as_float(bits)) <= 3.0e38f;
            else {
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const h2 p = __builtin_bit_cast(h2, bits);
                return fabsf((float)p.x) <= 3.0e38f && fabsf((float)p.y) <= 3.0e38f;
            }
        };
        walk([&](uint32_t rel0, uint32_t u, uint32_t bits) {
            const uint32_t rel = rel0 + u * P;
            if (!finite(bits)) { nonfinite_seen = 1u; return; }
            if constexpr (sizeof(T) == 4) {
                if (bits << 1) __hip_atomic_fetch_add(&bin_acc[rel], (unsigned long long)to_fixed(__uint_as_float(bits), scale), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const h2 p = __builtin_bit_cast(h2, bits);
                const float v0 = (float)p.x, v1 = (float)p.y;
                if (v0 != 0.f) __hip_atomic_fetch_add(&bin_acc[rel * 2u], (unsigned long long)to_fixed(v0, scale), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (v1 != 0.f) __hip_atomic_fetch_add(&bin_acc[rel * 2u + 1u], (unsigned long long)to_fixed(v1, scale), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }, false);
        __syncthreads();
        bool second_walk = nonfinite_seen != 0u;
        if (Gl > 1u) {
            // leave the slice's sums in its slot; the last of the group's items to arrive adds all slots
            unsigned long long* __restrict__ mine = slots + (size_t)(sp.slot0[level] + local) * kSlot;
            uint32_t* __restrict__ ticket = tickets + sp.tick0[level] + ip;
            // Hand-over between workgroups WITHOUT cache-wide fences (a release / acquire fence at agent scope writes back and invalidates a
            // whole L2: 16 waves x 70 items of that turned 90 us of accumulates into 470): the slots are written and read with agent-scope
            // (write-through / cache-bypassing) accesses, every wave waits for its own stores to be acknowledged, then the barrier, then
            // the ticket -- whoever draws the last ticket finds all slots in memory.
            for (uint32_t i = tid; i < kSlot; i += 1024) __hip_atomic_store(&mine[i], bin_acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0x0F70);                              // vmcnt(0): this wave's slot stores have reached memory
            __syncthreads();
            if (tid == 0) arrival = __hip_atomic_fetch_add(ticket, second_walk ? 0x10001u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (high half: slices with inf / nan entries)
            __syncthreads();
            const uint32_t arr = arrival;
            if ((arr & 0xFFFFu) != Gl - 1u) continue;                        // (uniform) not the last one: the next item's clear follows its own barrier
            second_walk = second_walk || (arr >> 16) != 0u;
            if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next call
            unsigned long long* __restrict__ first = slots + (size_t)(sp.slot0[level] + ip * Gplan) * kSlot;
            {   // all of a slot's loads of this thread in flight together (they bypass the caches: one memory round trip per slot, not per value)
                unsigned long long sum[kSlot / 1024u];
#pragma unroll
                for (uint32_t k = 0; k < kSlot / 1024u; ++k) sum[k] = 0ull;
                for (uint32_t g = 0; g < Gl; ++g) {
#pragma unroll
                    for (uint32_t k = 0; k < kSlot / 1024u; ++k)
                        sum[k] += __hip_atomic_load(first + (size_t)g * kSlot + tid + k * 1024u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (uint32_t k = 0; k < kSlot / 1024u; ++k) bin_acc[tid + k * 1024u] = sum[k];
            }
            __syncthreads();
        }
        for (uint32_t rel = tid; rel < ((dbg & 2u) ? 0u : SUB * P); rel += 1024) {
            const uint32_t u = rel >> kLog2P, rel0 = rel & (P - 1u);
            if (part0 + u >= part_end || rel0 >= rows_of(part0 + u)) continue;
            const uint32_t row = global_row(u, rel0);
            if (row >= size) continue;
            if constexpr (sizeof(T) == 4) {
                const long long a = (long long)bin_acc[rel];
                if (a != 0 || overwrite) {
                    const float f = (float)a * inv;
                    if (!(fabsf(f) <= (inf_bound > 0.0f ? inf_bound : 3.0e38f)) && found_inf) *found_inf = 1.0f;
                    float* dst = prk.world ? peer_row<float, 1>(prk, prt, row0 + row) : gtab + row;
                    if (overwrite) *dst = f;
                    else *dst += f;
                }
            } else {
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const long long a0 = (long long)bin_acc[rel * 2u], a1 = (long long)bin_acc[rel * 2u + 1u];
                if ((a0 | a1) != 0 || overwrite) {
                    const float f0 = (float)a0 * inv, f1 = (float)a1 * inv;
                    const float b16 = inf_bound > 0.0f ? inf_bound : 65504.0f;
                    if (!(fabsf(f0) <= b16 && fabsf(f1) <= b16) && found_inf) *found_inf = 1.0f;       // rounds to inf in fp16 (or could, summed over ranks)
                    h2* dst = reinterpret_cast<h2*>(prk.world ? peer_row<_Float16, 2>(prk, prt, row0 + row) : gtab + (size_t)row * 2u);
                    h2 o;
                    if (overwrite) {
                        o.x = (_Float16)f0;          // == (half)(0 + f): the sum below starts from +0
                        o.y = (_Float16)f1;
                    } else {
                        o = *dst;
                        o.x = (_Float16)((float)o.x + f0);
                        o.y = (_Float16)((float)o.y + f1);
                    }
                    *dst = o;
                }
            }
        }
        __syncthreads();
        if (second_walk && prk.world) {
            if (tid == 0 && found_inf) *found_inf = 1.0f;                  // routed rows live on another rank: no atomics across the link, the step is skipped anyway
        } else if (second_walk) {
            __threadfence();                                               // the stores above before the atomics below
            walk([&](uint32_t rel0, uint32_t u, uint32_t bits) { if (!finite(bits)) bypass(rel0, u, bits); }, Gl > 1u);
            __syncthreads();
        }
    }
}

// One table alone / both tables in ONE launch: workgroups [0, nb1) walk the fp32 table's items, the others the fp16 table's (own tickets, own
// scratch slots).  Two launches meant a kernel boundary (~5 us) and two tails -- 1 494 items on 512 workgroup slots each.
template <typename T, uint32_t C, uint32_t P, uint32_t SUB>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8)))
pm_accumulate_kernel(T* __restrict__ grad_table, BinPlan plan, PmPlan pm, PmSplit sp, LevelTable lv, uint32_t gridtype, bool align_corners,
                     const uint32_t* __restrict__ level_max, const uint32_t* __restrict__ cursors, const uint32_t* __restrict__ ovf_cursor,
                     const uint16_t* __restrict__ log_rel, const uint32_t* __restrict__ log_val, const uint32_t* __restrict__ ovf_key,
                     const uint32_t* __restrict__ ovf_val, unsigned long long* __restrict__ slots, uint32_t* __restrict__ tickets,
                     float* __restrict__ found_inf, bool overwrite, uint32_t dbg, float inf_bound, PeerRouteK prk, PeerRouteT prt) {
    pm_accumulate_items<T, C, P, SUB>(blockIdx.x, gridDim.x, grad_table, plan, pm, sp, lv, gridtype, align_corners, level_max, cursors, ovf_cursor, log_rel, log_val,
                                      ovf_key, ovf_val, slots, tickets, found_inf, overwrite, dbg, inf_bound, prk, prt);
}

struct PmBoth {
    BinPlan plan1, plan2;
    PmSplit sp1, sp2;
};
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8)))
pm_accumulate_both_kernel(float* __restrict__ table1, _Float16* __restrict__ table2, PmBoth pb, PmPlan pm, LevelTable lv, uint32_t gridtype, bool align_corners,
                          const uint32_t* __restrict__ level_max, const uint32_t* __restrict__ cursors, const uint32_t* __restrict__ ovf_cursor,
                          const uint16_t* __restrict__ log_rel, const uint32_t* __restrict__ log_v1, const uint32_t* __restrict__ log_v2,
                          const uint32_t* __restrict__ ovf_key, const uint32_t* __restrict__ ovf_v1, const uint32_t* __restrict__ ovf_v2,
                          unsigned long long* __restrict__ slots1, unsigned long long* __restrict__ slots2, uint32_t* __restrict__ tickets1,
                          uint32_t* __restrict__ tickets2, float* __restrict__ found_inf, bool overwrite, uint32_t dbg, float bound1, float bound2, uint32_t nb1,
                          PeerRouteK prk, PeerRouteT prt1, PeerRouteT prt2) {
    if (blockIdx.x < nb1)
        pm_accumulate_items<float, 1, kPairP, 2>(blockIdx.x, nb1, table1, pb.plan1, pm, pb.sp1, lv, gridtype, align_corners, level_max, cursors, ovf_cursor, log_rel,
                                                 log_v1, ovf_key, ovf_v1, slots1, tickets1, found_inf, overwrite, dbg, bound1, prk, prt1);
    else
        pm_accumulate_items<_Float16, 2, kPairP, 1>(blockIdx.x - nb1, gridDim.x - nb1, table2, pb.plan2, pm, pb.sp2, lv, gridtype, align_corners, level_max + kMaxLevels,
                                                    cursors, ovf_cursor, log_rel, log_v2, ovf_key, ovf_v2, slots2, tickets2, found_inf, overwrite, dbg, bound2, prk, prt2);
}

// grad_inputs[b,d] = sum_{l,c} grad[l,b,c] * dy_dx[b,l,d,c]; accumulates in T like the reference (:357-365)
template <typename T, uint32_t D, uint32_t C>
__global__ void grid_input_backward_kernel(const T* __restrict__ grad, const T* __restrict__ dy_dx, T* __restrict__ grad_inputs,
                                           uint32_t B, uint32_t L) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const T* j = dy_dx + (size_t)b * L * D * C + (size_t)d * C;
    T acc = (T)0;
    for (uint32_t l = 0; l < L; ++l) {
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) {
            const T gv = grad[((size_t)l * B + b) * C + c], jv = j[(size_t)l * D * C + c];
            if constexpr (sizeof(T) == 2) {
                const T p = (T)((float)gv * (float)jv);
                acc = (T)((float)acc + (float)p);
            } else {
                acc += gv * jv;
            }
        }
    }
    grad_inputs[t] = acc;
}

// ------------------------------------------------------------------------------------------ total variation
template <uint32_t D, uint32_t C>
__global__ void __launch_bounds__(256)
grid_tv_kernel(const float* __restrict__ inputs, const float* __restrict__ table, float* __restrict__ grad,
               const int32_t* __restrict__ offsets, float weight, uint32_t B, LevelTable lv, uint32_t gridtype,
               bool align_corners) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    const uint32_t level = blockIdx.y;
    if (b >= B) return;
    const uint32_t row0 = (uint32_t)offsets[level];
    const uint32_t size = (uint32_t)offsets[level + 1] - row0;
    const uint32_t resolution = lv.resolution[level];
    const float scale = lv.scale[level];
    const Indexer<D> ix(size, resolution, gridtype, align_corners);
    const float* __restrict__ tab = table + (size_t)row0 * C;
    float* gtab = grad + (size_t)row0 * C;

    float x[D];
    load_point<D>(inputs, b, x);
    if (outside_unit_cube<D>(x)) return;
    uint32_t cell[D];
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) cell[d] = (uint32_t)floorf(x[d] * scale + (align_corners ? 0.0f : 0.5f));

    const uint32_t here = ix.row(cell);
    const Row<float, C> centre = Row<float, C>::load(tab + (size_t)here * C);
    float sum[C], sq[C];
#pragma unroll
    for (uint32_t c = 0; c < C; ++c) { sum[c] = 0.f; sq[c] = 0.f; }
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) {
        const uint32_t cur = cell[d];
        if (cur < resolution) {
            cell[d] = cur + 1;
            const Row<float, C> nb = Row<float, C>::load(tab + (size_t)ix.row(cell) * C);
#pragma unroll
            for (uint32_t c = 0; c < C; ++c) { const float dv = centre.v[c] - nb.v[c]; sum[c] += dv; sq[c] += dv * dv; }
        }
        if (cur > 0) {
            cell[d] = cur - 1;
            const Row<float, C> nb = Row<float, C>::load(tab + (size_t)ix.row(cell) * C);
#pragma unroll
            for (uint32_t c = 0; c < C; ++c) { const float dv = centre.v[c] - nb.v[c]; sum[c] += dv; sq[c] += dv * dv; }
        }
        cell[d] = cur;
    }
    const float w = weight / (float)(2 * D);
#pragma unroll
    for (uint32_t c = 0; c < C; ++c) unsafeAtomicAdd(gtab + (size_t)here * C + c, w * sum[c] * (1.0f / sqrtf(sq[c] + 1e-9f)));
}

// ------------------------------------------------------------------------------------------------ dispatch

struct FwdArgs {
    const float* inputs; const void* table; const int32_t* offsets; void* outputs;
    uint32_t B, L, max_level; LevelTable lv; void* dy_dx; uint32_t gridtype; bool align; uint32_t interp; hipStream_t s;
};

template <typename T, uint32_t D, uint32_t C, bool BM>
void launch_forward(const FwdArgs& a) {
    static const bool generic_only = getenv("N2M_GRID_FWD_GENERIC") != nullptr;   // A/B switch
    if constexpr (D == 3) {
        if (!generic_only && a.dy_dx == nullptr) {
            static const bool xcd_map = getenv("N2M_GRID_FWD_XCD") != nullptr;   // measured slower: off by default
            const uint32_t n_tiles = n2m_ceil_div(a.B, 256);
            const uint32_t slots = (a.max_level + 7u) / 8u;
            const uint32_t nblk = xcd_map ? n_tiles * slots * 8u : n_tiles * a.max_level;
            grid_forward3_kernel<T, C, BM><<<nblk, 256, 0, a.s>>>(a.inputs, (const T*)a.table, a.offsets, (T*)a.outputs, a.B, a.L,
                                                                  a.max_level, a.lv, a.gridtype, a.align, a.interp, n_tiles, xcd_map);
            return;
        }
    }
    const dim3 grid(n2m_ceil_div(a.B, 256), a.max_level);
    grid_forward_kernel<T, D, C, BM><<<grid, 256, 0, a.s>>>(a.inputs, (const T*)a.table, a.offsets, (T*)a.outputs, a.B, a.L, a.lv,
                                                            (T*)a.dy_dx, a.gridtype, a.align, a.interp);
}

struct BwdArgs {
    const void* grad; const float* inputs; const int32_t* offsets; void* grad_table;
    uint32_t B, L, max_level; LevelTable lv; uint32_t gridtype; bool align; uint32_t interp; hipStream_t s;
    const void* dy_dx; void* grad_inputs;
};

template <typename T, uint32_t D, uint32_t C, bool BM>
void launch_backward(const BwdArgs& a) {
    static const bool use_scatter = getenv("N2M_GRID_BWD_SCATTER") != nullptr;   // A/B switch for measurements
    if (D == 3 && !use_scatter && !(sizeof(T) == 2 && (C & 1u))) {
        if constexpr (D == 3) {
            static bool attr_set = false;
            if (!attr_set) {
                (void)hipFuncSetAttribute((const void*)grid_backward_lds_kernel<T, C, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)kLdsBytes);
                attr_set = true;
            }
            const T* g = (const T*)a.grad;
            T* scratch = nullptr;
            if (BM) {   // sample-major producer: transpose to level-major first
                const size_t n = (size_t)a.B * a.L * C;
                if (hipMallocAsync((void**)&scratch, n * sizeof(T), a.s) != hipSuccess) scratch = nullptr;
                if (scratch) {
                    transpose_to_level_major_kernel<T><<<n2m_ceil_div(n, 256), 256, 0, a.s>>>((const T*)a.grad, scratch, a.B, a.L, C);
                    g = scratch;
                }
            }
            if (!BM || scratch) {
                const uint32_t G = a.B >= (1u << 18) ? 4u : a.B >= (1u << 16) ? 2u : 1u;
                grid_backward_lds_kernel<T, C, false><<<2048, 1024, kLdsBytes, a.s>>>(g, a.inputs, a.offsets, (T*)a.grad_table, a.B,
                                                                                       a.max_level, a.lv, a.gridtype, a.align, a.interp, G,
                                                                                       nullptr, 0.0f);
                if (scratch) (void)hipFreeAsync(scratch, a.s);
                if (!BM && a.dy_dx && a.grad_inputs)
                    grid_input_backward_kernel<T, D, C><<<n2m_ceil_div((uint64_t)a.B * D, 256), 256, 0, a.s>>>(
                        (const T*)a.grad, (const T*)a.dy_dx, (T*)a.grad_inputs, a.B, a.L);
                return;
            }
        }
    }
    const dim3 grid(n2m_ceil_div(a.B, 256), a.max_level);
    grid_backward_kernel<T, D, C, BM><<<grid, 256, 0, a.s>>>((const T*)a.grad, a.inputs, a.offsets, (T*)a.grad_table, a.B, a.L,
                                                             a.lv, a.gridtype, a.align, a.interp);
    if (!BM && a.dy_dx && a.grad_inputs)
        grid_input_backward_kernel<T, D, C><<<n2m_ceil_div((uint64_t)a.B * D, 256), 256, 0, a.s>>>(
            (const T*)a.grad, (const T*)a.dy_dx, (T*)a.grad_inputs, a.B, a.L);
}

struct TvArgs {
    const float* inputs; const float* table; float* grad; const int32_t* offsets; float weight;
    uint32_t B, L; LevelTable lv; uint32_t gridtype; bool align; hipStream_t s;
};

template <uint32_t D, uint32_t C>
void launch_tv(const TvArgs& a) {
    static const bool use_scatter = getenv("N2M_GRID_TV_SCATTER") != nullptr;   // A/B switch
    if constexpr (D == 3) {
        if (!use_scatter) {
            static bool attr_set = false;
            if (!attr_set) {
                (void)hipFuncSetAttribute((const void*)grid_backward_lds_kernel<float, C, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)kLdsBytes);
                attr_set = true;
            }
            grid_backward_lds_kernel<float, C, true><<<2048, 1024, kLdsBytes, a.s>>>(nullptr, a.inputs, a.offsets, a.grad, a.B, a.L, a.lv,
                                                                                      a.gridtype, a.align, 0u, 2u, a.table, a.weight);
            return;
        }
    }
    const dim3 grid(n2m_ceil_div(a.B, 256), a.L);
    grid_tv_kernel<D, C><<<grid, 256, 0, a.s>>>(a.inputs, a.table, a.grad, a.offsets, a.weight, a.B, a.lv, a.gridtype, a.align);
}


// ---- binned path: host plan + launches
// Process-wide settings of the binned backward (n2m_grid_backward_config): the row stride of the TV table and the margin of the
// gradient overflow check.  A multi-GPU caller that SUMS gradient tables over W ranks sets the margin to W: a row whose local sum exceeds
// max / W could overflow in the cross-rank sum, so it raises found_inf already (GradScaler then skips and backs off one notch early
// instead of never seeing an overflow that only the reduction produces).
// Settings of the CALLING THREAD's next table-backward calls (n2m_grid_backward_config).  Thread-local since round 4: as process-wide atomics two
// engines stepping on two threads could read each other's values between the setter and the call it precedes; a thread that restates them
// before each call (every caller in nerf2mesh_amd/ does) cannot be raced.
struct ThreadCfg { uint32_t v; uint32_t load() const { return v; } };
struct ThreadCfgF { float v; float load() const { return v; } };
static thread_local ThreadCfg g_cfg_tv_stride{1};
static thread_local ThreadCfgF g_cfg_overflow_div{1.0f};
static thread_local N2mPeerRoute g_peer_route{};          // world == 0: off (n2m_grid_backward_peer_route)

constexpr size_t kBinHeaderBytes = 512;          // [level maxima 2 x 32 words][ready token 8 B][pad]

struct BinLayout {
    BinPlan plan;
    size_t dir_words, log_entries, bytes;
    bool ok;
};

BinLayout make_bin_plan(uint32_t Bc, uint32_t C, uint32_t max_level, const int32_t* host_offsets, bool tv, uint32_t P = 0, uint32_t logs = 1) {
    BinLayout o{};
    o.ok = max_level >= 1 && max_level <= kMaxLevels && (C == 1 || C == 2);
    if (!o.ok) return o;
    if (P == 0) P = kBinAccBytes / (8u * C);
    const uint32_t per_tile = tv ? kTileEntries : 1024u;                     // samples per tile
    const uint32_t tiles = (Bc + per_tile - 1) / per_tile;
    o.plan.tiles = tiles;
    o.plan.levels = max_level;
    size_t dir = 0;
    uint32_t items = 0;
    for (uint32_t l = 0; l < max_level; ++l) {
        const int64_t size = (int64_t)host_offsets[l + 1] - (int64_t)host_offsets[l];
        if (size <= 0 || host_offsets[l] < 0) { o.ok = false; return o; }
        const uint32_t n_blocks = ((uint32_t)size + 15u) >> 4;
        const uint32_t parts = (n_blocks + P / 16u - 1) / (P / 16u);
        if (parts > kMaxPartsPerLevel) { o.ok = false; return o; }
        const uint64_t per_part = (uint64_t)(tv ? 1 : 8) * Bc / parts;       // expected entries of one partition
        uint32_t g = (uint32_t)((per_part + 65535u) / 65536u);
        g = g < 1u ? 1u : (g > 64u ? 64u : g);
        g = g > tiles ? tiles : g;
        o.plan.row0[l] = (uint32_t)host_offsets[l];
        o.plan.size[l] = (uint32_t)size;
        o.plan.parts[l] = parts;
        o.plan.groups[l] = g;
        o.plan.dir_base[l] = (uint32_t)dir;
        o.plan.item_prefix[l] = items;
        dir += (size_t)tiles * (parts + 1u);
        items += parts * g;
    }
    o.plan.item_prefix[max_level] = items;
    o.dir_words = dir;
    o.log_entries = (size_t)max_level * tiles * kTileEntries;
    o.bytes = kBinHeaderBytes + ((dir * 4 + 255) & ~(size_t)255) + (size_t)logs * o.log_entries * 8;
    if (dir >= (1ull << 32)) o.ok = false;
    return o;
}

template <typename T, uint32_t C, int MODE>
int launch_binned(const T* grad, const float* inputs, TvParams tv, T* grad_table, uint32_t B, uint32_t max_level,
                  const int32_t* host_offsets, const LevelTable& lv, uint32_t gridtype, bool align, uint32_t interp, void* workspace,
                  size_t workspace_bytes, hipStream_t s, const char* fn, float* found_inf = nullptr) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)bin_fill_kernel<T, C, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kTileEntries * 8));
        (void)hipFuncSetAttribute((const void*)bin_accumulate_kernel<T, C, BinGeom<C>::P, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBinAccBytes);
        attr_set = true;
    }
    for (uint32_t b0 = 0; b0 < B; b0 += kBinChunk) {
        const uint32_t Bc = B - b0 < kBinChunk ? B - b0 : kBinChunk;
        const BinLayout lay = make_bin_plan(Bc, C, max_level, host_offsets, MODE == 1);
        N2M_REQUIRE(lay.ok, N2M_EUNSUPPORTED, "%s: table layout not supported by the binned path", fn);
        N2M_REQUIRE(workspace_bytes >= lay.bytes, N2M_EINVAL, "%s: workspace too small (%zu < %zu bytes)", fn, workspace_bytes, lay.bytes);
        uint32_t* level_max = (uint32_t*)workspace;
        uint32_t* directory = (uint32_t*)((char*)workspace + 256);
        uint64_t* log = (uint64_t*)((char*)workspace + 256 + ((lay.dir_words * 4 + 255) & ~(size_t)255));
        N2M_HIP(hipMemsetAsync(level_max, 0, 256, s));
        // grad is [L, B, C]: a pass covers the column block b0 .. b0+Bc of every level (level stride = the full B)
        const T* g = grad ? grad + (size_t)b0 * C : nullptr;
        const float* x = inputs + (size_t)b0 * 3;
        bin_fill_kernel<T, C, MODE><<<dim3(lay.plan.tiles, max_level), 1024, kTileEntries * 8, s>>>(g, x, tv, Bc, B, lay.plan, lv, gridtype, align, interp,
                                                                                                     level_max, directory, log, found_inf);
        N2M_CHECK_LAUNCH();
        const uint32_t items = lay.plan.item_prefix[max_level];
        bin_accumulate_kernel<T, C, BinGeom<C>::P, 1><<<items < 2048u ? items : 2048u, 1024, kBinAccBytes, s>>>(
            grad_table, lay.plan, lv, gridtype, align, level_max, directory, log, found_inf, nullptr, nullptr, false, 0u,
            (sizeof(T) == 2 ? 65504.0f : 3.0e38f) / g_cfg_overflow_div.load());
        N2M_CHECK_LAUNCH();
    }
    return 0;
}



// ---- partition-major log: layout and launches (kernels: pm_fill_pair_kernel, pm_accumulate_kernel)
struct PmLayout {
    PmPlan pm;
    size_t home_entries, slot_count, bytes;
    bool ok;
};

// Bc = samples of a pass.  Region capacity = 2 x the partition's expected share of the level's 8 Bc entries + slack, never more than
// all of them (small batches: no overflow possible at all).
PmLayout make_pm_plan(uint32_t Bc, const BinPlan& plan) {
    PmLayout o{};
    o.ok = true;
    uint64_t home = 0;
    uint32_t cur = 0, mp = 1;
    const uint64_t all = ((uint64_t)8 * Bc + 255u) & ~(uint64_t)255u;
    for (uint32_t l = 0; l < plan.levels; ++l) {
        const uint32_t parts = plan.parts[l];
        uint64_t cap = (2u * (((uint64_t)8 * Bc + parts - 1) / parts) + 2048u + 255u) & ~(uint64_t)255u;
        if (cap > all) cap = all;
        o.pm.cur_base[l] = cur;
        o.pm.home_cap[l] = (uint32_t)cap;
        o.pm.home_base[l] = (uint32_t)home;
        home += cap * parts;
        cur += parts;
        mp = parts > mp ? parts : mp;
    }
    o.pm.max_parts = mp;
    o.pm.cursors = 3u * cur;               // cursors, then the tickets of the split levels, one set per table (workgroup 0 of the fill clears all)
    const uint64_t ovf = (uint64_t)plan.levels * 8u * Bc;
    o.pm.ovf_cap = (uint32_t)ovf;
    if (home >= (1ull << 30) || ovf >= (1ull << 32) || mp > kMaxPartsPerLevel) o.ok = false;      // (fill: byte offsets of an entry in 32 bits)
    o.home_entries = (size_t)home;
    // scratch slots of the levels whose partitions are split over several work items (64 KB each): the larger of the two tables' needs
    uint64_t split1 = 0, split2 = 0;
    for (uint32_t l = 0; l < plan.levels; ++l) {
        const uint32_t pairs = (plan.parts[l] + 1u) / 2u;
        uint32_t g1 = (uint32_t)(((uint64_t)8 * Bc / pairs + 65535u) / 65536u);
        g1 = g1 < 1u ? 1u : (g1 > 64u ? 64u : g1);
        if (g1 > 1u) split1 += (uint64_t)pairs * g1;
        if (plan.groups[l] > 1u) split2 += (uint64_t)plan.parts[l] * plan.groups[l];
    }
    o.slot_count = (size_t)(split1 + split2);                  // (both tables' items run in one launch)
    const size_t r = 255;
    o.bytes = kBinHeaderBytes + (((size_t)cur * 12 + r) & ~r) + ((home * 4 + r) & ~r) * 2 + ((home * 2 + r) & ~r) + (((size_t)ovf * 4 + r) & ~r) * 3 +
              o.slot_count * (size_t)kPairP * 16u;
    return o;
}

static uint32_t pm_tile_samples() {
    static const uint32_t ts = [] {
        const char* e = getenv("N2M_PM_TS");
        const uint32_t v = e ? (uint32_t)atoi(e) : 512u;
#ifdef N2M_PM_TS_EXTRA                 // (lab builds, tools/build_variant.py: one more tile size, e.g. 640 = ten waves)
        if (v == (uint32_t)N2M_PM_TS_EXTRA) return v;
#endif
        return (v == 256u || v == 512u || v == 1024u) ? v : 512u;
    }();
    return ts;
}
static bool pm_enabled() {
    static const bool on = getenv("N2M_BIN_PM") == nullptr || atoi(getenv("N2M_BIN_PM")) != 0;
    return on;
}

// An event the NEXT table backward of this thread records between its fill and its accumulate (n2m_grid_backward_mid_event): the step
// executor's go-ahead for the side stream.  The next batch's ray generation and march then start beside the accumulate instead of beside
// the optimizer update and are mostly over when the next step's lookup starts (the lookup ran 10 us longer with the marcher beside it).
static thread_local hipEvent_t g_mid_event = nullptr;
extern "C" int n2m_grid_backward_mid_event(void* event) {
    g_mid_event = (hipEvent_t)event;
    return 0;
}

// The order in which the table backward of this thread visits its samples (sticky until cleared with NULL): perm[i] = index of the i-th
// sample to visit.  Results do not depend on it beyond fp32 rounding of the run merge (the sums are fixed point); with the LIVE-FIRST
// order of n2m_sample_order_live_first the samples without a gradient -- the tails of the rays behind the early stop, about half of a
// trained batch -- fill whole waves, which take the fill's TV-only path.
// How many levels (from the coarsest) merge same-cell runs of consecutive samples in the NEXT table backwards of this thread (sticky; 0 = the
// default, kPairMergeLevels / N2M_BIN_MERGE_LEVELS).  Stage 1 sets 16: consecutive covered pixels of a frame share cells up to resolution ~800,
// where marched samples (the default's audience) stop sharing them at level 9.  Same sums either way (fixed point); only the fp32 rounding of a
// merged run differs.
static thread_local uint32_t g_merge_levels = 0u;
extern "C" int n2m_grid_backward_merge_levels(uint32_t levels) {
    N2M_REQUIRE(levels <= kMaxLevels, N2M_EINVAL, "n2m_grid_backward_merge_levels: %u > %u", levels, kMaxLevels);
    g_merge_levels = levels;
    return 0;
}

// The forward lookup's corner records for the NEXT table backward of this thread (sticky until cleared with NULL): [L, B, 4] fp32 written by
// n2m_grid_encode_forward_packed_tv over the SAME B samples and the SAME table state.  Consumed by the partition-major fill when it computes
// the TV term itself (tv_embeddings given), one pass (B <= 2^20), samples in input order, no folded copies, one point list; ignored otherwise.
static thread_local const float* g_tv_corners = nullptr;
extern "C" int n2m_grid_backward_tv_corners(const float* corners) {
    N2M_REQUIRE(((uintptr_t)corners & 15u) == 0, N2M_EINVAL, "n2m_grid_backward_tv_corners: records must be 16-byte aligned");
    g_tv_corners = corners;
    return 0;
}

static unsigned int g_fill_dbg_host = 0u;          // host mirror of g_fill_timing_on (n2m_debug_fill_times): a non-zero word selects the EX kernels
static thread_local const uint32_t* g_sample_order = nullptr;
extern "C" int n2m_grid_backward_sample_order(const uint32_t* perm) {
    g_sample_order = perm;
    return 0;
}

template <uint32_t TS>
int launch_pm_fill(dim3 grid, size_t lds, hipStream_t s, bool fold_on, int tvmode, const float* g1, const _Float16* g2, const float* x, TvParams tv,
                   const float* tvt, uint32_t Bc, uint32_t B, const BinPlan& plan, const PmPlan& pm, const LevelTable& lv, uint32_t gridtype, bool align,
                   uint32_t interp, uint32_t* level_max, uint32_t* cursors, uint32_t* ovf_cursor, uint16_t* log_rel, uint32_t* log_v1, uint32_t* log_v2,
                   uint32_t* ovf_key, uint32_t* ovf_v1, uint32_t* ovf_v2, float* found_inf, float in_scale, float in_offset, float* clear1,
                   _Float16* clear2, uint32_t cm1, uint32_t cm2, uint32_t merge_levels, uint32_t groups_x, uint32_t slot_begin,
                   unsigned long long* lm_ready, unsigned long long lm_token, uint32_t in_level_stride, const FoldArgs& fo, const uint32_t* perm) {
    static bool attr_set = false;
    if (!attr_set) {
        const int cap = 160 * 1024 - 1024;
        (void)hipFuncSetAttribute((const void*)pm_fill_pair_kernel<0, false, TS>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute((const void*)pm_fill_pair_kernel<1, false, TS>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute((const void*)pm_fill_pair_kernel<2, false, TS>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute((const void*)pm_fill_pair_kernel<0, true, TS>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        (void)hipFuncSetAttribute((const void*)pm_fill_pair_kernel<1, true, TS>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
        attr_set = true;
    }
#define N2M_PM_ARGS g1, g2, x, tv, tvt, Bc, B, plan, pm, lv, gridtype, align, interp, level_max, cursors, ovf_cursor, log_rel, log_v1, log_v2, ovf_key, ovf_v1, \
                    ovf_v2, found_inf, in_scale, in_offset, clear1, clear2, cm1, cm2, merge_levels, groups_x, slot_begin, lm_ready, lm_token, in_level_stride, fo, perm
    const bool extras = perm != nullptr || g_fill_dbg_host != 0u;
    if (extras && !fold_on) {
        static bool attr_ex = false;
        if (!attr_ex) {
            const int cap = 160 * 1024 - 1024;
            (void)hipFuncSetAttribute((const void*)pm_fill_pair_kernel<0, false, TS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            (void)hipFuncSetAttribute((const void*)pm_fill_pair_kernel<1, false, TS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            (void)hipFuncSetAttribute((const void*)pm_fill_pair_kernel<2, false, TS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            attr_ex = true;
        }
        if (tvmode == 1) N2M_LAUNCH((pm_fill_pair_kernel<1, false, TS, true>), grid, TS, lds, s, N2M_PM_ARGS);
        else if (tvmode == 2) N2M_LAUNCH((pm_fill_pair_kernel<2, false, TS, true>), grid, TS, lds, s, N2M_PM_ARGS);
        else N2M_LAUNCH((pm_fill_pair_kernel<0, false, TS, true>), grid, TS, lds, s, N2M_PM_ARGS);
    } else if (fold_on) {
        if (tvmode == 1) N2M_LAUNCH((pm_fill_pair_kernel<1, true, TS>), grid, TS, lds, s, N2M_PM_ARGS);
        else N2M_LAUNCH((pm_fill_pair_kernel<0, true, TS>), grid, TS, lds, s, N2M_PM_ARGS);
    } else if (tvmode == 1) N2M_LAUNCH((pm_fill_pair_kernel<1, false, TS>), grid, TS, lds, s, N2M_PM_ARGS);
    else if (tvmode == 2) N2M_LAUNCH((pm_fill_pair_kernel<2, false, TS>), grid, TS, lds, s, N2M_PM_ARGS);
    else N2M_LAUNCH((pm_fill_pair_kernel<0, false, TS>), grid, TS, lds, s, N2M_PM_ARGS);
#undef N2M_PM_ARGS
    N2M_CHECK_LAUNCH();
    return 0;
}

// returns -1 when the partition-major path does not cover the call (the caller then runs the tile-major path)
int launch_binned_pair_pm(const float* grad1, const _Float16* grad2, const float* inputs, TvParams tv, float* table1, _Float16* table2, uint32_t B,
                          uint32_t max_level, const int32_t* host_offsets, const LevelTable& lv, uint32_t gridtype, bool align, uint32_t interp,
                          void* workspace, size_t workspace_bytes, hipStream_t s, const char* fn, float* found_inf, float in_scale,
                          float in_offset, bool overwrite, uint32_t L, int half, const float* tv_terms, const FoldArgs* fold, uint32_t in_level_stride) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)pm_accumulate_kernel<float, 1, kPairP, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kPairP * 16));
        (void)hipFuncSetAttribute((const void*)pm_accumulate_kernel<_Float16, 2, kPairP, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kPairP * 16));
        (void)hipFuncSetAttribute((const void*)pm_accumulate_both_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kPairP * 16));
        attr_set = true;
    }
    const uint32_t TS = pm_tile_samples();
    N2M_REQUIRE(in_level_stride == 0 || B <= kBinChunk, N2M_EINVAL, "%s: per-level point lists need one pass (B <= %u)", fn, kBinChunk);
    for (uint32_t b0 = 0; b0 < B; b0 += kBinChunk) {
        const uint32_t Bc = B - b0 < kBinChunk ? B - b0 : kBinChunk;
        BinLayout lay = make_bin_plan(Bc, 2, max_level, host_offsets, false, kPairP, 2);
        N2M_REQUIRE(lay.ok, N2M_EUNSUPPORTED, "%s: table layout not supported by the binned path", fn);
        lay.plan.tiles = (Bc + TS - 1) / TS;
        const PmLayout pl = make_pm_plan(Bc, lay.plan);
        if (!pl.ok) return -1;
        {   // A call over a few small dense levels only (the SDF recipe's early schedule: 4-5 levels, 18-45 partition pairs, 1.5 M stacked
            // samples whose entries pile onto a few thousand rows) stays on the tile-major path: measured 127 + 196 us here against 99 + 69 us
            // there (tile groups split such a level 64 ways by construction; slices of a region cannot).
            uint32_t pairs_all = 0;
            for (uint32_t l = 0; l < max_level; ++l) pairs_all += (lay.plan.parts[l] + 1u) / 2u;
            if (pairs_all < 64u) return -1;
        }
        N2M_REQUIRE(workspace_bytes >= pl.bytes, N2M_EINVAL, "%s: workspace too small (%zu < %zu bytes)", fn, workspace_bytes, pl.bytes);
        const size_t r = 255;
        char* w = (char*)workspace;
        uint32_t* level_max = (uint32_t*)w;
        unsigned long long* lm_ready = (unsigned long long*)(w + 256);
        uint32_t* ovf_cursor = (uint32_t*)(w + 264);
        uint32_t* cursors = (uint32_t*)(w + kBinHeaderBytes);
        uint32_t* tickets = cursors + pl.pm.cursors / 3u;
        uint32_t* tickets2 = tickets + pl.pm.cursors / 3u;
        w += kBinHeaderBytes + (((size_t)pl.pm.cursors * 4 + r) & ~r);
        uint32_t* log_v1 = (uint32_t*)w; w += (pl.home_entries * 4 + r) & ~r;
        uint32_t* log_v2 = (uint32_t*)w; w += (pl.home_entries * 4 + r) & ~r;
        uint16_t* log_rel = (uint16_t*)w; w += (pl.home_entries * 2 + r) & ~r;
        uint32_t* ovf_key = (uint32_t*)w; w += ((size_t)pl.pm.ovf_cap * 4 + r) & ~r;
        uint32_t* ovf_v1 = (uint32_t*)w; w += ((size_t)pl.pm.ovf_cap * 4 + r) & ~r;
        uint32_t* ovf_v2 = (uint32_t*)w; w += ((size_t)pl.pm.ovf_cap * 4 + r) & ~r;
        unsigned long long* slots = (unsigned long long*)w;
        static std::atomic<unsigned long long> launch_counter{1};
        const unsigned long long lm_token = (0x6e326d50ull << 32) | (launch_counter.fetch_add(1) & 0xFFFFFFFFull);
        const bool ow = overwrite && b0 == 0;
        const bool both = grad1 != nullptr, has2 = grad2 != nullptr;
        const float* g1 = both ? grad1 + (size_t)b0 : nullptr;
        if (!both) { log_v1 = nullptr; ovf_v1 = nullptr; }
        const _Float16* g2 = has2 ? grad2 + (size_t)b0 * 2 : nullptr;
        if (!has2) { log_v2 = nullptr; ovf_v2 = nullptr; }
        const float* x = inputs + (size_t)b0 * 3;
        auto in_half = [&](uint32_t l) { return half == 0 || (half == 1 ? l >= 8u : l < 8u); };
        BinPlan plan1 = lay.plan, plan2 = lay.plan;
        uint32_t items1 = 0, items2 = 0, slots1 = 0, slots2 = 0;
        const uint32_t cm1 = 0, cm2 = 0;                      // (no rows to clear: split levels end in stores too)
        PmSplit sp1{}, sp2{};
        for (uint32_t l = 0; l < max_level; ++l) {
            const uint32_t pairs = (plan1.parts[l] + 1u) / 2u;
            const uint64_t per_item = (uint64_t)8 * Bc / pairs;
            uint32_t g = (uint32_t)((per_item + 65535u) / 65536u);
            g = g < 1u ? 1u : (g > 64u ? 64u : g);
            uint32_t g2n = lay.plan.groups[l];                      // (make_bin_plan clamped it to ITS tile count)
            plan1.groups[l] = g;
            plan2.groups[l] = g2n;
            plan1.item_prefix[l] = items1;
            plan2.item_prefix[l] = items2;
            sp1.slot0[l] = slots1; sp2.slot0[l] = slots2;
            sp1.tick0[l] = sp2.tick0[l] = pl.pm.cur_base[l];            // (a ticket per partition group: never more than partitions)
            if (!in_half(l)) continue;
            items1 += pairs * g;
            items2 += lay.plan.parts[l] * g2n;
            if (g > 1u) slots1 += pairs * g;
            if (g2n > 1u) slots2 += lay.plan.parts[l] * g2n;
        }
        N2M_REQUIRE((size_t)slots1 + slots2 <= pl.slot_count, N2M_EINVAL, "%s: scratch slots of the split levels exceed the layout", fn);
        {   // partition groups of the launched levels: few of them = short slices (see pm_accumulate_items)
            uint32_t groups1 = 0, groups2 = 0;
            for (uint32_t l = 0; l < max_level; ++l)
                if (in_half(l)) { groups1 += (plan1.parts[l] + 1u) / 2u; groups2 += plan2.parts[l]; }
            sp1.slice_log2 = groups1 < 256u ? 13u : 16u;
            sp2.slice_log2 = groups2 < 512u ? 13u : 15u;
        }
        unsigned long long* slots_t2 = slots + (size_t)slots1 * (kPairP * 2u);
        plan1.item_prefix[max_level] = items1;
        plan2.item_prefix[max_level] = items2;
        if (ow && max_level < L) {
            const size_t t0 = (size_t)host_offsets[max_level], t1 = (size_t)host_offsets[L];
            if (both) N2M_HIP(hipMemsetAsync(table1 + t0, 0, (t1 - t0) * sizeof(float), s));
            if (has2) N2M_HIP(hipMemsetAsync(table2 + t0 * 2u, 0, (t1 - t0) * 2u * sizeof(_Float16), s));
        }
        static const uint32_t merge_env = getenv("N2M_BIN_MERGE_LEVELS") ? (uint32_t)atoi(getenv("N2M_BIN_MERGE_LEVELS")) : kPairMergeLevels;
        const uint32_t merge_levels = has2 ? (g_merge_levels ? g_merge_levels : merge_env) : kMaxLevels;
        static const bool xcd_map = getenv("N2M_FILL_NO_XCD") == nullptr;
        static const uint32_t tiles_per_wg = getenv("N2M_PM_TILES") ? (uint32_t)atoi(getenv("N2M_PM_TILES")) : kPairTilesPerWg;
        dim3 grid((lay.plan.tiles + tiles_per_wg - 1) / tiles_per_wg, max_level);
        uint32_t groups_x = 0, slot_begin = 0;
        N2M_REQUIRE(half == 0 || (xcd_map && max_level == 16u && (half == 1 || half == 2)), N2M_EUNSUPPORTED,
                    "%s: level halves need max_level == 16 and the XCD-aware fill", fn);
        if (xcd_map && (max_level == 16u || half != 0)) {
            groups_x = grid.x;
            uint32_t slots = 2u * ((max_level + 15u) / 16u);
            if (half != 0) { slot_begin = (uint32_t)half - 1u; slots = 1u; }
            grid = dim3(8u * slots * groups_x, 1);
        }
        const float* tvt = tv_terms ? tv_terms + (size_t)b0 : nullptr;
        FoldArgs fo{};
        if (fold) {
            N2M_REQUIRE(both && !tvt && B <= kBinChunk && half == 0 && in_level_stride == 0, N2M_EINVAL,
                        "%s: folded copies need the density gradient, one pass (B <= %u), all levels and one point list", fn, kBinChunk);
            fo = *fold;
            fo.stride6 = 6u * B;
        }
        const uint32_t MP = (pl.pm.max_parts + 127u) & ~127u;
        const size_t lds = (size_t)TS * 8u * 12u + (size_t)MP * 5u * 4u;
        const int tvmode = tv.table ? 1 : (tvt ? 2 : 0);
        const uint32_t* perm = g_sample_order;
        tv.corners = (g_tv_corners && tv.table && B <= kBinChunk && b0 == 0 && !fold && in_level_stride == 0 && perm == nullptr && g_fill_dbg_host == 0u)
                         ? g_tv_corners : nullptr;
        N2M_REQUIRE(perm == nullptr || (B <= kBinChunk && !fold && in_level_stride == 0), N2M_EUNSUPPORTED,
                    "%s: a sample order (n2m_grid_backward_sample_order) needs one pass (B <= %u), one point list and no folded copies", fn, kBinChunk);
        int rc;
        // [round 6, MEASURED AND REJECTED -- off by default, N2M_PM_SPLIT=1 turns it on] Fine levels (8..15) through pm_fill_fine_kernel at three
        // workgroups per CU, the coarse levels through the general kernel behind it (same stream; it continues the fill: no clear, same
        // token).  Covers the standard table (16 levels, the upper eight hashed with 2..128 partitions each), 512-sample tiles, no folded
        // copies, no sample order, no measurement switches; level 8 then goes unmerged.  Measured (ABBA on one box, profiles/r06_fill_split.txt):
        // backward 207.6 -> 224.8 us.  By kernel: fine levels alone 87.8 us at 96 workgroups per XCD (3 per CU), 92.3 at 64 (2 per CU), 102.6
        // at 48, 98.9 at 128 -- a third resident workgroup buys 5 %, i.e. the fine levels are NOT bound by their waits but by what eight
        // levels' worth of scattered requests cost their XCD's L2 (six TV stencil gathers per sample and level + the log's run stores); the
        // coarse levels alone take 86-88 us at two workgroups per CU, and the one-kernel fill (158.5 us in the same trace) wins because a
        // fine and a coarse workgroup SHARE each CU: different bottlenecks side by side.
        static const bool split_env = getenv("N2M_PM_SPLIT") != nullptr && atoi(getenv("N2M_PM_SPLIT")) != 0;
        bool fine = split_env && TS == 512u && xcd_map && max_level == 16u && gridtype == 0u && !fold && perm == nullptr && g_fill_dbg_host == 0u &&
                    merge_levels <= 9u && lay.plan.tiles >= 8u;
        for (uint32_t l = 8; fine && l < 16u; ++l) {
            const uint64_t R = align ? lv.resolution[l] : lv.resolution[l] + 1u;
            const uint32_t sz = lay.plan.size[l];
            fine = R * R * R > sz && (sz & (sz - 1u)) == 0u && sz <= (1u << 19) && lay.plan.parts[l] >= 2u && lay.plan.parts[l] <= 128u &&
                   lay.plan.parts[l] * kPairP == sz;
        }
        if (fine) {
            static bool attr_fine = false;
            if (!attr_fine) {
                const int capb = 160 * 1024 - 1024;
                (void)hipFuncSetAttribute((const void*)pm_fill_fine_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, capb);
                (void)hipFuncSetAttribute((const void*)pm_fill_fine_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, capb);
                (void)hipFuncSetAttribute((const void*)pm_fill_fine_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, capb);
                attr_fine = true;
            }
            static const uint32_t fine_groups = getenv("N2M_PM_FINE_GROUPS") ? (uint32_t)atoi(getenv("N2M_PM_FINE_GROUPS")) : 96u;      // 3 workgroups x 32 CUs of an XCD
            if (half != 2) {
                const uint32_t gx = lay.plan.tiles < fine_groups ? lay.plan.tiles : fine_groups;
                const size_t lds_f = (size_t)512u * 8u * 12u + (size_t)128u * 5u * 4u;
#define N2M_PM_FINE_ARGS g1, g2, x, tv, tvt, Bc, B, lay.plan, pl.pm, lv, align, interp, level_max, cursors, ovf_cursor, log_rel, log_v1, log_v2, ovf_key, ovf_v1, \
                         ovf_v2, found_inf, in_scale, in_offset, gx, 8u, lm_ready, lm_token, in_level_stride
                if (tvmode == 1) N2M_LAUNCH((pm_fill_fine_kernel<1>), dim3(8u * gx), 512, lds_f, s, N2M_PM_FINE_ARGS);
                else if (tvmode == 2) N2M_LAUNCH((pm_fill_fine_kernel<2>), dim3(8u * gx), 512, lds_f, s, N2M_PM_FINE_ARGS);
                else N2M_LAUNCH((pm_fill_fine_kernel<0>), dim3(8u * gx), 512, lds_f, s, N2M_PM_FINE_ARGS);
#undef N2M_PM_FINE_ARGS
                N2M_CHECK_LAUNCH();
            }
            if (half == 0) {                 // the coarse half through the general kernel: slot 1 of the paired mapping = levels 0..7, continuing
                slot_begin = 1u | 0x80000000u;
                grid = dim3(8u * groups_x, 1);
            }
        }
        if (fine && half == 1) rc = 0;
        else
#define N2M_PM_CALL(TSV) launch_pm_fill<TSV>(grid, lds, s, fold != nullptr, tvmode, g1, g2, x, tv, tvt, Bc, B, lay.plan, pl.pm, lv, gridtype, align, interp, level_max, \
                                             cursors, ovf_cursor, log_rel, log_v1, log_v2, ovf_key, ovf_v1, ovf_v2, found_inf, in_scale, in_offset,                   \
                                             ow ? table1 : nullptr, ow ? table2 : nullptr, cm1, cm2, merge_levels, groups_x, slot_begin, lm_ready, lm_token,      \
                                             in_level_stride, fo, perm)
        if (TS == 256u) rc = N2M_PM_CALL(256);
        else if (TS == 1024u) rc = N2M_PM_CALL(1024);
#ifdef N2M_PM_TS_EXTRA
        else if (TS == (uint32_t)N2M_PM_TS_EXTRA) rc = N2M_PM_CALL(N2M_PM_TS_EXTRA);
#endif
        else rc = N2M_PM_CALL(512);
#undef N2M_PM_CALL
        if (rc) return rc;
        if (g_mid_event) {                                  // (one-shot: the caller arms it per call)
            (void)hipEventRecord(g_mid_event, s);
            g_mid_event = nullptr;
        }
        static const uint32_t acc_cap = getenv("N2M_ACC_GRID") ? (uint32_t)atoi(getenv("N2M_ACC_GRID")) : 4096u;
        static const uint32_t acc_dbg = getenv("N2M_ACC_DEBUG") ? (uint32_t)atoi(getenv("N2M_ACC_DEBUG")) : 0u;
        const float odiv = g_cfg_overflow_div.load();
        PeerRouteK prk{};
        PeerRouteT prt1{}, prt2{};
        if (g_peer_route.world) {
            N2M_REQUIRE(ow && B <= kBinChunk && max_level == L && !fold, N2M_EUNSUPPORTED,
                        "%s: a routed flush (n2m_grid_backward_peer_route) needs overwrite mode, one pass, all levels and no folded copies", fn);
            prk = PeerRouteK{g_peer_route.world, g_peer_route.split_row, g_peer_route.rows_c, g_peer_route.rows_f};
            for (int h = 0; h < 2; ++h)
                for (uint32_t r = 0; r < g_peer_route.world; ++r) { prt1.base[h][r] = g_peer_route.g1[h][r]; prt2.base[h][r] = g_peer_route.g2[h][r]; }
        }
        const uint32_t nb1 = items1 < acc_cap ? items1 : acc_cap, nb2 = items2 < acc_cap ? items2 : acc_cap;
        static const bool one_launch = getenv("N2M_PM_ACC_SPLIT") == nullptr;
        if (both && has2 && one_launch) {
            PmBoth pb{plan1, plan2, sp1, sp2};
            N2M_LAUNCH(pm_accumulate_both_kernel, nb1 + nb2, 1024, kPairP * 16, s, table1, table2, pb, pl.pm, lv, gridtype, align, level_max, cursors, ovf_cursor, log_rel, log_v1,
                                                                           log_v2, ovf_key, ovf_v1, ovf_v2, slots, slots_t2, tickets, tickets2, found_inf, ow, acc_dbg,
                                                                           3.0e38f / odiv, 65504.0f / odiv, nb1, prk, prt1, prt2);
            N2M_CHECK_LAUNCH();
            continue;
        }
        if (both) {
            N2M_LAUNCH((pm_accumulate_kernel<float, 1, kPairP, 2>), nb1, 1024, kPairP * 16, s, 
                table1, plan1, pl.pm, sp1, lv, gridtype, align, level_max, cursors, ovf_cursor, log_rel, log_v1, ovf_key, ovf_v1, slots, tickets, found_inf, ow, acc_dbg,
                3.0e38f / odiv, prk, prt1);
            N2M_CHECK_LAUNCH();
        }
        if (has2) {
            N2M_LAUNCH((pm_accumulate_kernel<_Float16, 2, kPairP, 1>), nb2, 1024, kPairP * 16, s, 
                table2, plan2, pl.pm, sp2, lv, gridtype, align, level_max + kMaxLevels, cursors, ovf_cursor, log_rel, log_v2, ovf_key, ovf_v2, slots_t2, tickets2, found_inf, ow,
                acc_dbg, 65504.0f / odiv, prk, prt2);
            N2M_CHECK_LAUNCH();
        }
    }
    return 0;
}

int launch_binned_pair(const float* grad1, const _Float16* grad2, const float* inputs, TvParams tv, float* table1, _Float16* table2, uint32_t B,
                       uint32_t max_level, const int32_t* host_offsets, const LevelTable& lv, uint32_t gridtype, bool align, uint32_t interp,
                       void* workspace, size_t workspace_bytes, hipStream_t s, const char* fn, float* found_inf, float in_scale,
                       float in_offset, bool overwrite, uint32_t L, int half = 0, const float* tv_terms = nullptr,
                       const AdamFuse* fuse1 = nullptr, const AdamFuse* fuse2 = nullptr, const FoldArgs* fold = nullptr,
                       uint32_t in_level_stride = 0) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)bin_fill_pair_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kTileEntries * 10));
        (void)hipFuncSetAttribute((const void*)bin_fill_pair_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kTileEntries * 10));
        (void)hipFuncSetAttribute((const void*)bin_fill_pair_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kTileEntries * 10));
        (void)hipFuncSetAttribute((const void*)bin_fill_pair_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kTileEntries * 10));
        (void)hipFuncSetAttribute((const void*)bin_fill_pair_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kTileEntries * 10));
        (void)hipFuncSetAttribute((const void*)bin_accumulate_kernel<float, 1, kPairP, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kPairP * 16));
        (void)hipFuncSetAttribute((const void*)bin_accumulate_kernel<_Float16, 2, kPairP, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kPairP * 16));
        (void)hipFuncSetAttribute((const void*)bin_accumulate_kernel<float, 1, kPairP, 2, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kPairP * 16));
        (void)hipFuncSetAttribute((const void*)bin_accumulate_kernel<_Float16, 2, kPairP, 1, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kPairP * 16));
        attr_set = true;
    }
    if (pm_enabled() && !fuse1 && !fuse2) {                     // partition-major log (round 4); -1: layout not covered, tile-major path below
        const int rc = launch_binned_pair_pm(grad1, grad2, inputs, tv, table1, table2, B, max_level, host_offsets, lv, gridtype, align, interp, workspace,
                                             workspace_bytes, s, fn, found_inf, in_scale, in_offset, overwrite, L, half, tv_terms, fold, in_level_stride);
        if (rc != -1) return rc;
        n2m_prof_fall_back_to_markers(s);                      // the tile-major launches below are plain ones
    }
    N2M_REQUIRE(g_sample_order == nullptr, N2M_EUNSUPPORTED, "%s: a sample order (n2m_grid_backward_sample_order) needs the partition-major path", fn);
    N2M_REQUIRE(g_peer_route.world == 0, N2M_EUNSUPPORTED, "%s: a routed flush (n2m_grid_backward_peer_route) needs the partition-major path", fn);
    N2M_REQUIRE(in_level_stride == 0 || B <= kBinChunk, N2M_EINVAL, "%s: per-level point lists need one pass (B <= %u)", fn, kBinChunk);
    for (uint32_t b0 = 0; b0 < B; b0 += kBinChunk) {
        const uint32_t Bc = B - b0 < kBinChunk ? B - b0 : kBinChunk;
        const BinLayout lay = make_bin_plan(Bc, 2, max_level, host_offsets, false, kPairP, 2);     // 2 x 8 B per entry >= the 10 B used
        N2M_REQUIRE(lay.ok, N2M_EUNSUPPORTED, "%s: table layout not supported by the binned path", fn);
        N2M_REQUIRE(workspace_bytes >= lay.bytes, N2M_EINVAL, "%s: workspace too small (%zu < %zu bytes)", fn, workspace_bytes, lay.bytes);
        uint32_t* level_max = (uint32_t*)workspace;                                  // [2][32]
        unsigned long long* lm_ready = (unsigned long long*)((char*)workspace + 256);  // token of the launch that last cleared level_max
        uint32_t* directory = (uint32_t*)((char*)workspace + kBinHeaderBytes);
        uint32_t* log_v1 = (uint32_t*)((char*)workspace + kBinHeaderBytes + ((lay.dir_words * 4 + 255) & ~(size_t)255));
        uint32_t* log_v2 = log_v1 + lay.log_entries;
        uint16_t* log_rel = (uint16_t*)(log_v2 + lay.log_entries);
        static std::atomic<unsigned long long> launch_counter{1};
        const unsigned long long lm_token = (0x6e326d4cull << 32) | (launch_counter.fetch_add(1) & 0xFFFFFFFFull);   // never what stale memory holds
        const bool ow = overwrite && b0 == 0;                 // later passes add onto the first one's sums
        // grad1 == NULL: the colour table alone (stage 1 shades with the colour networks only, nerf/renderer.py:875-881): the fill neither
        // reads a density gradient nor writes its log, the fp32 accumulate is not launched, table1 is not touched
        // grad2 == NULL: the density table alone (the stacked finite-difference evaluations of the SDF head, nerf/network.py:143-154)
        const bool both = grad1 != nullptr, has2 = grad2 != nullptr;
        const float* g1 = both ? grad1 + (size_t)b0 : nullptr;
        if (!both) log_v1 = nullptr;
        const _Float16* g2 = has2 ? grad2 + (size_t)b0 * 2 : nullptr;
        if (!has2) log_v2 = nullptr;
        const float* x = inputs + (size_t)b0 * 3;
        // the fp32 table accumulates two partitions per work item: same item count and LDS bytes as on its own 8192-row structure
        // half != 0 (max_level == 16, XCD-aware fill): this call covers the levels of ONE fill slot only -- 1: levels 8..15 (slot 0),
        // 2: levels 0..7 (slot 1).  A caller that exchanges gradients between GPUs issues the fine half first and starts its collective
        // on those table rows while the coarse half is still being computed (nerf2mesh_amd/engine.py, multi-rank path).
        auto in_half = [&](uint32_t l) { return half == 0 || (half == 1 ? l >= 8u : l < 8u); };
        BinPlan plan1 = lay.plan, plan2 = lay.plan;
        uint32_t items1 = 0, items2 = 0, cm1 = 0, cm2 = 0;
        for (uint32_t l = 0; l < max_level; ++l) {
            const uint32_t pairs = (plan1.parts[l] + 1u) / 2u;
            const uint64_t per_item = (uint64_t)8 * Bc / pairs;
            uint32_t g = (uint32_t)((per_item + 65535u) / 65536u);
            g = g < 1u ? 1u : (g > 64u ? 64u : g);
            g = g > plan1.tiles ? plan1.tiles : g;
            plan1.groups[l] = g;
            plan1.item_prefix[l] = items1;
            plan2.item_prefix[l] = items2;
            if (!in_half(l)) continue;                              // no work items for the levels of the other half
            items1 += pairs * g;
            items2 += lay.plan.parts[l] * lay.plan.groups[l];
            if (ow && g > 1u && both) cm1 |= 1u << l;              // levels that add atomically: cleared by the fill kernel
            if (ow && lay.plan.groups[l] > 1u && has2) cm2 |= 1u << l;
        }
        plan1.item_prefix[max_level] = items1;
        plan2.item_prefix[max_level] = items2;
        if (ow && max_level < L) {                                  // levels the call does not touch
            const size_t t0 = (size_t)host_offsets[max_level], t1 = (size_t)host_offsets[L];
            if (both) N2M_HIP(hipMemsetAsync(table1 + t0, 0, (t1 - t0) * sizeof(float), s));
            if (has2) N2M_HIP(hipMemsetAsync(table2 + t0 * 2u, 0, (t1 - t0) * 2u * sizeof(_Float16), s));
        }
        static const uint32_t merge_env = getenv("N2M_BIN_MERGE_LEVELS") ? (uint32_t)atoi(getenv("N2M_BIN_MERGE_LEVELS")) : kPairMergeLevels;
        // the density table alone = the SDF head's finite-difference copies, six adjacent samples a few 1e-4 apart: same cell on ALL levels
        const uint32_t merge_levels = has2 ? merge_env : kMaxLevels;
        static const bool xcd_map = getenv("N2M_FILL_NO_XCD") == nullptr;       // A/B switch; measured 332 -> 311 us for fill + accumulates
        dim3 grid((lay.plan.tiles + kPairTilesPerWg - 1) / kPairTilesPerWg, max_level);           // each workgroup walks ~kPairTilesPerWg tiles
        uint32_t groups_x = 0, slot_begin = 0;
        N2M_REQUIRE(half == 0 || (xcd_map && max_level == 16u && (half == 1 || half == 2)), N2M_EUNSUPPORTED,
                    "%s: level halves need max_level == 16 and the XCD-aware fill", fn);
        // (the XCD-aware order deals level PAIRS (p, L-1-p) to the 8 XCDs: with fewer than 16 levels -- progressive training of the SDF recipe,
        // max_level 4..15 -- it would leave most XCDs without work, so those calls keep the plain (group, level) grid)
        if (xcd_map && (max_level == 16u || half != 0)) {           // 1-D grid: 8 XCDs x ceil(levels / 8) level slots (rounded to pairs) x groups
            groups_x = grid.x;
            uint32_t slots = 2u * ((max_level + 15u) / 16u);
            if (half != 0) { slot_begin = (uint32_t)half - 1u; slots = 1u; }
            grid = dim3(8u * slots * groups_x, 1);
        }
        const float* tvt = tv_terms ? tv_terms + (size_t)b0 : nullptr;          // [L, B]: this pass's column block
#define N2M_FILL_ARGS g1, g2, x, tv, tvt, Bc, B, lay.plan, lv, gridtype, align, interp, level_max, directory, log_rel, log_v1, log_v2, found_inf, in_scale, \
                      in_offset, ow ? table1 : nullptr, ow ? table2 : nullptr, cm1, cm2, merge_levels, groups_x, slot_begin, lm_ready, lm_token, in_level_stride
        if (fold) {
            N2M_REQUIRE(both && !tvt && B <= kBinChunk && half == 0 && in_level_stride == 0, N2M_EINVAL,
                        "%s: folded copies need the density gradient, one pass (B <= %u), all levels and one point list", fn, kBinChunk);
            FoldArgs fo = *fold;
            fo.stride6 = 6u * B;
            if (tv.table) bin_fill_pair_kernel<1, true><<<grid, 1024, kTileEntries * 10, s>>>(N2M_FILL_ARGS, fo);
            else bin_fill_pair_kernel<0, true><<<grid, 1024, kTileEntries * 10, s>>>(N2M_FILL_ARGS, fo);
        } else
        if (tv.table) bin_fill_pair_kernel<1><<<grid, 1024, kTileEntries * 10, s>>>(N2M_FILL_ARGS);
        else if (tvt) bin_fill_pair_kernel<2><<<grid, 1024, kTileEntries * 10, s>>>(N2M_FILL_ARGS);
        else bin_fill_pair_kernel<0><<<grid, 1024, kTileEntries * 10, s>>>(N2M_FILL_ARGS);
#undef N2M_FILL_ARGS
        N2M_CHECK_LAUNCH();
        const uint32_t items = items2;
        static const uint32_t acc_cap2 = getenv("N2M_ACC_GRID") ? (uint32_t)atoi(getenv("N2M_ACC_GRID")) : 4096u;
        const uint32_t nb = items < acc_cap2 ? items : acc_cap2;
        static const uint32_t acc_dbg = getenv("N2M_ACC_DEBUG") ? (uint32_t)atoi(getenv("N2M_ACC_DEBUG")) : 0u;   // measurement switches (wrong results)
        static const uint32_t acc_cap = getenv("N2M_ACC_GRID") ? (uint32_t)atoi(getenv("N2M_ACC_GRID")) : 4096u;    // A/B: persistent workgroups
        const float odiv = g_cfg_overflow_div.load();
        if (fuse1 || fuse2) {
            // Adam in the flush: every fused level must be owned by one work item per partition (its flush holds the FINAL sums), in one pass
            N2M_REQUIRE(both && has2 && fuse1 && fuse2 && ow && half == 0 && B <= kBinChunk, N2M_EINVAL,
                        "%s: the fused optimizer pass needs both tables, overwrite mode, all levels and one pass (B <= %u)", fn, kBinChunk);
            for (uint32_t l = fuse1->first_level; l < max_level; ++l)
                N2M_REQUIRE(plan1.groups[l] == 1u && lay.plan.groups[l] == 1u, N2M_EINVAL,
                            "%s: level %u is split over tile groups at B = %u: it cannot take the fused optimizer pass (n2m_grid_pair_fuse_plan)", fn, l, B);
            bin_accumulate_kernel<float, 1, kPairP, 2, true, true><<<items1 < acc_cap ? items1 : acc_cap, 1024, kPairP * 16, s>>>(
                table1, plan1, lv, gridtype, align, level_max, directory, nullptr, found_inf, log_rel, log_v1, ow, acc_dbg, 3.0e38f / odiv, *fuse1);
            N2M_CHECK_LAUNCH();
            bin_accumulate_kernel<_Float16, 2, kPairP, 1, true, true><<<nb, 1024, kPairP * 16, s>>>(
                table2, plan2, lv, gridtype, align, level_max + kMaxLevels, directory, nullptr, found_inf, log_rel, log_v2, ow, acc_dbg, 65504.0f / odiv, *fuse2);
            N2M_CHECK_LAUNCH();
            continue;
        }
        if (both) {
            bin_accumulate_kernel<float, 1, kPairP, 2, true><<<items1 < acc_cap ? items1 : acc_cap, 1024, kPairP * 16, s>>>(
                table1, plan1, lv, gridtype, align, level_max, directory, nullptr, found_inf, log_rel, log_v1, ow, acc_dbg, 3.0e38f / odiv);
            N2M_CHECK_LAUNCH();
        }
        if (has2) {
            bin_accumulate_kernel<_Float16, 2, kPairP, 1, true><<<nb, 1024, kPairP * 16, s>>>(table2, plan2, lv, gridtype, align, level_max + kMaxLevels,
                                                                                            directory, nullptr, found_inf, log_rel, log_v2, ow, acc_dbg, 65504.0f / odiv);
            N2M_CHECK_LAUNCH();
        }
    }
    return 0;
}

// D in {2,3,4,5} x C in {1,2,4,8}: pick the instantiation
#define N2M_DISPATCH_DC(D, C, FN, ...)                                             \
    switch ((D) * 16 + (C)) {                                                      \
        case 2 * 16 + 1: FN(2, 1, __VA_ARGS__); break;                             \
        case 2 * 16 + 2: FN(2, 2, __VA_ARGS__); break;                             \
        case 2 * 16 + 4: FN(2, 4, __VA_ARGS__); break;                             \
        case 2 * 16 + 8: FN(2, 8, __VA_ARGS__); break;                             \
        case 3 * 16 + 1: FN(3, 1, __VA_ARGS__); break;                             \
        case 3 * 16 + 2: FN(3, 2, __VA_ARGS__); break;                             \
        case 3 * 16 + 4: FN(3, 4, __VA_ARGS__); break;                             \
        case 3 * 16 + 8: FN(3, 8, __VA_ARGS__); break;                             \
        case 4 * 16 + 1: FN(4, 1, __VA_ARGS__); break;                             \
        case 4 * 16 + 2: FN(4, 2, __VA_ARGS__); break;                             \
        case 4 * 16 + 4: FN(4, 4, __VA_ARGS__); break;                             \
        case 4 * 16 + 8: FN(4, 8, __VA_ARGS__); break;                             \
        case 5 * 16 + 1: FN(5, 1, __VA_ARGS__); break;                             \
        case 5 * 16 + 2: FN(5, 2, __VA_ARGS__); break;                             \
        case 5 * 16 + 4: FN(5, 4, __VA_ARGS__); break;                             \
        case 5 * 16 + 8: FN(5, 8, __VA_ARGS__); break;                             \
        default: break;                                                            \
    }

int check_dims(const char* fn, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level, int dtype) {
    N2M_REQUIRE(D >= 2 && D <= 5, N2M_EINVAL, "%s: GridEncoding: D must be 2, 3, 4 or 5 (got %u)", fn, D);
    N2M_REQUIRE(C == 1 || C == 2 || C == 4 || C == 8, N2M_EINVAL, "%s: GridEncoding: C must be 1, 2, 4, or 8 (got %u)", fn, C);
    N2M_REQUIRE(L >= 1 && L <= kMaxLevels, N2M_EINVAL, "%s: L must be in 1..%u (got %u)", fn, kMaxLevels, L);
    N2M_REQUIRE(max_level <= L, N2M_EINVAL, "%s: max_level %u > L %u", fn, max_level, L);
    N2M_REQUIRE(dtype == N2M_F32 || dtype == N2M_F16, N2M_EINVAL, "%s: dtype must be N2M_F32 or N2M_F16 (got %d)", fn, dtype);
    return 0;
}

#define FWD_CASE(D, C, T, BM, a) launch_forward<T, D, C, BM>(a)
#define BWD_CASE(D, C, T, BM, a) launch_backward<T, D, C, BM>(a)
#define TV_CASE(D, C, a) launch_tv<D, C>(a)

template <bool BM>
int forward_any(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs, uint32_t B, uint32_t D,
                uint32_t C, uint32_t L, uint32_t max_level, float S, uint32_t H, void* dy_dx, uint32_t gridtype,
                int align_corners, uint32_t interp, int dtype, void* stream, const char* fn) {
    if (int rc = check_dims(fn, D, C, L, max_level, dtype)) return rc;
    N2M_REQUIRE(inputs && embeddings && offsets && outputs, N2M_ENULL, "%s: NULL tensor", fn);
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) return 0;
    const size_t esz = dtype == N2M_F16 ? 2 : 4;
    if (BM && max_level < L) {
        const uint32_t first = max_level * C, count = (L - max_level) * C;
        const uint32_t nb = n2m_ceil_div((uint64_t)B * count, 256);
        if (dtype == N2M_F16) zero_levels_kernel<_Float16><<<nb, 256, 0, s>>>((_Float16*)outputs, B, L * C, first, count);
        else zero_levels_kernel<float><<<nb, 256, 0, s>>>((float*)outputs, B, L * C, first, count);
        N2M_CHECK_LAUNCH();
    }
    if (max_level == 0) return 0;
    FwdArgs a{inputs, embeddings, offsets, outputs, B, L, max_level, make_levels(L, S, H), dy_dx, gridtype, align_corners != 0, interp, s};
    const double bytes = (double)B * (4.0 * D + (double)max_level * (1u << D) * C * esz + (double)max_level * C * esz +
                                      (dy_dx ? (double)max_level * D * C * esz : 0.0));
    N2M_PROF(N2M_K_GRID_FWD, s, bytes);
    if (dtype == N2M_F16) { N2M_DISPATCH_DC(D, C, FWD_CASE, _Float16, BM, a); }
    else { N2M_DISPATCH_DC(D, C, FWD_CASE, float, BM, a); }
    N2M_CHECK_LAUNCH();
    return 0;
}

template <bool BM>
int backward_any(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets, void* grad_embeddings,
                 uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level, float S, uint32_t H, const void* dy_dx,
                 void* grad_inputs, uint32_t gridtype, int align_corners, uint32_t interp, int dtype, void* stream,
                 const char* fn) {
    (void)embeddings;
    if (int rc = check_dims(fn, D, C, L, max_level, dtype)) return rc;
    N2M_REQUIRE(grad && inputs && offsets && grad_embeddings, N2M_ENULL, "%s: NULL tensor", fn);
    N2M_REQUIRE((dy_dx == nullptr) == (grad_inputs == nullptr), N2M_EINVAL, "%s: dy_dx and grad_inputs go together", fn);
    N2M_REQUIRE(!(dtype == N2M_F16 && (C & 1u)), N2M_EUNSUPPORTED,
                "%s: fp16 tables with odd C are not supported (the reference's atomicAdd(at::Half*) is an empty stub, "
                "gridencoder.cu:22-26, and grid.py:45 never selects it)", fn);
    if (B == 0 || max_level == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const size_t esz = dtype == N2M_F16 ? 2 : 4;
    BwdArgs a{grad, inputs, offsets, grad_embeddings, B, L, max_level, make_levels(L, S, H), gridtype, align_corners != 0, interp, s,
              dy_dx, grad_inputs};
    const double bytes = (double)B * (4.0 * D + (double)max_level * C * esz + 2.0 * max_level * (1u << D) * C * esz);
    N2M_PROF(N2M_K_GRID_BWD, s, bytes);
    if (dtype == N2M_F16) { N2M_DISPATCH_DC(D, C, BWD_CASE, _Float16, BM, a); }
    else { N2M_DISPATCH_DC(D, C, BWD_CASE, float, BM, a); }
    N2M_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// ================================================================================================== C ABI

extern "C" int n2m_grid_encode_forward(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs,
                                       uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level, float S,
                                       uint32_t H, void* dy_dx, uint32_t gridtype, int align_corners, uint32_t interp,
                                       int dtype, void* stream) {
    return forward_any<false>(inputs, embeddings, offsets, outputs, B, D, C, L, max_level, S, H, dy_dx, gridtype, align_corners,
                              interp, dtype, stream, "grid_encode_forward");
}

extern "C" int n2m_grid_encode_forward_bm(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs,
                                          uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level, float S,
                                          uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, int dtype,
                                          void* stream) {
    return forward_any<true>(inputs, embeddings, offsets, outputs, B, D, C, L, max_level, S, H, nullptr, gridtype, align_corners,
                             interp, dtype, stream, "grid_encode_forward_bm");
}

extern "C" int n2m_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                                        void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                        uint32_t max_level, float S, uint32_t H, const void* dy_dx, void* grad_inputs,
                                        uint32_t gridtype, int align_corners, uint32_t interp, int dtype, void* stream) {
    return backward_any<false>(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, max_level, S, H, dy_dx, grad_inputs,
                               gridtype, align_corners, interp, dtype, stream, "grid_encode_backward");
}

extern "C" int n2m_grid_encode_backward_bm(const void* grad, const float* inputs, const void* embeddings,
                                           const int32_t* offsets, void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                                           uint32_t L, uint32_t max_level, float S, uint32_t H, uint32_t gridtype,
                                           int align_corners, uint32_t interp, int dtype, void* stream) {
    return backward_any<true>(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, max_level, S, H, nullptr, nullptr,
                              gridtype, align_corners, interp, dtype, stream, "grid_encode_backward_bm");
}

extern "C" int n2m_grad_total_variation(const void* inputs, const void* embeddings, void* grad, const int32_t* offsets,
                                        float weight, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                        uint32_t gridtype, int align_corners, int dtype, void* stream) {
    if (int rc = check_dims("grad_total_variation", D, C, L, L, dtype)) return rc;
    N2M_REQUIRE(inputs && embeddings && grad && offsets, N2M_ENULL, "grad_total_variation: NULL tensor");
    N2M_REQUIRE(dtype == N2M_F32, N2M_EUNSUPPORTED,
                "grad_total_variation: fp32 tables only (the reference's fp16 path ends in the empty atomicAdd(at::Half*) "
                "stub, gridencoder.cu:22-26,606; gridencoder/grid.py:170 runs TV with autocast disabled)");
    if (B == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    TvArgs a{(const float*)inputs, (const float*)embeddings, (float*)grad, offsets, weight, B, L, make_levels(L, S, H), gridtype,
             align_corners != 0, s};
    N2M_PROF(N2M_K_GRID_TV, s, (double)B * (4.0 * D + (double)L * (1 + 2 * D) * C * 4.0 + (double)L * C * 8.0));
    N2M_DISPATCH_DC(D, C, TV_CASE, a);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" uint64_t n2m_grid_binned_workspace_bytes(uint32_t B, uint32_t D, uint32_t C, uint32_t max_level, const int32_t* host_offsets,
                                                    int dtype, int tv) {
    if (D != 3 || !host_offsets || B == 0) return 0;
    if (B > kBinChunk) B = kBinChunk;           // larger batches run in passes of kBinChunk samples over the same workspace
    if (tv ? !(dtype == N2M_F32 && C == 1) : !((dtype == N2M_F32 && C == 1) || (dtype == N2M_F16 && C == 2))) return 0;
    const BinLayout lay = make_bin_plan(B, C, max_level, host_offsets, tv != 0);
    return lay.ok ? (uint64_t)lay.bytes : 0;
}

extern "C" int n2m_grid_encode_backward_binned(const void* grad, const float* inputs, const int32_t* host_offsets, void* grad_embeddings,
                                               uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level, float S, uint32_t H,
                                               uint32_t gridtype, int align_corners, uint32_t interp, int dtype, const float* tv_embeddings,
                                               float tv_weight, float tv_weight_outer, float tv_inner01, const float* tv_scale,
                                               float* found_inf, void* workspace, uint64_t workspace_bytes, void* stream) {
    const char* fn = "grid_encode_backward_binned";
    if (int rc = check_dims(fn, D, C, L, max_level, dtype)) return rc;
    N2M_REQUIRE(grad && inputs && host_offsets && grad_embeddings && workspace, N2M_ENULL, "%s: NULL tensor", fn);
    N2M_REQUIRE(D == 3 && ((dtype == N2M_F32 && C == 1) || (dtype == N2M_F16 && C == 2)), N2M_EUNSUPPORTED,
                "%s: D=3 with fp32 C=1 or fp16 C=2 tables only (use n2m_grid_encode_backward otherwise)", fn);
    N2M_REQUIRE(!tv_embeddings || (dtype == N2M_F32 && C == 1 && max_level == L), N2M_EUNSUPPORTED,
                "%s: the fused TV term needs an fp32 C=1 table and max_level == L", fn);
    if (B == 0 || max_level == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const size_t esz = dtype == N2M_F16 ? 2 : 4;
    const LevelTable lv = make_levels(L, S, H);
    const TvParams tv{tv_embeddings, tv_weight, tv_weight_outer, tv_inner01, tv_scale, g_cfg_tv_stride.load()};
    N2M_PROF(N2M_K_GRID_BWD, s, (double)B * (4.0 * D + (double)max_level * C * esz + 2.0 * max_level * (1u << D) * C * esz +
                                             (tv_embeddings ? (double)L * (1 + 2 * D) * 4.0 : 0.0)));
    if (dtype == N2M_F16)
        return launch_binned<_Float16, 2, 0>((const _Float16*)grad, inputs, tv, (_Float16*)grad_embeddings, B, max_level, host_offsets, lv, gridtype,
                                             align_corners != 0, interp, workspace, (size_t)workspace_bytes, s, fn, found_inf);
    if (tv_embeddings)
        return launch_binned<float, 1, 2>((const float*)grad, inputs, tv, (float*)grad_embeddings, B, max_level, host_offsets, lv, gridtype,
                                          align_corners != 0, interp, workspace, (size_t)workspace_bytes, s, fn, found_inf);
    return launch_binned<float, 1, 0>((const float*)grad, inputs, tv, (float*)grad_embeddings, B, max_level, host_offsets, lv, gridtype,
                                      align_corners != 0, interp, workspace, (size_t)workspace_bytes, s, fn, found_inf);
}

extern "C" int n2m_grad_total_variation_binned(const float* inputs, const float* embeddings, float* grad, const int32_t* host_offsets,
                                               float weight, float weight_outer, float inner01, const float* weight_scale, uint32_t B,
                                               uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners,
                                               void* workspace, uint64_t workspace_bytes, void* stream) {
    const char* fn = "grad_total_variation_binned";
    if (int rc = check_dims(fn, D, C, L, L, N2M_F32)) return rc;
    N2M_REQUIRE(inputs && embeddings && grad && host_offsets && workspace, N2M_ENULL, "%s: NULL tensor", fn);
    N2M_REQUIRE(D == 3 && C == 1, N2M_EUNSUPPORTED, "%s: D=3, C=1 fp32 tables only (use n2m_grad_total_variation otherwise)", fn);
    if (B == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const LevelTable lv = make_levels(L, S, H);
    const TvParams tv{embeddings, weight, weight_outer, inner01, weight_scale, g_cfg_tv_stride.load()};
    N2M_PROF(N2M_K_GRID_TV, s, (double)B * (4.0 * D + (double)L * (1 + 2 * D) * C * 4.0 + (double)L * C * 8.0));
    return launch_binned<float, 1, 1>(nullptr, inputs, tv, grad, B, L, host_offsets, lv, gridtype, align_corners != 0, 0u, workspace,
                                      (size_t)workspace_bytes, s, fn);
}

extern "C" uint64_t n2m_grid_binned_pair_workspace_bytes(uint32_t B, uint32_t max_level, const int32_t* host_offsets) {
    if (!host_offsets || B == 0) return 0;
    if (B > kBinChunk) B = kBinChunk;
    const BinLayout lay = make_bin_plan(B, 2, max_level, host_offsets, false, kPairP, 2);
    if (!lay.ok) return 0;
    const PmLayout pl = make_pm_plan(B, lay.plan);          // the partition-major layout needs more room (regions at twice the expected fill + overflow log)
    return (uint64_t)(pl.ok && pl.bytes > lay.bytes ? pl.bytes : lay.bytes);
}

static int binned_pair_entry(const float* grad1, const void* grad2, const float* inputs, const int32_t* host_offsets,
                             float* grad_embeddings1, void* grad_embeddings2, uint32_t B, uint32_t L, uint32_t max_level,
                             float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                             const float* tv_embeddings, float tv_weight, float tv_weight_outer, float tv_inner01,
                             const float* tv_scale, float* found_inf, float in_scale, float in_offset, int overwrite,
                             void* workspace, uint64_t workspace_bytes, void* stream, int half, const float* tv_terms = nullptr,
                             const AdamFuse* fuse1 = nullptr, const AdamFuse* fuse2 = nullptr, const FoldArgs* fold = nullptr,
                             uint32_t in_level_stride = 0);

extern "C" int n2m_grid_encode_backward_binned_pair(const float* grad1, const void* grad2, const float* inputs, const int32_t* host_offsets,
                                                    float* grad_embeddings1, void* grad_embeddings2, uint32_t B, uint32_t L, uint32_t max_level,
                                                    float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                                                    const float* tv_embeddings, float tv_weight, float tv_weight_outer, float tv_inner01,
                                                    const float* tv_scale, float* found_inf, float in_scale, float in_offset, int overwrite,
                                                    void* workspace, uint64_t workspace_bytes, void* stream) {
    return binned_pair_entry(grad1, grad2, inputs, host_offsets, grad_embeddings1, grad_embeddings2, B, L, max_level, S, H, gridtype, align_corners,
                             interp, tv_embeddings, tv_weight, tv_weight_outer, tv_inner01, tv_scale, found_inf, in_scale, in_offset, overwrite,
                             workspace, workspace_bytes, stream, 0);
}

extern "C" int n2m_grid_encode_backward_binned_pair_half(const float* grad1, const void* grad2, const float* inputs, const int32_t* host_offsets,
                                                         float* grad_embeddings1, void* grad_embeddings2, uint32_t B, uint32_t L, uint32_t max_level,
                                                         float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                                                         const float* tv_embeddings, float tv_weight, float tv_weight_outer, float tv_inner01,
                                                         const float* tv_scale, float* found_inf, float in_scale, float in_offset, int overwrite,
                                                         void* workspace, uint64_t workspace_bytes, void* stream, int half) {
    N2M_REQUIRE(half == 1 || half == 2, N2M_EINVAL, "grid_encode_backward_binned_pair_half: half must be 1 (levels 8..15) or 2 (levels 0..7)");
    return binned_pair_entry(grad1, grad2, inputs, host_offsets, grad_embeddings1, grad_embeddings2, B, L, max_level, S, H, gridtype, align_corners,
                             interp, tv_embeddings, tv_weight, tv_weight_outer, tv_inner01, tv_scale, found_inf, in_scale, in_offset, overwrite,
                             workspace, workspace_bytes, stream, half);
}

static int binned_pair_entry(const float* grad1, const void* grad2, const float* inputs, const int32_t* host_offsets,
                             float* grad_embeddings1, void* grad_embeddings2, uint32_t B, uint32_t L, uint32_t max_level,
                             float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                             const float* tv_embeddings, float tv_weight, float tv_weight_outer, float tv_inner01,
                             const float* tv_scale, float* found_inf, float in_scale, float in_offset, int overwrite,
                             void* workspace, uint64_t workspace_bytes, void* stream, int half, const float* tv_terms, const AdamFuse* fuse1,
                             const AdamFuse* fuse2, const FoldArgs* fold, uint32_t in_level_stride) {
    const char* fn = "grid_encode_backward_binned_pair";
    if (int rc = check_dims(fn, 3, 2, L, max_level, N2M_F16)) return rc;
    N2M_REQUIRE(inputs && host_offsets && workspace && (grad1 || grad2), N2M_ENULL, "%s: NULL tensor", fn);
    N2M_REQUIRE((grad1 == nullptr) == (grad_embeddings1 == nullptr), N2M_ENULL, "%s: grad1 and grad_embeddings1 come together (both NULL: colour table only)", fn);
    N2M_REQUIRE((grad2 == nullptr) == (grad_embeddings2 == nullptr), N2M_ENULL, "%s: grad2 and grad_embeddings2 come together (both NULL: density table only)", fn);
    N2M_REQUIRE(grad1 != nullptr || tv_embeddings == nullptr, N2M_EINVAL, "%s: the TV term rides on the density table's entries", fn);
    N2M_REQUIRE(!tv_embeddings || max_level == L, N2M_EUNSUPPORTED, "%s: the fused TV term needs max_level == L", fn);
    hipStream_t s = (hipStream_t)stream;
    if (B == 0 || max_level == 0) {
        if (overwrite && host_offsets[L] > 0 && half != 2) {
            if (grad_embeddings1) N2M_HIP(hipMemsetAsync(grad_embeddings1, 0, (size_t)host_offsets[L] * sizeof(float), s));
            if (grad_embeddings2) N2M_HIP(hipMemsetAsync(grad_embeddings2, 0, (size_t)host_offsets[L] * 2u * sizeof(_Float16), s));
        }
        return 0;
    }
    const LevelTable lv = make_levels(L, S, H);
    const TvParams tv{tv_embeddings, tv_weight, tv_weight_outer, tv_inner01, tv_scale, g_cfg_tv_stride.load()};
    // algorithmic bytes of BOTH encoders' backward + the TV stencil reads (a half call: its eight levels)
    const double lvls = half ? 8.0 : (double)max_level;
    const double esz = (grad1 && grad2) ? 8.0 : 4.0;       // bytes per (vertex, table set): fp32 C=1 + fp16 C=2, or one of them alone
    N2M_REQUIRE(!(tv_terms && tv_embeddings), N2M_EINVAL, "%s: either the TV table (computed in place) or precomputed TV terms", fn);
    N2M_REQUIRE(!tv_terms || (grad1 && max_level == L), N2M_EINVAL, "%s: TV terms ride on the density table's entries and need max_level == L", fn);
    // (+ the TV stencil's reads when it is evaluated in place; precomputed terms: 4 B per (sample, level), the stencil is n2m_grid_tv_terms')
    // (kernel-attached events on the partition-major path, whose launches go through N2M_LAUNCH; marker events around the tile-major one)
    N2mProfScope prof_scope__(N2M_K_GRID_BWD, s, (double)B * (12.0 + lvls * esz + 2.0 * lvls * 8 * esz + (tv_embeddings ? lvls * 7 * 4.0 : 0.0) + (tv_terms ? lvls * 4.0 : 0.0)),
                              pm_enabled() && !fuse1 && !fuse2);
    const int rc = launch_binned_pair(grad1, (const _Float16*)grad2, inputs, tv, grad_embeddings1, (_Float16*)grad_embeddings2, B, max_level, host_offsets, lv,
                                      gridtype, align_corners != 0, interp, workspace, (size_t)workspace_bytes, s, fn, found_inf, in_scale, in_offset, overwrite != 0, L,
                                      half, tv_terms, fuse1, fuse2, fold, in_level_stride);
    if (g_mid_event) {                                      // a path without a point between fill and accumulate: behind the call
        (void)hipEventRecord(g_mid_event, s);
        g_mid_event = nullptr;
    }
    return rc;
}

// ---- SDF recipe: which (copy, level) pairs fold into their centre sample (FoldArgs), and the others as one compact list per level
namespace {
constexpr uint32_t kFoldThreads = 512u;                    // 2^18 samples = 512 workgroups: two per CU, no half-empty second round
__global__ void __launch_bounds__(kFoldThreads)
sdf_fold_plan_kernel(const float* __restrict__ xyz, uint32_t M, float eps, float bound, uint32_t max_level, uint32_t L, LevelTable lv, bool align_corners,
                     uint8_t* __restrict__ flags /*[L, M]*/, float* __restrict__ left_pts01 /*[L, cap, 3]*/, uint32_t* __restrict__ left_src /*[L, cap]*/,
                     uint32_t cap, uint32_t* __restrict__ counters /*[2][kMaxLevels]*/, uint32_t parity) {
    // two barriers per workgroup, not three per level (the first version: 54 us for 2^18 samples): every wave scans all levels' counts first
    // (lane totals in LDS), then thread l places level l's sixteen wave totals with ONE atomic, then everybody writes its entries
    constexpr uint32_t kFoldLevels = 16u;                  // (loops unrolled over a constant bound: per-level values stay in registers)
    constexpr uint32_t kWaves = kFoldThreads / 64u;
    __shared__ uint32_t wave_tot[kFoldLevels][kWaves], wave_base[kFoldLevels][kWaves];
    const uint32_t tid = threadIdx.x, m = blockIdx.x * kFoldThreads + tid, lane = tid & 63u, wid = tid >> 6;
    if (blockIdx.x == 0u && tid < kMaxLevels) counters[(parity ^ 1u) * kMaxLevels + tid] = 0u;      // the other step's counters: nobody reads them any more
    float x[3] = {0.f, 0.f, 0.f}, p01[6], c01[3] = {2.f, 2.f, 2.f};
    bool centre_in = false;
    if (m < M) {
#pragma unroll
        for (uint32_t a = 0; a < 3; ++a) {
            x[a] = xyz[3 * (size_t)m + a];
            // the centre as the table backward maps it (x * in_scale + in_offset == (x + bound) / (2 bound) for the power-of-two bounds it accepts)
            c01[a] = (x[a] + bound) / (2.0f * bound);
        }
        centre_in = !outside_unit_cube<3>(c01) && fabsf(x[0]) <= bound && fabsf(x[1]) <= bound && fabsf(x[2]) <= bound;
    }
#pragma unroll
    for (uint32_t c = 0; c < 6; ++c) {
        const float pw = fminf(fmaxf(x[c >> 1] + ((c & 1u) ? -eps : eps), -bound), bound);      // n2m_sdf_offsets' own expression
        p01[c] = (pw + bound) / (2.0f * bound);
    }
    const float half = align_corners ? 0.0f : 0.5f;
    unsigned long long fl_all[2] = {0ull, 0ull};          // 6 bits per level, levels 0..9 | 10..15
    uint32_t before[kFoldLevels];                          // this thread's offset inside its wave's part of each level's list
#pragma unroll
    for (uint32_t l = 0; l < kFoldLevels; ++l) {
        before[l] = 0u;
        if (l >= max_level) {                              // (block-uniform) a level above the active ones folds nothing: its flags are DEFINED as zero --
            if (l < L && m < M) flags[(size_t)l * M + m] = 0u;      // a fold call over all L levels (TV on every level while the levels are progressive) reads them
            continue;
        }
        const float scale = lv.scale[l];
        uint32_t fl = 0u;
        if (centre_in) {
#pragma unroll
            for (uint32_t c = 0; c < 6; ++c)
                if (floorf(p01[c] * scale + half) == floorf(c01[c >> 1] * scale + half)) fl |= 1u << c;
        }
        const uint32_t n_left = m < M ? 6u - (uint32_t)__builtin_popcount(fl) : 0u;
        if (m < M) flags[(size_t)l * M + m] = (uint8_t)fl;
        fl_all[l / 10u] |= (unsigned long long)fl << (6u * (l % 10u));
        const uint32_t incl = n2m_wave_scan_add_u32(n_left, 0);
        before[l] = incl - n_left;
        if (lane == 63u) wave_tot[l][wid] = incl;
    }
    __syncthreads();
    if (tid < max_level) {                                 // level tid: its sixteen wave totals, one atomic for the workgroup
        uint32_t run = 0u;
        for (uint32_t w = 0; w < kWaves; ++w) { wave_base[tid][w] = run; run += wave_tot[tid][w]; }
        const uint32_t base = run != 0u ? atomicAdd(counters + parity * kMaxLevels + tid, run) : 0u;
        for (uint32_t w = 0; w < kWaves; ++w) wave_base[tid][w] += base;
    }
    __syncthreads();
    if (m >= M) return;
    // the lists (order: by workgroup arrival -- the sums they feed are order-free fixed point or float atomics)
#pragma unroll
    for (uint32_t l = 0; l < kFoldLevels; ++l) {
        if (l >= max_level) continue;
        const uint32_t fl = (uint32_t)(fl_all[l / 10u] >> (6u * (l % 10u))) & 63u;
        if (fl == 63u) continue;
        uint32_t j = wave_base[l][wid] + before[l];
#pragma unroll
        for (uint32_t c = 0; c < 6; ++c) {
            if ((fl >> c) & 1u) continue;
            float* dst = left_pts01 + ((size_t)l * cap + j) * 3u;
#pragma unroll
            for (uint32_t a = 0; a < 3; ++a)      // the other axes: n2m_sdf_offsets clamps them too
                dst[a] = a == (c >> 1) ? p01[c] : (fminf(fmaxf(x[a] + 0.0f, -bound), bound) + bound) / (2.0f * bound);
            left_src[(size_t)l * cap + j] = m * 6u + c;
            ++j;
        }
    }
}

// columns of the listed copies, and the lists padded to a common length B with points outside the unit cube (the fill drops those)
__global__ void __launch_bounds__(256)
sdf_fold_gather_kernel(const float* __restrict__ grad6, uint32_t stride6, const uint32_t* __restrict__ left_src, float* __restrict__ left_pts01,
                       uint32_t cap, const uint32_t* __restrict__ counters, uint32_t B, float* __restrict__ out) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x, l = blockIdx.y;
    if (j >= B) return;
    const uint32_t K = counters[l];
    if (j < K) out[(size_t)l * B + j] = grad6[(size_t)l * stride6 + left_src[(size_t)l * cap + j]];
    else {
        out[(size_t)l * B + j] = 0.0f;
        float* dst = left_pts01 + ((size_t)l * cap + j) * 3u;
        dst[0] = dst[1] = dst[2] = 2.0f;
    }
}
}  // namespace

extern "C" int n2m_sdf_fold_plan(const float* xyz, uint32_t M, float eps, float bound, uint32_t L, uint32_t max_level, float S, uint32_t H,
                                 int align_corners, uint8_t* flags, float* left_pts01, uint32_t* left_src, uint32_t cap, uint32_t* counters,
                                 uint32_t parity, void* stream) {
    const char* fn = "sdf_fold_plan";
    N2M_REQUIRE(xyz && flags && left_pts01 && left_src && counters, N2M_ENULL, "%s: NULL tensor", fn);
    N2M_REQUIRE(L >= 1 && L <= kMaxLevels && max_level <= L && max_level <= 16u && parity <= 1u && eps > 0.0f && bound > 0.0f && (uint64_t)cap >= 6ull * M, N2M_EINVAL,
                "%s: bad arguments (cap must hold all 6 M copies)", fn);
    if (M == 0) return 0;
    sdf_fold_plan_kernel<<<n2m_ceil_div(M, kFoldThreads), kFoldThreads, 0, (hipStream_t)stream>>>(xyz, M, eps, bound, max_level, L, make_levels(L, S, H), align_corners != 0,
                                                                                 flags, left_pts01, left_src, cap, counters, parity);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_sdf_fold_gather(const float* grad6, uint32_t M, uint32_t levels, const uint32_t* left_src, float* left_pts01, uint32_t cap,
                                   const uint32_t* counters, uint32_t B, float* out, void* stream) {
    N2M_REQUIRE(grad6 && left_src && left_pts01 && counters && out, N2M_ENULL, "sdf_fold_gather: NULL tensor");
    N2M_REQUIRE(B <= cap, N2M_EINVAL, "sdf_fold_gather: B %u exceeds the lists' capacity %u", B, cap);
    if (M == 0 || levels == 0 || B == 0) return 0;
    sdf_fold_gather_kernel<<<dim3(n2m_ceil_div(B, 256), levels), 256, 0, (hipStream_t)stream>>>(grad6, 6u * M, left_src, left_pts01, cap, counters, B, out);
    N2M_CHECK_LAUNCH();
    return 0;
}

// density table alone, every level with its OWN point list (inputs [L, level_stride, 3], grad1 [L, B]; points outside the unit cube pad a list)
extern "C" int n2m_grid_encode_backward_binned_lists(const float* grad1, const float* inputs, uint32_t level_stride, const int32_t* host_offsets,
                                                     float* grad_embeddings1, uint32_t B, uint32_t L, uint32_t max_level, float S, uint32_t H,
                                                     uint32_t gridtype, int align_corners, uint32_t interp, float* found_inf, int overwrite,
                                                     void* workspace, uint64_t workspace_bytes, void* stream) {
    N2M_REQUIRE(grad1 && level_stride >= B && level_stride > 0, N2M_EINVAL, "grid_encode_backward_binned_lists: needs the gradient and level_stride >= B");
    return binned_pair_entry(grad1, nullptr, inputs, host_offsets, grad_embeddings1, nullptr, B, L, max_level, S, H, gridtype, align_corners, interp,
                             nullptr, 0.0f, 0.0f, 1.0f, nullptr, found_inf, 1.0f, 0.0f, overwrite, workspace, workspace_bytes, stream, 0, nullptr, nullptr,
                             nullptr, nullptr, level_stride * 3u);
}

extern "C" int n2m_grid_encode_backward_binned_pair_fold(const float* grad1, const void* grad2, const float* inputs, const int32_t* host_offsets,
                                                         float* grad_embeddings1, void* grad_embeddings2, uint32_t B, uint32_t L, uint32_t max_level,
                                                         float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                                                         const float* tv_embeddings, float tv_weight, float tv_weight_outer, float tv_inner01,
                                                         const float* tv_scale, float* found_inf, float in_scale, float in_offset, int overwrite,
                                                         void* workspace, uint64_t workspace_bytes, const uint8_t* fold_flags, const float* fold_grad6,
                                                         float fold_eps, float fold_bound, void* stream) {
    const char* fn = "grid_encode_backward_binned_pair_fold";
    N2M_REQUIRE(fold_flags && fold_grad6 && grad1 && fold_eps > 0.0f && fold_bound > 0.0f, N2M_EINVAL, "%s: needs flags, the copies' gradient and the density gradient", fn);
    // the plan compares cells of (x + bound) / (2 bound): the fill must map its samples the same way
    N2M_REQUIRE(in_scale == 1.0f / (2.0f * fold_bound) && in_offset == 0.5f, N2M_EINVAL, "%s: in_scale / in_offset must be the map of bound %g", fn, (double)fold_bound);
    FoldArgs fo{};
    fo.flags = fold_flags; fo.grad6 = fold_grad6; fo.raw = inputs; fo.eps = fold_eps; fo.bound = fold_bound;
    return binned_pair_entry(grad1, grad2, inputs, host_offsets, grad_embeddings1, grad_embeddings2, B, L, max_level, S, H, gridtype, align_corners,
                             interp, tv_embeddings, tv_weight, tv_weight_outer, tv_inner01, tv_scale, found_inf, in_scale, in_offset, overwrite, workspace,
                             workspace_bytes, stream, 0, nullptr, nullptr, nullptr, &fo);
}

// ---- the table backward with the optimizer pass of the hashed levels inside the accumulates' flush (see AdamFuse)
extern "C" int n2m_grid_pair_fuse_plan(uint32_t max_samples, uint32_t L, const int32_t* host_offsets, uint32_t* first_level, uint32_t* first_row) {
    N2M_REQUIRE(host_offsets && first_level && first_row && max_samples > 0 && max_samples <= kBinChunk, N2M_EINVAL, "grid_pair_fuse_plan: bad arguments");
    const BinLayout lay = make_bin_plan(max_samples, 2, L, host_offsets, false, kPairP, 2);
    N2M_REQUIRE(lay.ok, N2M_EUNSUPPORTED, "grid_pair_fuse_plan: table layout not supported by the binned path");
    uint32_t fl = L;
    for (uint32_t l = L; l-- > 0;) {
        const uint32_t pairs = (lay.plan.parts[l] + 1u) / 2u;
        const uint64_t per_item = (uint64_t)8 * max_samples / pairs;
        const uint32_t g1 = (uint32_t)((per_item + 65535u) / 65536u);
        if (g1 > 1u || lay.plan.groups[l] > 1u) break;        // same rule as launch_binned_pair
        fl = l;
    }
    *first_level = fl;
    *first_row = fl < L ? (uint32_t)host_offsets[fl] : (uint32_t)host_offsets[L];
    return 0;
}

namespace {
__global__ void __launch_bounds__(256)
adam_fuse_restore_kernel(const float* __restrict__ found_inf, AdamFuse f1, AdamFuse f2, uint32_t first_row, uint32_t n_rows) {
    // rows below first_row: n2m_adam_step has updated them IN PLACE in the live buffers (or left them alone on a skipped step) -- the other
    // buffers, live from the next step on, get a copy either way (6 % of the rows).  Rows from first_row on: the flush has written the other
    // buffers; on a skipped step (GradScaler) they must hold the OLD state again, and the packed rows the lookup reads the old parameters.
    const uint32_t end = *found_inf != 0.0f ? n_rows : first_row;
    for (uint32_t r = blockIdx.x * 256u + threadIdx.x; r < end; r += gridDim.x * 256u) {
        const float p1 = f1.p_in[r];
        f1.p_out[r] = p1; f1.m_out[r] = f1.m_in[r]; f1.v_out[r] = f1.v_in[r];
        const float2 p2 = *reinterpret_cast<const float2*>(f2.p_in + 2u * (size_t)r);
        *reinterpret_cast<float2*>(f2.p_out + 2u * (size_t)r) = p2;
        *reinterpret_cast<float2*>(f2.m_out + 2u * (size_t)r) = *reinterpret_cast<const float2*>(f2.m_in + 2u * (size_t)r);
        *reinterpret_cast<float2*>(f2.v_out + 2u * (size_t)r) = *reinterpret_cast<const float2*>(f2.v_in + 2u * (size_t)r);
        if (r >= first_row) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            h2 c; c.x = (_Float16)p2.x; c.y = (_Float16)p2.y;
            *reinterpret_cast<uint2*>(f1.packed + (size_t)r * 2u) = make_uint2(__float_as_uint(p1), __builtin_bit_cast(uint32_t, c));
        }
    }
}

int make_fuse(const N2mAdamFuse* d, AdamFuse& f1, AdamFuse& f2, const char* fn) {
    N2M_REQUIRE(d && d->packed && d->bias, N2M_ENULL, "%s: NULL fuse descriptor / packed table / bias", fn);
    for (int t = 0; t < 2; ++t)
        N2M_REQUIRE(d->p_in[t] && d->m_in[t] && d->v_in[t] && d->p_out[t] && d->m_out[t] && d->v_out[t] && d->p_in[t] != d->p_out[t], N2M_ENULL,
                    "%s: the fused optimizer pass needs live and shadow buffers of parameter and moments (distinct)", fn);
    AdamFuse* f[2] = {&f1, &f2};
    for (int t = 0; t < 2; ++t) {
        f[t]->p_in = d->p_in[t]; f[t]->m_in = d->m_in[t]; f[t]->v_in = d->v_in[t];
        f[t]->p_out = d->p_out[t]; f[t]->m_out = d->m_out[t]; f[t]->v_out = d->v_out[t];
        f[t]->packed = (uint32_t*)d->packed; f[t]->first_level = d->first_level;
        f[t]->lr = d->lr[t]; f[t]->beta1 = (float)d->beta1; f[t]->beta2 = (float)d->beta2;
        f[t]->omb1 = (float)(1.0 - d->beta1); f[t]->omb2 = (float)(1.0 - d->beta2); f[t]->eps = d->eps;
        f[t]->scale = d->scale; f[t]->bias = d->bias; f[t]->slot = (uint32_t)d->slot[t];
    }
    return 0;
}
}  // namespace

extern "C" int n2m_grid_encode_backward_binned_pair_adam(const float* grad1, const void* grad2, const float* inputs, const int32_t* host_offsets,
                                                         float* grad_embeddings1, void* grad_embeddings2, uint32_t B, uint32_t L, uint32_t max_level,
                                                         float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                                                         const float* tv_embeddings, float tv_weight, float tv_weight_outer, float tv_inner01,
                                                         const float* tv_scale, float* found_inf, float in_scale, float in_offset,
                                                         void* workspace, uint64_t workspace_bytes, const N2mAdamFuse* fuse, void* stream) {
    const char* fn = "grid_encode_backward_binned_pair_adam";
    N2M_REQUIRE(host_offsets && fuse && max_level == L && B > 0 && found_inf, N2M_EINVAL, "%s: needs all levels, a non-empty batch and found_inf", fn);
    N2M_REQUIRE(fuse->first_level < L, N2M_EINVAL, "%s: no level takes the fused pass (first_level %u)", fn, fuse->first_level);
    AdamFuse f1{}, f2{};
    if (int rc = make_fuse(fuse, f1, f2, fn)) return rc;
    return binned_pair_entry(grad1, grad2, inputs, host_offsets, grad_embeddings1, grad_embeddings2, B, L, max_level, S, H, gridtype, align_corners,
                             interp, tv_embeddings, tv_weight, tv_weight_outer, tv_inner01, tv_scale, found_inf, in_scale, in_offset, 1, workspace,
                             workspace_bytes, stream, 0, nullptr, &f1, &f2);
}

extern "C" int n2m_adam_fuse_restore(const N2mAdamFuse* fuse, const int32_t* host_offsets, uint32_t L, const float* found_inf, void* stream) {
    const char* fn = "adam_fuse_restore";
    N2M_REQUIRE(host_offsets && fuse && found_inf && fuse->first_level < L, N2M_EINVAL, "%s: bad arguments", fn);
    AdamFuse f1{}, f2{};
    const uint32_t first_row = (uint32_t)host_offsets[fuse->first_level];
    if (int rc = make_fuse(fuse, f1, f2, fn)) return rc;
    adam_fuse_restore_kernel<<<n2m_ceil_div(first_row > 0 ? first_row : 1u, 256u), 256, 0, (hipStream_t)stream>>>(found_inf, f1, f2, first_row,
                                                                                                              (uint32_t)host_offsets[L]);
    N2M_CHECK_LAUNCH();
    return 0;
}

// The same call with the TV terms of the batch precomputed by n2m_grid_tv_terms (tv_terms [L, B] f32): the fill adds tv_terms[level, s] to
// vertex 000's entry instead of gathering the stencil itself.  Same bits as the tv_embeddings form (one device function computes the term).
extern "C" int n2m_grid_encode_backward_binned_pair_tvt(const float* grad1, const void* grad2, const float* inputs, const int32_t* host_offsets,
                                                        float* grad_embeddings1, void* grad_embeddings2, uint32_t B, uint32_t L, uint32_t max_level,
                                                        float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                                                        const float* tv_terms, float* found_inf, float in_scale, float in_offset, int overwrite,
                                                        void* workspace, uint64_t workspace_bytes, void* stream, int half) {
    N2M_REQUIRE(half >= 0 && half <= 2, N2M_EINVAL, "grid_encode_backward_binned_pair_tvt: half must be 0 (all levels), 1 (8..15) or 2 (0..7)");
    return binned_pair_entry(grad1, grad2, inputs, host_offsets, grad_embeddings1, grad_embeddings2, B, L, max_level, S, H, gridtype, align_corners,
                             interp, nullptr, 0.0f, 0.0f, 1.0f, nullptr, found_inf, in_scale, in_offset, overwrite, workspace, workspace_bytes, stream,
                             half, tv_terms);
}

// TV terms of a batch: tv_out[level, s] for B samples, all L levels (what the shared fill would add to vertex 000's entry of the density table).
extern "C" int n2m_grid_tv_terms(const float* inputs, const float* tv_embeddings, const int32_t* host_offsets, uint32_t B, uint32_t L, float S,
                                 uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, float tv_weight, float tv_weight_outer,
                                 float tv_inner01, const float* tv_scale, float in_scale, float in_offset, float* tv_out, void* stream) {
    const char* fn = "grid_tv_terms";
    if (int rc = check_dims(fn, 3, 1, L, L, N2M_F32)) return rc;
    N2M_REQUIRE(inputs && tv_embeddings && host_offsets && tv_out, N2M_ENULL, "%s: NULL tensor", fn);
    N2M_REQUIRE(B <= kBinChunk, N2M_EUNSUPPORTED, "%s: at most %u samples per call", fn, kBinChunk);
    if (B == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const BinLayout lay = make_bin_plan(B, 2, L, host_offsets, false, kPairP, 2);
    N2M_REQUIRE(lay.ok, N2M_EUNSUPPORTED, "%s: table layout not supported", fn);
    const LevelTable lv = make_levels(L, S, H);
    const TvParams tv{tv_embeddings, tv_weight, tv_weight_outer, tv_inner01, tv_scale, g_cfg_tv_stride.load()};
    N2M_PROF(N2M_K_GRID_TV, s, (double)B * (12.0 + (double)L * 7 * 4.0 + (double)L * 4.0));
    tv_terms_kernel<<<dim3(n2m_ceil_div(B, 256), L), 256, 0, s>>>(inputs, tv, B, B, lay.plan, lv, gridtype, align_corners != 0, interp, in_scale, in_offset, tv_out);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_grid_encode_forward_pair(const float* inputs, const float* embeddings1, const void* embeddings2, const int32_t* offsets,
                                            float* outputs1, void* outputs2, uint32_t B, uint32_t L, uint32_t max_level, float S, uint32_t H,
                                            uint32_t gridtype, int align_corners, uint32_t interp, float in_scale, float in_offset,
                                            void* stream) {
    const char* fn = "grid_encode_forward_pair";
    if (int rc = check_dims(fn, 3, 2, L, max_level, N2M_F16)) return rc;
    N2M_REQUIRE(inputs && embeddings1 && embeddings2 && offsets && outputs1 && outputs2, N2M_ENULL, "%s: NULL tensor", fn);
    if (B == 0 || max_level == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const LevelTable lv = make_levels(L, S, H);
    // algorithmic bytes of both encoders' forward (588 B/sample each at L = 16)
    N2M_PROF(N2M_K_GRID_FWD, s, (double)B * (12.0 + (double)max_level * 8 * (4 + 4) + (double)max_level * (4 + 4)));
    const uint32_t n_tiles = n2m_ceil_div(B, 256);
    grid_forward3_pair_kernel<<<n_tiles * max_level, 256, 0, s>>>(inputs, embeddings1, (const _Float16*)embeddings2, offsets, outputs1,
                                                                 (_Float16*)outputs2, B, max_level, lv, gridtype, align_corners != 0, interp, n_tiles,
                                                                 in_scale, in_offset);
    N2M_CHECK_LAUNCH();
    return 0;
}

static int forward_packed(const float* inputs, const void* packed, const int32_t* offsets, float* outputs1, void* outputs2, uint32_t B, uint32_t L,
                          uint32_t max_level, float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, float in_scale, float in_offset,
                          void* stream, uint32_t level_begin, uint32_t n_levels, float* tv_corners = nullptr, const TvParams* tvp = nullptr,
                          float* tv_out = nullptr);

extern "C" int n2m_grid_encode_forward_packed(const float* inputs, const void* packed, const int32_t* offsets, float* outputs1, void* outputs2,
                                              uint32_t B, uint32_t L, uint32_t max_level, float S, uint32_t H, uint32_t gridtype,
                                              int align_corners, uint32_t interp, float in_scale, float in_offset, void* stream) {
    return forward_packed(inputs, packed, offsets, outputs1, outputs2, B, L, max_level, S, H, gridtype, align_corners, interp, in_scale, in_offset, stream, 0u,
                          max_level);
}

// Levels [level_begin, level_begin + n_levels) only: the rows of outputs1 / outputs2 of those levels, bit-identical to the full call's.  For a
// caller whose packed rows arrive in level chunks (sharded optimizer: each rank refreshes its own rows, the others' come by all-gather): the
// lookup of the coarse half runs while the fine half's rows are still on the wire.
extern "C" int n2m_grid_encode_forward_packed_levels(const float* inputs, const void* packed, const int32_t* offsets, float* outputs1, void* outputs2,
                                                     uint32_t B, uint32_t L, uint32_t max_level, float S, uint32_t H, uint32_t gridtype,
                                                     int align_corners, uint32_t interp, float in_scale, float in_offset, uint32_t level_begin,
                                                     uint32_t n_levels, void* stream) {
    N2M_REQUIRE(n_levels >= 1 && level_begin + n_levels <= max_level, N2M_EINVAL,
                "grid_encode_forward_packed_levels: levels [%u, %u) outside [0, %u)", level_begin, level_begin + n_levels, max_level);
    return forward_packed(inputs, packed, offsets, outputs1, outputs2, B, L, max_level, S, H, gridtype, align_corners, interp, in_scale, in_offset, stream,
                          level_begin, n_levels);
}

static int forward_packed(const float* inputs, const void* packed, const int32_t* offsets, float* outputs1, void* outputs2, uint32_t B, uint32_t L,
                          uint32_t max_level, float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, float in_scale, float in_offset,
                          void* stream, uint32_t level_begin, uint32_t n_levels, float* tv_corners, const TvParams* tvp, float* tv_out) {
    const char* fn = "grid_encode_forward_packed";
    if (int rc = check_dims(fn, 3, 2, L, max_level, N2M_F16)) return rc;
    // outputs2 == NULL: the density encoder alone from the packed rows.  (Built for the occupancy refresh's 2 M-point query and measured
    // slower there than n2m_grid_encode_forward on the plain table, 488 against 382 us: those points are Morton-ordered cell centres whose
    // 4-byte gathers coalesce in the L1; the refresh stays on the plain table.)
    N2M_REQUIRE(inputs && packed && offsets && outputs1, N2M_ENULL, "%s: NULL tensor", fn);
    N2M_REQUIRE(((uintptr_t)packed & 15u) == 0, N2M_EINVAL, "%s: the packed table must be 16-byte aligned", fn);
    if (B == 0 || max_level == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const LevelTable lv = make_levels(L, S, H);
    // algorithmic bytes of both encoders' forward (588 B/sample each at L = 16, one 12-byte input read shared)
    // (one scope object for the whole launch: its destructor records the closing event)
    // (+ with the TV terms: the stencil's three rows beyond the corners and the 4-byte term, per level)
    N2M_PROF_K(outputs2 ? N2M_K_GRID_FWD_PACKED : N2M_K_GRID_FWD, s,
             (outputs2 ? (double)B * (12.0 + (double)n_levels * 8 * (4 + 4) + (double)n_levels * (4 + 4))
                       : (double)B * (12.0 + (double)n_levels * 8 * 4 + (double)n_levels * 4)) + (tv_out ? (double)B * n_levels * (3 * 4.0 + 4.0) : 0.0));
    const uint32_t n_tiles = n2m_ceil_div(B, 256);
    static const uint32_t xg_env = getenv("N2M_FWD_XCD_GROUP") ? (uint32_t)atoi(getenv("N2M_FWD_XCD_GROUP")) : 4u;     // A/B switch: 0 = level-major grid
    // XCD groups: every group of 8 / xg XCD sets owns n_levels / groups levels in pairs (coarse with fine): 16 levels, or a half of 8
    const uint32_t xg = ((n_levels == 16u || n_levels == 8u) && (xg_env == 1u || xg_env == 2u || xg_env == 4u) && n_levels % (2u * (8u / xg_env)) == 0u) ? xg_env : 0u;
    const uint32_t blocks = xg ? 8u * (n_levels / (8u / xg)) * n2m_ceil_div(n_tiles, xg) : n_tiles * n_levels;
    if (tv_out) {
        TvParams tv = *tvp;
        tv.table = reinterpret_cast<const float*>(packed);      // the density column of the packed rows (stride 2), the table the lookup reads anyway
        tv.stride = 2u;
        N2M_LAUNCH(grid_forward3_packed_kernel<true>, blocks, 256, 0, s, inputs, (const uint2*)packed, offsets, outputs1, (_Float16*)outputs2, B,
                   level_begin + n_levels, lv, gridtype, align_corners != 0, interp, n_tiles, in_scale, in_offset, xg, level_begin, n_levels,
                   (float4*)nullptr, tv, tv_out);
    } else {
        N2M_LAUNCH(grid_forward3_packed_kernel<false>, blocks, 256, 0, s, inputs, (const uint2*)packed, offsets, outputs1, (_Float16*)outputs2, B,
                   level_begin + n_levels, lv, gridtype, align_corners != 0, interp, n_tiles, in_scale, in_offset, xg, level_begin, n_levels,
                   reinterpret_cast<float4*>(tv_corners), TvParams{}, (float*)nullptr);
    }
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_grid_encode_forward_packed_tv(const float* inputs, const void* packed, const int32_t* offsets, float* outputs1, void* outputs2,
                                                 uint32_t B, uint32_t L, uint32_t max_level, float S, uint32_t H, uint32_t gridtype, int align_corners,
                                                 uint32_t interp, float in_scale, float in_offset, float* tv_corners, void* stream) {
    N2M_REQUIRE(tv_corners == nullptr || ((uintptr_t)tv_corners & 15u) == 0, N2M_EINVAL, "grid_encode_forward_packed_tv: tv_corners must be 16-byte aligned");
    return forward_packed(inputs, packed, offsets, outputs1, outputs2, B, L, max_level, S, H, gridtype, align_corners, interp, in_scale, in_offset, stream,
                          0u, max_level < L ? max_level : L, tv_corners);
}

// The lookup that also leaves the FINISHED TV terms of its samples (round 6): tv_out [L, B] fp32, tv_out[level, b] = the value n2m_grid_tv_terms computes
// for (sample b, level) from the same packed table (density column) -- bit for bit -- at the cost of at most three more gathers per (sample, level):
// the stencil's centre and +x / +y / +z neighbours are corners the lookup has in registers.  n2m_grid_encode_backward_binned_pair_tvt consumes the
// terms.  tv_scale (device, may be NULL) multiplies both weights (the GradScaler factor the backward of the same step will run under).
// Reference semantics: gridencoder.cu:505-609 (kernel_grad_tv) at nerf/utils.py:800-823's samples and weights.
extern "C" int n2m_grid_encode_forward_packed_tvterms(const float* inputs, const void* packed, const int32_t* offsets, float* outputs1, void* outputs2,
                                                      uint32_t B, uint32_t L, uint32_t max_level, float S, uint32_t H, uint32_t gridtype,
                                                      int align_corners, uint32_t interp, float in_scale, float in_offset, float tv_weight,
                                                      float tv_weight_outer, float tv_inner01, const float* tv_scale, float* tv_out, void* stream) {
    N2M_REQUIRE(tv_out != nullptr, N2M_ENULL, "grid_encode_forward_packed_tvterms: NULL tv_out");
    N2M_REQUIRE(max_level == L, N2M_EINVAL, "grid_encode_forward_packed_tvterms: TV terms need max_level == L (every level of tv_out is written)");
    const TvParams tv{nullptr, tv_weight, tv_weight_outer, tv_inner01, tv_scale, 2u};
    return forward_packed(inputs, packed, offsets, outputs1, outputs2, B, L, max_level, S, H, gridtype, align_corners, interp, in_scale, in_offset, stream,
                          0u, L, nullptr, &tv, tv_out);
}

// Measurement aid: on != 0 arms the stamps of bin_fill_pair_kernel (two workgroups, tid 0); out (may be NULL, else 116 words) receives the 2 x 8 x 6 fill stamps
// and the 2 x 2 x 5 accumulate stamps (fp32 table kernel, fp16 table kernel; two work items each) of the last armed launch.  Synchronises the device.
extern "C" int n2m_debug_fill_times(int on, unsigned long long* out) {
    if (out) N2M_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fill_t), sizeof(unsigned long long) * 96));
    if (out) N2M_HIP(hipMemcpyFromSymbol(out + 96, HIP_SYMBOL(g_acc_t), sizeof(unsigned long long) * 20));
    const unsigned int v = (unsigned int)on;        // bit 0: stamps, bit 1: plain log stores (measurement)
    N2M_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_fill_timing_on), &v, sizeof(v)));
    g_fill_dbg_host = v;
    return 0;
}

extern "C" int n2m_grid_backward_peer_route(const N2mPeerRoute* route) {
    if (!route) { g_peer_route = N2mPeerRoute{}; return 0; }
    // (the LAST owner's chunk of the coarse half may be shorter: chunks are padded to a multiple of four rows so that every chunk starts 16-byte
    //  aligned in the packed table -- W = 4 and W = 8 at the standard table, where split_row / W is not a multiple of four)
    N2M_REQUIRE(route->world >= 1 && route->world <= N2M_PEER_MAX && route->rows_c > 0 && route->rows_f > 0 &&
                (uint64_t)route->world * route->rows_c >= route->split_row && (uint64_t)(route->world - 1u) * route->rows_c < route->split_row,
                N2M_EINVAL, "grid_backward_peer_route: 1..%d ranks, (world - 1) * rows_c < split_row <= world * rows_c", N2M_PEER_MAX);
    for (int h = 0; h < 2; ++h)
        for (uint32_t r = 0; r < route->world; ++r)
            N2M_REQUIRE(route->g1[h][r] && route->g2[h][r], N2M_ENULL, "grid_backward_peer_route: NULL staging slot (half %d, owner %u)", h, r);
    g_peer_route = *route;
    return 0;
}

extern "C" int n2m_grid_backward_config(int tv_stride, float overflow_div) {
    N2M_REQUIRE((tv_stride == 1 || tv_stride == 2) && overflow_div >= 1.0f, N2M_EINVAL, "grid_backward_config: tv_stride 1 or 2, overflow_div >= 1");
    g_cfg_tv_stride.v = (uint32_t)tv_stride;
    g_cfg_overflow_div.v = overflow_div;
    return 0;
}

