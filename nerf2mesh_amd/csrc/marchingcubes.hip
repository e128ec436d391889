// Marching cubes on the device: the stage-0 mesh extraction the reference does on the host with PyMCubes
// (`mcubes.marching_cubes(sigmas, density_thresh)`, nerf/renderer.py:524-527; outer cascades :563, :616) after copying the whole
// density volume over PCIe (:518).  SURVEY.md 8f-4.
//
// The volume is HBM-resident fp32 [R0][R1][R2] (C order, like the torch tensor the reference builds).  One thread per grid NODE; a node
// owns the three grid edges that leave it in +x, +y, +z (their vertices) and is the low corner of one cell (its triangles):
//   count   classify, per-workgroup sums of owned vertices / triangles, per-node word (edge mask << 29 | vertex prefix inside the
//           workgroup); two-level scan of the workgroup sums -> exclusive offsets + totals (the host reads them, allocates, like the
//           march_rays_train protocol);
//   emit    vertices in node order (then axis), triangles in cell order (then table order): deterministic, no atomics.  A triangle's
//           three vertex ids come from the owning nodes' words.
// HBM-bound byte work: 4 B read + 4 B written per node in the count pass, 8 B read per node in the emit pass (neighbour values are
// cache hits) + the mesh itself; workgroups without a crossing skip the emit pass after one load.
// Conventions (case table, solid = !(value < iso), interpolation in double with the lower corner first): tools/gen_mc_table.py.
#include <math.h>

#include "n2m_common.hpp"
#define N2M_MC_QUAL __device__
#include "mc_table.inc"      // kMcNumTris[256], kMcTris[256][3 * N2M_MC_MAX_TRIS]: device-resident constants

namespace {

constexpr uint32_t kMcBlock = 256;
constexpr uint32_t kPrefixMask = (1u << 29) - 1u;

// A thread takes VEC consecutive nodes of one row (VEC = 4 when R2 % 4 == 0: the row loads are shared -- 4 x (dwordx4 + 1) loads and ONE
// (i, j, k) computation per 4 nodes instead of 32 loads and 4 index computations, which is what bounded the first version: VALU work
// per node, not bytes; VEC = 1 otherwise).  A workgroup covers kMcBlock * VEC consecutive nodes.
template <uint32_t VEC>
struct McThread {
    float v[4][VEC + 1u];  // rows (y + 2 x) of the thread's nodes and the node after them (missing neighbours repeat the last value)
    uint32_t solid[4];     // bit z of row r: !(v[r][z] < iso)
    uint32_t i, j, k0;
    bool hx, hy, hz_last;  // neighbours exist: plane i + 1, row j + 1, the node after the thread's last one
};

// (double)v < iso for a float v  <=>  v < c with c = the smallest float >= iso (computed on the host): the classification compares floats
template <uint32_t VEC>
__device__ __forceinline__ void mc_load(const float* __restrict__ f, uint32_t n0, uint32_t R0, uint32_t R1, uint32_t R2, float iso_up, McThread<VEC>& t) {
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    const uint32_t plane = R1 * R2;
    t.i = n0 / plane;
    const uint32_t r = n0 - t.i * plane;
    t.j = r / R2;
    t.k0 = r - t.j * R2;
    t.hx = t.i + 1u < R0;
    t.hy = t.j + 1u < R1;
    t.hz_last = t.k0 + VEC < R2;
    const float* __restrict__ base = f + n0;
#pragma unroll
    for (uint32_t rr = 0; rr < 4; ++rr) {
        const float* __restrict__ row = base + ((rr & 2u) && t.hx ? plane : 0u) + ((rr & 1u) && t.hy ? R2 : 0u);
        if constexpr (VEC == 4u) {
            const f4u q = *reinterpret_cast<const f4u*>(row);
            t.v[rr][0] = q.x; t.v[rr][1] = q.y; t.v[rr][2] = q.z; t.v[rr][3] = q.w;
        } else {
            t.v[rr][0] = row[0];
        }
        t.v[rr][VEC] = t.hz_last ? row[VEC] : t.v[rr][VEC - 1u];
        uint32_t sb = 0;
#pragma unroll
        for (uint32_t z = 0; z <= VEC; ++z) sb |= (t.v[rr][z] < iso_up ? 0u : 1u) << z;
        t.solid[rr] = sb;
    }
}

// node q of the thread: owned crossed edges (bit a = the edge along dim a) and the cell case (0 when the node is no cell's low corner)
template <uint32_t VEC>
__device__ __forceinline__ void mc_node(const McThread<VEC>& t, uint32_t q, uint32_t& mask, uint32_t& cas) {
    const bool hz = q + 1u < VEC || t.hz_last;
    uint32_t bits = 0;
#pragma unroll
    for (uint32_t c = 0; c < 8; ++c) {
        const uint32_t x = c & 1u, y = (c >> 1) & 1u, z = c >> 2;
        bits |= ((t.solid[y + 2u * x] >> (q + z)) & 1u) << c;
    }
    const uint32_t s0 = bits & 1u;
    mask = ((t.hx && ((bits >> 1) & 1u) != s0) ? 1u : 0u) | ((t.hy && ((bits >> 2) & 1u) != s0) ? 2u : 0u) | ((hz && ((bits >> 4) & 1u) != s0) ? 4u : 0u);
    cas = (t.hx && t.hy && hz) ? bits : 0u;
}

// exclusive scan of (nt << 16 | nv) over the workgroup; returns the packed exclusive prefix, *total = packed workgroup sum
__device__ __forceinline__ uint32_t mc_block_scan(uint32_t packed, uint32_t* wave_tot, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t incl = n2m_wave_scan_add_u32(packed, (int)lane);
    if (lane == 63u) wave_tot[w] = incl;
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (uint32_t q = 0; q < kMcBlock / 64u; ++q) {
        const uint32_t t = wave_tot[q];
        if (q < w) off += t;
        tot += t;
    }
    *total = tot;
    return off + incl - packed;
}

template <uint32_t VEC>
__global__ void __launch_bounds__(kMcBlock)
mc_count_kernel(const float* __restrict__ field, uint32_t R0, uint32_t R1, uint32_t R2, uint32_t N, float iso_up, uint32_t* __restrict__ info,
                uint32_t* __restrict__ bsum) {
    __shared__ uint32_t wave_tot[kMcBlock / 64u];
    const uint32_t n0 = (blockIdx.x * kMcBlock + threadIdx.x) * VEC;       // N % VEC == 0: a thread's nodes are all inside or all outside
    uint32_t mask[VEC], packed = 0;
    if (n0 < N) {
        McThread<VEC> t;
        mc_load<VEC>(field, n0, R0, R1, R2, iso_up, t);
#pragma unroll
        for (uint32_t q = 0; q < VEC; ++q) {
            uint32_t cas;
            mc_node<VEC>(t, q, mask[q], cas);
            packed += ((uint32_t)kMcNumTris[cas] << 16) | (uint32_t)__popc(mask[q]);
        }
    }
    uint32_t total;
    uint32_t excl = mc_block_scan(packed, wave_tot, &total) & 0xFFFFu;     // vertices in front of this thread's nodes inside the workgroup
    if (n0 < N) {
        uint32_t w[VEC];
#pragma unroll
        for (uint32_t q = 0; q < VEC; ++q) {
            w[q] = (mask[q] << 29) | excl;
            excl += (uint32_t)__popc(mask[q]);
        }
        if constexpr (VEC == 4u) *reinterpret_cast<uint4*>(info + n0) = make_uint4(w[0], w[1], w[2], w[3]);
        else info[n0] = w[0];
    }
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;                       // triangles << 16 | vertices of this workgroup's nodes
}

// Offsets of the workgroups, two levels (a single workgroup walking all 524 288 sums of a 512^3 volume took 0.9 ms -- three times the
// count pass itself): every scan workgroup takes a chunk of kMcChunk workgroup sums, writes their exclusive prefixes inside the chunk
// (vertices and triangles) and the chunk's totals; one small workgroup then scans the chunk totals.  offset(b) = voff[b] + cv[b / chunk].
constexpr uint32_t kMcChunk = 8192;              // 1024 threads x 8 sums
__global__ void __launch_bounds__(1024)
mc_scan_chunks_kernel(const uint32_t* __restrict__ bsum, uint32_t nb, uint32_t* __restrict__ voff, uint32_t* __restrict__ toff,
                      unsigned long long* __restrict__ chunk_v, unsigned long long* __restrict__ chunk_t) {
    __shared__ unsigned long long wave_tot[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const uint32_t b0 = blockIdx.x * kMcChunk + tid * 8u;
    uint32_t x[8];
#pragma unroll
    for (uint32_t e = 0; e < 8; ++e) x[e] = (b0 + e) < nb ? bsum[b0 + e] : 0u;
    uint32_t mv = 0, mt = 0;
#pragma unroll
    for (uint32_t e = 0; e < 8; ++e) { mv += x[e] & 0xFFFFu; mt += x[e] >> 16; }
    // per chunk: vertices <= 8192 * 768 < 2^23, triangles <= 8192 * 1280 < 2^24: both prefixes fit one 64-bit scan (32 bits each)
    const uint32_t iv = n2m_wave_scan_add_u32(mv, (int)lane), it = n2m_wave_scan_add_u32(mt, (int)lane);
    if (lane == 63u) wave_tot[w] = ((unsigned long long)it << 32) | iv;
    __syncthreads();
    uint32_t ov = 0, ot = 0, tv = 0, tt = 0;
#pragma unroll
    for (uint32_t q = 0; q < 16; ++q) {
        const unsigned long long t = wave_tot[q];
        if (q < w) { ov += (uint32_t)t; ot += (uint32_t)(t >> 32); }
        tv += (uint32_t)t; tt += (uint32_t)(t >> 32);
    }
    uint32_t rv = ov + iv - mv, rt = ot + it - mt;
#pragma unroll
    for (uint32_t e = 0; e < 8; ++e) {
        if ((b0 + e) < nb) { voff[b0 + e] = rv; toff[b0 + e] = rt; }
        rv += x[e] & 0xFFFFu;
        rt += x[e] >> 16;
    }
    if (tid == 0) { chunk_v[blockIdx.x] = tv; chunk_t[blockIdx.x] = tt; }
}

// chunk totals -> exclusive chunk offsets in place (<= 1024 chunks: one value per thread), totals[0] = vertices, totals[1] = triangles
__global__ void __launch_bounds__(1024)
mc_scan_top_kernel(unsigned long long* __restrict__ chunk_v, unsigned long long* __restrict__ chunk_t, uint32_t nc,
                   unsigned long long* __restrict__ totals) {
    __shared__ unsigned long long sv[1024], st[1024];
    const uint32_t tid = threadIdx.x;
    sv[tid] = tid < nc ? chunk_v[tid] : 0ull;
    st[tid] = tid < nc ? chunk_t[tid] : 0ull;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {                            // Hillis-Steele over 1024 values: ten steps, launched once per mesh
        const unsigned long long a = tid >= d ? sv[tid - d] : 0ull, b = tid >= d ? st[tid - d] : 0ull;
        __syncthreads();
        sv[tid] += a; st[tid] += b;
        __syncthreads();
    }
    if (tid < nc) {
        chunk_v[tid] = tid ? sv[tid - 1u] : 0ull;
        chunk_t[tid] = tid ? st[tid - 1u] : 0ull;
    }
    if (tid == 0) { totals[0] = sv[1023]; totals[1] = st[1023]; }
}

template <typename VT, uint32_t VEC>
__global__ void __launch_bounds__(kMcBlock)
mc_emit_kernel(const float* __restrict__ field, uint32_t R0, uint32_t R1, uint32_t R2, uint32_t N, double iso, float iso_up,
               const uint32_t* __restrict__ info, const uint32_t* __restrict__ bsum, const uint32_t* __restrict__ voff,
               const uint32_t* __restrict__ toff, const unsigned long long* __restrict__ chunk_v, const unsigned long long* __restrict__ chunk_t,
               double div, double mul, double add, VT* __restrict__ vertices, uint32_t cap_v, int32_t* __restrict__ triangles, uint32_t cap_t) {
    __shared__ uint32_t wave_tot[kMcBlock / 64u];
    constexpr uint32_t kNodes = kMcBlock * VEC;                           // nodes per workgroup
    const uint32_t b = blockIdx.x;
    if (bsum[b] == 0u) return;                                            // nothing crosses this workgroup's nodes (block-uniform)
    const uint32_t v0 = voff[b] + (uint32_t)chunk_v[b / kMcChunk], t0 = toff[b] + (uint32_t)chunk_t[b / kMcChunk];
    const uint32_t n0 = (b * kMcBlock + threadIdx.x) * VEC;
    McThread<VEC> th;
    uint32_t mask[VEC], cas[VEC], packed = 0;
#pragma unroll
    for (uint32_t q = 0; q < VEC; ++q) { mask[q] = 0; cas[q] = 0; }
    if (n0 < N) {
        mc_load<VEC>(field, n0, R0, R1, R2, iso_up, th);
#pragma unroll
        for (uint32_t q = 0; q < VEC; ++q) {
            mc_node<VEC>(th, q, mask[q], cas[q]);
            packed += ((uint32_t)kMcNumTris[cas[q]] << 16) | (uint32_t)__popc(mask[q]);
        }
    }
    uint32_t total;
    const uint32_t excl = mc_block_scan(packed, wave_tot, &total);
    if (n0 >= N || packed == 0u) return;
    uint32_t vid = v0 + (excl & 0xFFFFu), tid_out = t0 + (excl >> 16);
    const uint32_t plane = R1 * R2;
#pragma unroll
    for (uint32_t q = 0; q < VEC; ++q) {
        // ---- vertices of the owned edges: PyMCubes' interpolation, x1 + (iso - f1) / (f2 - f1) in double, lower corner first
#pragma unroll
        for (uint32_t a = 0; a < 3; ++a) {
            if (!((mask[q] >> a) & 1u)) continue;
            const double f1 = (double)th.v[0][q], f2 = (double)(a == 0u ? th.v[2][q] : (a == 1u ? th.v[1][q] : th.v[0][q + 1u]));
            double p[3] = {(double)th.i, (double)th.j, (double)(th.k0 + q)};
            p[a] = p[a] + (iso - f1) / (f2 - f1);
            if (vid < cap_v) {
#pragma unroll
                for (uint32_t d = 0; d < 3; ++d) vertices[(size_t)vid * 3u + d] = (VT)(((p[d] / div) * mul) + add);
            }
            ++vid;
        }
        // ---- triangles of the cell
        const uint32_t nt = kMcNumTris[cas[q]];
        const unsigned char* __restrict__ row = kMcTris[cas[q]];
        const uint32_t n = n0 + q;
        for (uint32_t t = 0; t < nt; ++t, ++tid_out) {
            int32_t ids[3];
#pragma unroll
            for (uint32_t m = 0; m < 3; ++m) {
                const uint32_t e = row[3u * t + m];
                const uint32_t a = e >> 2, u = e & 1u, v = (e >> 1) & 1u;
                // the edge's lower corner: the other two coordinates in axis order (axis 0: (y, z), 1: (x, z), 2: (x, y))
                const uint32_t dx = a == 0u ? 0u : u, dy = a == 0u ? u : (a == 1u ? 0u : v), dz = a == 2u ? 0u : v;
                const uint32_t owner = n + dx * plane + dy * R2 + dz;
                const uint32_t w = info[owner];
                const uint32_t ob = owner / kNodes;
                ids[m] = (int32_t)(voff[ob] + (uint32_t)chunk_v[ob / kMcChunk] + (w & kPrefixMask) + (uint32_t)__popc((w >> 29) & ((1u << a) - 1u)));
            }
            if (tid_out < cap_t) {
                triangles[(size_t)tid_out * 3u] = ids[0];
                triangles[(size_t)tid_out * 3u + 1u] = ids[1];
                triangles[(size_t)tid_out * 3u + 2u] = ids[2];
            }
        }
    }
}

struct McLayout {
    uint32_t N, nb, nc, vec;
    size_t info, bsum, voff, toff, chunk_v, chunk_t, total;
};
McLayout mc_layout(uint32_t R0, uint32_t R1, uint32_t R2) {
    McLayout l;
    l.N = R0 * R1 * R2;
    l.vec = (R2 % 4u == 0u) ? 4u : 1u;
    l.nb = n2m_ceil_div(l.N, kMcBlock * l.vec);
    l.nc = n2m_ceil_div(l.nb, kMcChunk);              // <= 1024 for fewer than 2^31 nodes
    auto up = [](size_t v) { return (v + 255u) & ~(size_t)255u; };
    l.info = 0;
    l.bsum = up((size_t)l.N * 4u);
    l.voff = l.bsum + up((size_t)l.nb * 4u);
    l.toff = l.voff + up((size_t)l.nb * 4u);
    l.chunk_v = l.toff + up((size_t)l.nb * 4u);
    l.chunk_t = l.chunk_v + up((size_t)l.nc * 8u);
    l.total = l.chunk_t + up((size_t)l.nc * 8u);
    return l;
}

// smallest float >= iso: (double)v < iso  <=>  v < mc_iso_up(iso) for every float v
float mc_iso_up(double iso) {
    float c = (float)iso;
    if ((double)c < iso) c = nextafterf(c, INFINITY);
    return c;
}

int mc_check(const char* fn, const void* field, uint32_t R0, uint32_t R1, uint32_t R2, const void* ws, uint64_t ws_bytes) {
    N2M_REQUIRE(field != nullptr && ws != nullptr, N2M_ENULL, "%s: NULL field / workspace", fn);
    N2M_REQUIRE(R0 >= 1 && R1 >= 1 && R2 >= 1 && R0 <= 2048 && R1 <= 2048 && R2 <= 2048 && (uint64_t)R0 * R1 * R2 < (1ull << 31), N2M_EINVAL,
                "%s: volume %u x %u x %u is outside the supported range (each <= 2048, fewer than 2^31 nodes)", fn, R0, R1, R2);
    N2M_REQUIRE(ws_bytes >= mc_layout(R0, R1, R2).total, N2M_EINVAL, "%s: workspace of %llu bytes is too small (n2m_marching_cubes_workspace_bytes)", fn,
                (unsigned long long)ws_bytes);
    return 0;
}

}  // namespace

extern "C" uint64_t n2m_marching_cubes_workspace_bytes(uint32_t R0, uint32_t R1, uint32_t R2) {
    if (R0 == 0 || R1 == 0 || R2 == 0 || (uint64_t)R0 * R1 * R2 >= (1ull << 31)) return 0;
    return mc_layout(R0, R1, R2).total;
}

extern "C" int n2m_marching_cubes_count(const float* field, uint32_t R0, uint32_t R1, uint32_t R2, double iso, void* workspace,
                                        uint64_t workspace_bytes, uint64_t* totals, void* stream) {
    if (int rc = mc_check(__func__, field, R0, R1, R2, workspace, workspace_bytes)) return rc;
    N2M_NOTNULL(totals);
    N2M_REQUIRE(iso == iso, N2M_EINVAL, "marching_cubes_count: iso is NaN");
    const McLayout l = mc_layout(R0, R1, R2);
    char* ws = static_cast<char*>(workspace);
    uint32_t* info = reinterpret_cast<uint32_t*>(ws + l.info);
    uint32_t* bsum = reinterpret_cast<uint32_t*>(ws + l.bsum);
    uint32_t* voff = reinterpret_cast<uint32_t*>(ws + l.voff);
    uint32_t* toff = reinterpret_cast<uint32_t*>(ws + l.toff);
    unsigned long long* cv = reinterpret_cast<unsigned long long*>(ws + l.chunk_v);
    unsigned long long* ct = reinterpret_cast<unsigned long long*>(ws + l.chunk_t);
    hipStream_t s = (hipStream_t)stream;
    if (l.vec == 4u) mc_count_kernel<4><<<l.nb, kMcBlock, 0, s>>>(field, R0, R1, R2, l.N, mc_iso_up(iso), info, bsum);
    else mc_count_kernel<1><<<l.nb, kMcBlock, 0, s>>>(field, R0, R1, R2, l.N, mc_iso_up(iso), info, bsum);
    N2M_CHECK_LAUNCH();
    mc_scan_chunks_kernel<<<l.nc, 1024, 0, s>>>(bsum, l.nb, voff, toff, cv, ct);
    N2M_CHECK_LAUNCH();
    mc_scan_top_kernel<<<1, 1024, 0, s>>>(cv, ct, l.nc, reinterpret_cast<unsigned long long*>(totals));
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_marching_cubes_emit(const float* field, uint32_t R0, uint32_t R1, uint32_t R2, double iso, const void* workspace,
                                       uint64_t workspace_bytes, double div, double mul, double add, void* vertices, int vertices_f64,
                                       uint32_t cap_v, int32_t* triangles, uint32_t cap_t, void* stream) {
    if (int rc = mc_check(__func__, field, R0, R1, R2, workspace, workspace_bytes)) return rc;
    N2M_REQUIRE((vertices != nullptr || cap_v == 0) && (triangles != nullptr || cap_t == 0), N2M_ENULL, "marching_cubes_emit: NULL output with a non-zero capacity");
    N2M_REQUIRE(cap_v < (1u << 29), N2M_EINVAL, "marching_cubes_emit: vertex ids are 29-bit (cap_v = %u)", cap_v);
    N2M_REQUIRE(div != 0.0, N2M_EINVAL, "marching_cubes_emit: div = 0");
    if (cap_v == 0 && cap_t == 0) return 0;
    const McLayout l = mc_layout(R0, R1, R2);
    const char* ws = static_cast<const char*>(workspace);
    const uint32_t* info = reinterpret_cast<const uint32_t*>(ws + l.info);
    const uint32_t* bsum = reinterpret_cast<const uint32_t*>(ws + l.bsum);
    const uint32_t* voff = reinterpret_cast<const uint32_t*>(ws + l.voff);
    const uint32_t* toff = reinterpret_cast<const uint32_t*>(ws + l.toff);
    const unsigned long long* cv = reinterpret_cast<const unsigned long long*>(ws + l.chunk_v);
    const unsigned long long* ct = reinterpret_cast<const unsigned long long*>(ws + l.chunk_t);
    hipStream_t s = (hipStream_t)stream;
    const float up = mc_iso_up(iso);
#define N2M_MC_EMIT(VT, VEC)                                                                                                              \
    mc_emit_kernel<VT, VEC><<<l.nb, kMcBlock, 0, s>>>(field, R0, R1, R2, l.N, iso, up, info, bsum, voff, toff, cv, ct, div, mul, add,     \
                                                     static_cast<VT*>(vertices), cap_v, triangles, cap_t)
    if (vertices_f64) { if (l.vec == 4u) N2M_MC_EMIT(double, 4); else N2M_MC_EMIT(double, 1); }
    else { if (l.vec == 4u) N2M_MC_EMIT(float, 4); else N2M_MC_EMIT(float, 1); }
#undef N2M_MC_EMIT
    N2M_CHECK_LAUNCH();
    return 0;
}
