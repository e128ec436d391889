// Stand-alone laboratory for the packed hash-grid forward (north_star's lookup): variants of grid_forward3_packed_kernel on a
// synthetic ray-ordered batch, per-level times, and raw gather-rate probes of the memory hierarchy.  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/fwd_lab.hip -o tools/fwd_lab && tools/fwd_lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr uint32_t L = 16;
constexpr uint32_t P1 = 2654435761u, P2 = 805459861u;
struct Levels { float scale[L]; uint32_t res[L]; uint32_t off[L + 1]; uint32_t hashed[L]; };

static Levels make_levels() {
    Levels lv;
    const float S = log2f(exp2f(log2f(2048.0f / 16.0f) / 15.0f));
    const double pls = exp2(log2(2048.0 / 16.0) / 15.0);
    uint32_t off = 0;
    for (uint32_t l = 0; l < L; ++l) {
        const float sc = exp2f((float)l * S) * 16.0f - 1.0f;
        lv.scale[l] = sc;
        lv.res[l] = (uint32_t)ceilf(sc) + 1u;
        const uint32_t res_py = (uint32_t)ceil(16.0 * pow(pls, (double)l));
        uint64_t n = (uint64_t)(res_py + 1) * (res_py + 1) * (res_py + 1);
        if (n > (1u << 19)) n = 1u << 19;
        n = (n + 7) / 8 * 8;
        lv.off[l] = off;
        off += (uint32_t)n;
        const uint64_t dense = (uint64_t)(lv.res[l] + 1) * (lv.res[l] + 1) * (lv.res[l] + 1);
        lv.hashed[l] = dense > n;
    }
    lv.off[L] = off;
    return lv;
}

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 nt_load16(const uint2* p) { const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint2 nt_load8(const uint2* p) { const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p)); return make_uint2(v.x, v.y); }

__device__ __forceinline__ void accum(float& acc, float w, float g) { acc += w * g; }
__device__ __forceinline__ void accum(_Float16& acc, float w, _Float16 g) {
    const _Float16 p = (_Float16)(w * (float)g);
    acc = (_Float16)((float)acc + (float)p);
}

__device__ __forceinline__ uint4 ld16_unaligned(const uint2* p) {      // 16 bytes at an 8-byte-aligned address (hardware: dword alignment is enough)
    uint4 v;
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

struct Geo {
    uint32_t cell[3]; float frac[3];
};

__device__ __forceinline__ bool locate(const float* __restrict__ inputs, uint32_t b, float scale, Geo& g) {
    float x[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) x[d] = inputs[(size_t)b * 3 + d] * 0.5f + 0.5f;
    if (x[0] < 0.f || x[0] > 1.f || x[1] < 0.f || x[1] > 1.f || x[2] < 0.f || x[2] > 1.f) return false;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float p = x[d] * scale + 0.5f;
        const float fl = floorf(p);
        g.cell[d] = (uint32_t)fl;
        g.frac[d] = p - (float)g.cell[d];
    }
    return true;
}

__device__ __forceinline__ void finish(const uint2 (&g)[8], const Geo& geo, float* o1, _Float16* o2) {
    float a1 = 0.0f;
    _Float16 a2[2] = {(_Float16)0, (_Float16)0};
#pragma unroll
    for (uint32_t corner = 0; corner < 8; ++corner) {
        float w = 1.0f;
#pragma unroll
        for (uint32_t d = 0; d < 3; ++d) w *= (corner & (1u << d)) ? geo.frac[d] : 1 - geo.frac[d];
        const h2 c2 = __builtin_bit_cast(h2, g[corner].y);
        accum(a1, w, __uint_as_float(g[corner].x));
        accum(a2[0], w, c2.x);
        accum(a2[1], w, c2.y);
    }
    *o1 = a1;
    h2 r; r.x = a2[0]; r.y = a2[1];
    *reinterpret_cast<h2*>(o2) = r;
}

// MODE 0: library kernel as is.  MODE 1: nontemporal loads on hashed levels.  MODE 2: hashed odd-x partner fetched from the SAME
// 16-byte/32-byte neighbourhood when possible (unaligned 16-byte load), dense odd rows with one unaligned 16-byte load.
// ORDER 0: blockIdx = level * n_tiles + tile (level-major).  ORDER 1: blockIdx = tile * levels + level (tile-major).
template <int MODE, int ORDER>
__global__ void __launch_bounds__(256)
fwd_kernel(const float* __restrict__ inputs, const uint2* __restrict__ packed, float* __restrict__ out1, _Float16* __restrict__ out2, uint32_t B,
           Levels lv, uint32_t n_tiles, uint32_t level_lo, uint32_t level_hi) {
    const uint32_t nl = level_hi - level_lo;
    uint32_t level, tile;
    if (ORDER == 0) { level = blockIdx.x / n_tiles; tile = blockIdx.x - level * n_tiles; }
    else if (ORDER == 1) { tile = blockIdx.x / nl; level = blockIdx.x - tile * nl; }
    else {
        // XCD groups: workgroup b runs on XCD b % 8 (observed dispatch rule).  level_lo carries g = XCDs per group; group k = xcd / g owns
        // 16 / (8 / g) levels (coarse and fine mixed) and walks them one after the other, its g XCDs splitting the tiles -- each level
        // table is then pulled through g L2s instead of 8.
        const uint32_t g = level_lo, groups = 8u / g, per_group = 16u / groups;
        const uint32_t xcd = blockIdx.x & 7u, k = xcd / g, j = xcd - k * g, i = blockIdx.x >> 3;
        const uint32_t tiles_per_xcd = (n_tiles + g - 1u) / g;
        const uint32_t li = i / tiles_per_xcd;
        tile = (i - li * tiles_per_xcd) * g + j;
        if (li >= per_group || tile >= n_tiles) return;
        // li-th level of group k: pairs (p, 15 - p) dealt round-robin to the groups
        const uint32_t pair = k + groups * (li >> 1);
        level = (li & 1u) ? 15u - pair : pair;
        level_lo = 0;
    }
    level += level_lo;
    const uint32_t b = tile * 256 + threadIdx.x;
    if (b >= B) return;
    const uint32_t row0 = lv.off[level], size = lv.off[level + 1] - row0, mask = size - 1u;
    const uint2* __restrict__ tab = packed + row0;
    float* o1 = out1 + (size_t)level * B + b;
    _Float16* o2 = out2 + ((size_t)level * B + b) * 2;
    Geo geo;
    if (!locate(inputs, b, lv.scale[level], geo)) { *o1 = 0.f; h2 z = {(_Float16)0, (_Float16)0}; *reinterpret_cast<h2*>(o2) = z; return; }
    uint2 g[8];
    if (!lv.hashed[level]) {
        const uint32_t s1 = lv.res[level] + 1u, s2 = s1 * s1;
        const uint32_t base = geo.cell[0] + geo.cell[1] * s1 + geo.cell[2] * s2;
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const uint32_t r = base + ((q & 1u) ? s1 : 0u) + ((q & 2u) ? s2 : 0u);
            if (MODE == 2 || MODE == 4) {
                const uint4 v = *reinterpret_cast<const uint4 __attribute__((aligned(8)))*>(tab + r);
                g[2 * q] = make_uint2(v.x, v.y); g[2 * q + 1] = make_uint2(v.z, v.w);
            } else if ((r & 1u) == 0u) {
                const uint4 v = *reinterpret_cast<const uint4*>(tab + r);
                g[2 * q] = make_uint2(v.x, v.y); g[2 * q + 1] = make_uint2(v.z, v.w);
            } else {
                g[2 * q] = tab[r]; g[2 * q + 1] = tab[r + 1u];
            }
        }
    } else {
        const uint32_t hy0 = geo.cell[1] * P1, hy1 = hy0 + P1, hz0 = geo.cell[2] * P2, hz1 = hz0 + P2;
        const bool x_even = (geo.cell[0] & 1u) == 0u;
        uint32_t rx[4], rx1[4];
        uint4 pr[4];
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const uint32_t h = ((q & 1u) ? hy1 : hy0) ^ ((q & 2u) ? hz1 : hz0);
            rx[q] = (geo.cell[0] ^ h) & mask;
            rx1[q] = ((geo.cell[0] + 1u) ^ h) & mask;
            if (MODE == 1) pr[q] = nt_load16(tab + (rx[q] & ~1u));
            else pr[q] = *reinterpret_cast<const uint4*>(tab + (rx[q] & ~1u));
        }
        uint2 extra[4];
        if (MODE >= 3) {
            // the partner row of an odd x lies in the SAME 64-byte line as its row 7 times out of 8 (rows r and r ^ 3 / r ^ 7): issued
            // while that line is still in flight it becomes a second L2 request; issued after the pair loads have landed it hits L1
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (!x_even) {
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) {
                if (MODE == 1) extra[q] = nt_load8(tab + rx1[q]);
                else extra[q] = tab[rx1[q]];
            }
        }
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const bool odd_row = (rx[q] & 1u) != 0u;
            const uint2 lo = make_uint2(pr[q].x, pr[q].y), hi = make_uint2(pr[q].z, pr[q].w);
            g[2 * q] = odd_row ? hi : lo;
            g[2 * q + 1] = x_even ? (odd_row ? lo : hi) : extra[q];
        }
    }
    finish(g, geo, o1, o2);
}

// Two levels per thread (level, level + 8): position work shared, twice the loads in flight per lane, half the waves.
template <int DUMMY>
__global__ void __launch_bounds__(256)
fwd2_kernel(const float* __restrict__ inputs, const uint2* __restrict__ packed, float* __restrict__ out1, _Float16* __restrict__ out2, uint32_t B,
            Levels lv, uint32_t n_tiles) {
    const uint32_t lp = blockIdx.x / n_tiles, tile = blockIdx.x - lp * n_tiles;
    const uint32_t b = tile * 256 + threadIdx.x;
    if (b >= B) return;
    Geo geo[2];
    uint2 g[2][8];
    bool ok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const uint32_t level = lp + 8u * u;
        const uint32_t row0 = lv.off[level], size = lv.off[level + 1] - row0, mask = size - 1u;
        const uint2* __restrict__ tab = packed + row0;
        ok[u] = locate(inputs, b, lv.scale[level], geo[u]);
        if (!ok[u]) continue;
        if (!lv.hashed[level]) {
            const uint32_t s1 = lv.res[level] + 1u, s2 = s1 * s1;
            const uint32_t base = geo[u].cell[0] + geo[u].cell[1] * s1 + geo[u].cell[2] * s2;
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) {
                const uint32_t r = base + ((q & 1u) ? s1 : 0u) + ((q & 2u) ? s2 : 0u);
                if ((r & 1u) == 0u) {
                    const uint4 v = *reinterpret_cast<const uint4*>(tab + r);
                    g[u][2 * q] = make_uint2(v.x, v.y); g[u][2 * q + 1] = make_uint2(v.z, v.w);
                } else { g[u][2 * q] = tab[r]; g[u][2 * q + 1] = tab[r + 1u]; }
            }
        } else {
            const uint32_t hy0 = geo[u].cell[1] * P1, hy1 = hy0 + P1, hz0 = geo[u].cell[2] * P2, hz1 = hz0 + P2;
            const bool x_even = (geo[u].cell[0] & 1u) == 0u;
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) {
                const uint32_t h = ((q & 1u) ? hy1 : hy0) ^ ((q & 2u) ? hz1 : hz0);
                const uint32_t rx = (geo[u].cell[0] ^ h) & mask, rx1 = ((geo[u].cell[0] + 1u) ^ h) & mask;
                const uint4 pr = *reinterpret_cast<const uint4*>(tab + (rx & ~1u));
                const bool odd_row = (rx & 1u) != 0u;
                const uint2 lo = make_uint2(pr.x, pr.y), hi = make_uint2(pr.z, pr.w);
                g[u][2 * q] = odd_row ? hi : lo;
                g[u][2 * q + 1] = x_even ? (odd_row ? lo : hi) : tab[rx1];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const uint32_t level = lp + 8u * u;
        float* o1 = out1 + (size_t)level * B + b;
        _Float16* o2 = out2 + ((size_t)level * B + b) * 2;
        if (!ok[u]) { *o1 = 0.f; h2 z = {(_Float16)0, (_Float16)0}; *reinterpret_cast<h2*>(o2) = z; }
        else finish(g[u], geo[u], o1, o2);
    }
}

// Raw gather probe: every lane issues `per_lane` independent 16-byte loads at pseudo-random 16-byte slots of a table of `slots` slots.
template <int BYTES>
__global__ void __launch_bounds__(256) gather_probe(const uint4* __restrict__ tab, uint32_t slots_mask, uint32_t per_lane, uint32_t* __restrict__ sink) {
    uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < per_lane; i += 4) {
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s = s * 1664525u + 1013904223u;
            const uint32_t slot = (s >> 8) & slots_mask;
            if (BYTES == 16) v[k] = tab[slot];
            else { const uint2 t = reinterpret_cast<const uint2*>(tab)[slot * 2u]; v[k] = make_uint4(t.x, t.y, 0, 0); }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

static float rnd(uint32_t& s) { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 16777216.0f); }

// ray-ordered samples of a lego-like scene (union of boxes, synthetic.py's list), dt = 2 sqrt(3) / 1024
static std::vector<float> make_samples(uint32_t B) {
    static const float bx[12][6] = {{-0.55f, -0.35f, -0.45f, 0.55f, 0.35f, -0.37f}, {-0.40f, -0.22f, -0.37f, 0.30f, 0.22f, -0.12f},
                                    {-0.05f, -0.18f, -0.12f, 0.28f, 0.18f, 0.16f}, {-0.40f, -0.06f, -0.12f, -0.05f, 0.06f, 0.02f},
                                    {-0.58f, -0.04f, 0.02f, -0.20f, 0.04f, 0.10f}, {-0.60f, -0.16f, -0.20f, -0.52f, 0.16f, 0.10f},
                                    {-0.46f, -0.30f, -0.45f, -0.22f, -0.22f, -0.25f}, {-0.46f, 0.22f, -0.45f, -0.22f, 0.30f, -0.25f},
                                    {0.08f, -0.30f, -0.45f, 0.32f, -0.22f, -0.25f}, {0.08f, 0.22f, -0.45f, 0.32f, 0.30f, -0.25f},
                                    {0.02f, -0.10f, 0.16f, 0.10f, -0.02f, 0.22f}, {0.30f, -0.20f, -0.37f, 0.50f, 0.20f, -0.30f}};
    std::vector<float> x;
    x.reserve((size_t)B * 3);
    uint32_t s = 7;
    const float dt = 2.0f * 1.7320508f / 1024.0f;
    while (x.size() < (size_t)B * 3) {
        const float th = 6.2831853f * rnd(s), el = (5.0f + 75.0f * rnd(s)) * 0.0174533f;
        const float o[3] = {3.2249f * cosf(el) * cosf(th), 3.2249f * cosf(el) * sinf(th), 3.2249f * sinf(el)};
        const float tg[3] = {1.2f * rnd(s) - 0.6f, 0.8f * rnd(s) - 0.4f, 0.8f * rnd(s) - 0.45f};
        float d[3] = {tg[0] - o[0], tg[1] - o[1], tg[2] - o[2]};
        const float n = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        for (int k = 0; k < 3; ++k) d[k] /= n;
        for (float t = 2.0f + dt * rnd(s); t < 4.6f && x.size() < (size_t)B * 3; t += dt) {
            const float p[3] = {o[0] + t * d[0], o[1] + t * d[1], o[2] + t * d[2]};
            bool in = false;
            for (int b = 0; b < 12 && !in; ++b)      // occupancy cells are 1/64 wide: grow the boxes by one cell
                in = p[0] >= bx[b][0] - 0.016f && p[0] <= bx[b][3] + 0.016f && p[1] >= bx[b][1] - 0.016f && p[1] <= bx[b][4] + 0.016f &&
                     p[2] >= bx[b][2] - 0.016f && p[2] <= bx[b][5] + 0.016f;
            if (in) { x.push_back(p[0]); x.push_back(p[1]); x.push_back(p[2]); }
        }
    }
    return x;
}

template <class F> static float time_us(F&& f, int reps = 20) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return 1e3f * ms / reps;
}

int main(int argc, char** argv) {
    const uint32_t B = argc > 1 ? (uint32_t)atoi(argv[1]) : (1u << 18);
    const Levels lv = make_levels();
    printf("levels:");
    for (uint32_t l = 0; l < L; ++l) printf(" %u%s", lv.res[l], lv.hashed[l] ? "h" : "");
    printf("  rows %u\n", lv.off[L]);
    std::vector<float> hx = make_samples(B);
    std::vector<uint2> htab(lv.off[L]);
    uint32_t s = 99;
    for (auto& r : htab) {
        const float f = rnd(s) - 0.5f;
        h2 c; c.x = (_Float16)(rnd(s) - 0.5f); c.y = (_Float16)(rnd(s) - 0.5f);
        uint32_t fb, cb;
        memcpy(&fb, &f, 4); memcpy(&cb, &c, 4);
        r = make_uint2(fb, cb);
    }
    float *dx, *o1; _Float16* o2; uint2* tab; uint32_t* sink;
    CK(hipMalloc(&dx, (size_t)B * 12)); CK(hipMalloc(&o1, (size_t)B * L * 4)); CK(hipMalloc(&o2, (size_t)B * L * 4));
    CK(hipMalloc(&tab, htab.size() * 8)); CK(hipMalloc(&sink, 64));
    CK(hipMemcpy(dx, hx.data(), (size_t)B * 12, hipMemcpyHostToDevice));
    CK(hipMemcpy(tab, htab.data(), htab.size() * 8, hipMemcpyHostToDevice));
    const uint32_t n_tiles = (B + 255) / 256;
    const double algo = (double)B * (12.0 + 16.0 * 8 * 8 + 16.0 * 8);
    std::vector<float> ref((size_t)B * L), got((size_t)B * L);
    std::vector<uint32_t> ref2((size_t)B * L), got2((size_t)B * L);
    auto check = [&](const char* name) {
        CK(hipMemcpy(got.data(), o1, got.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(got2.data(), o2, got2.size() * 4, hipMemcpyDeviceToHost));
        const bool same = memcmp(got.data(), ref.data(), got.size() * 4) == 0 && memcmp(got2.data(), ref2.data(), got2.size() * 4) == 0;
        if (!same) printf("   !! %s: outputs differ from the baseline\n", name);
    };
    auto report = [&](const char* name, float us) { printf("%-44s %8.1f us  %6.0f GB/s algorithmic = %4.1f %% of 8 TB/s\n", name, us, algo / us * 1e-3, algo / us * 1e-3 / 80.0); };

    float us = time_us([&] { fwd_kernel<0, 0><<<n_tiles * L, 256>>>(dx, tab, o1, o2, B, lv, n_tiles, 0, L); });
    CK(hipMemcpy(ref.data(), o1, ref.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ref2.data(), o2, ref2.size() * 4, hipMemcpyDeviceToHost));
    report("baseline (level-major grid)", us);
    CK(hipMemset(o1, 0, (size_t)B * L * 4));
    us = time_us([&] { fwd_kernel<0, 1><<<n_tiles * L, 256>>>(dx, tab, o1, o2, B, lv, n_tiles, 0, L); }); check("tile-major"); report("tile-major grid", us);
    for (uint32_t g : {4u, 2u, 1u}) {
        const uint32_t tpx = (n_tiles + g - 1) / g, blocks = 8u * (16u / (8u / g)) * tpx;
        us = time_us([&] { fwd_kernel<0, 2><<<blocks, 256>>>(dx, tab, o1, o2, B, lv, n_tiles, g, L); });
        char name[64]; snprintf(name, sizeof name, "XCD groups: %u XCDs per level", g);
        check(name); report(name, us);
    }
    for (uint32_t g : {4u, 2u}) {
        const uint32_t tpx = (n_tiles + g - 1) / g, blocks = 8u * (16u / (8u / g)) * tpx;
        us = time_us([&] { fwd_kernel<2, 2><<<blocks, 256>>>(dx, tab, o1, o2, B, lv, n_tiles, g, L); });
        char name[64]; snprintf(name, sizeof name, "XCD groups %u + dense unaligned", g);
        check(name); report(name, us);
    }
    us = time_us([&] { fwd_kernel<1, 0><<<n_tiles * L, 256>>>(dx, tab, o1, o2, B, lv, n_tiles, 0, L); }); check("nt"); report("nontemporal loads on hashed levels", us);
    us = time_us([&] { fwd_kernel<2, 0><<<n_tiles * L, 256>>>(dx, tab, o1, o2, B, lv, n_tiles, 0, L); }); check("unaligned"); report("dense: one unaligned 16-byte load per pair", us);
    us = time_us([&] { fwd_kernel<3, 0><<<n_tiles * L, 256>>>(dx, tab, o1, o2, B, lv, n_tiles, 0, L); }); check("deferred"); report("hashed: partner rows after the pair loads landed", us);
    us = time_us([&] { fwd_kernel<4, 0><<<n_tiles * L, 256>>>(dx, tab, o1, o2, B, lv, n_tiles, 0, L); }); check("deferred+unaligned"); report("  + dense unaligned pairs", us);
    us = time_us([&] { fwd2_kernel<0><<<n_tiles * 8, 256>>>(dx, tab, o1, o2, B, lv, n_tiles); }); check("two levels"); report("two levels per thread (l, l+8)", us);
    printf("per level (baseline kernel, one level per launch):\n");
    float sum = 0;
    for (uint32_t l = 0; l < L; ++l) {
        const float t = time_us([&] { fwd_kernel<0, 0><<<n_tiles, 256>>>(dx, tab, o1, o2, B, lv, n_tiles, l, l + 1); }, 10);
        sum += t;
        printf("   level %2u res %4u %s %7.1f us\n", l, lv.res[l], lv.hashed[l] ? "hash " : "dense", t);
    }
    printf("   sum %.1f us\n", sum);
    // raw gather probes: 2^18 * 16 * 4 = 16.8 M lane-loads (the pair loads of one forward), random slots
    printf("raw 16-byte / 8-byte random gathers, 16.8 M lane-loads:\n");
    uint4* big;
    CK(hipMalloc(&big, 64u << 20));
    CK(hipMemset(big, 1, 64u << 20));
    for (uint32_t bytes : {32u << 10, 1u << 20, 4u << 20, 48u << 20}) {
        uint32_t slots = 1;
        while (slots * 2 * 16 <= bytes) slots *= 2;
        const uint32_t blocks = 16384, per_lane = 4;     // 16384 * 256 * 4 = 16.8 M
        const float t16 = time_us([&] { gather_probe<16><<<blocks, 256>>>(big, slots - 1, per_lane, sink); }, 10);
        const float t8 = time_us([&] { gather_probe<8><<<blocks, 256>>>(big, slots - 1, per_lane, sink); }, 10);
        printf("   table %6u KiB: 16-byte %7.1f us = %6.1f G loads/s   8-byte %7.1f us = %6.1f G loads/s\n", slots * 16 / 1024, t16, 16.777216e6 / t16 * 1e-3,
               t8, 16.777216e6 / t8 * 1e-3);
    }
    return 0;
}
