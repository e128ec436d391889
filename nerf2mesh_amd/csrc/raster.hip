// Stage-1 rasterise / interpolate / antialias for gfx950 -- include/n2m_raster.h.
// Stands in for the nvdiffrast operators nerf2mesh calls (nerf/renderer.py:126-128,338-340,860-863,886-887,961-968);
// nvdiffrast itself is not vendored in the reference, so this implements the published semantics
// from scratch.
//
// Rasteriser design (MI355X): no fixed-function pipeline, so visibility is a 64-bit atomic z-buffer:
//   key = order-preserving(z/w) << 32 | triangle id,  atomicMin per covered pixel  (global_atomic_umin_x2).
// One lane per triangle walks the pixels of its bounding box (meshes at this stage have ~8 px per triangle); triangles
// whose box exceeds a budget, or that touch w <= 0, are queued and swept by whole workgroups in a second kernel so a few
// large triangles cannot serialise a wave.  A resolve pass turns the winning id of every pixel into (u, v, z/w, id+1).
// Coverage is decided in 1/256-pixel fixed point with a top-left tie rule (watertight: a pixel centre on a shared edge
// belongs to exactly one of the two triangles, independent of draw order); depth and barycentrics are evaluated from
// the un-snapped clip coordinates with homogeneous edge functions, so they are exactly the quantities the backward
// differentiates.
#include "n2m_common.hpp"
#include "../../include/n2m_raster.h"

namespace {

constexpr int kSub = 256;                       // sub-pixel resolution of the coverage test
constexpr uint32_t kLaneBox = 16;               // a lane rasterises boxes up to this many pixels itself
constexpr uint32_t kMidLanes = 16;              // lanes per triangle of the middle class
constexpr uint32_t kSmallBox = 1024;            // ... which ends here: larger boxes get a workgroup each

struct Tri {
    float x[3], y[3], z[3], w[3];
};

__device__ __forceinline__ bool load_tri(const float* __restrict__ pos, const int32_t* __restrict__ tri, uint32_t f, uint32_t V, Tri& t) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int32_t i = tri[3 * f + k];
        if (i < 0 || (uint32_t)i >= V) return false;
        const float4 p = *reinterpret_cast<const float4*>(pos + 4 * (size_t)i);
        t.x[k] = p.x; t.y[k] = p.y; t.z[k] = p.z; t.w[k] = p.w;
    }
    return true;
}

// Perspective-correct barycentrics (weights of vertices 0 and 1), z/w and interpolated w at NDC point (fx, fy),
// from homogeneous edge functions.  Returns false for a degenerate triangle.
__device__ __forceinline__ bool eval_point(const Tri& t, float fx, float fy, float& b0, float& b1, float& zw, float& wp) {
    const float p0x = t.x[0] - fx * t.w[0], p0y = t.y[0] - fy * t.w[0];
    const float p1x = t.x[1] - fx * t.w[1], p1y = t.y[1] - fy * t.w[1];
    const float p2x = t.x[2] - fx * t.w[2], p2y = t.y[2] - fy * t.w[2];
    const float a0 = p1x * p2y - p1y * p2x, a1 = p2x * p0y - p2y * p0x, a2 = p0x * p1y - p0y * p1x;
    const float S = a0 + a1 + a2;
    if (S == 0.0f) return false;
    const float iw = 1.0f / S;
    b0 = a0 * iw; b1 = a1 * iw;
    const float b2 = a2 * iw;
    const float z = t.z[0] * b0 + t.z[1] * b1 + t.z[2] * b2;
    wp = t.w[0] * b0 + t.w[1] * b1 + t.w[2] * b2;
    zw = z / wp;
    return true;
}

__device__ __forceinline__ uint32_t depth_key(float z) {   // monotonic float -> uint
    const uint32_t u = __float_as_uint(z);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ long long floor_div(long long a, long long b) {   // b > 0
    return a >= 0 ? a / b : -((-a + b - 1) / b);
}

struct Setup {
    Tri t;
    long long X[3], Y[3];     // fixed-point screen coordinates (1/256 px)
    long long sign;           // +1 / -1 orientation (0 = degenerate)
    int x0, x1, y0, y1;       // pixel bounding box (inclusive), clipped to the image
    bool fixed;               // coverage by fixed-point edge functions (all w > 0), else float homogeneous
    bool valid;
};

__device__ __forceinline__ void setup_tri(const float* pos, const int32_t* tri, uint32_t f, uint32_t V, uint32_t H, uint32_t W, Setup& s) {
    s.valid = load_tri(pos, tri, f, V, s.t);
    if (!s.valid) return;
    const Tri& t = s.t;
    s.fixed = t.w[0] > 1e-12f && t.w[1] > 1e-12f && t.w[2] > 1e-12f;
    if (s.fixed) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float px = (t.x[k] / t.w[k] * 0.5f + 0.5f) * (float)W, py = (t.y[k] / t.w[k] * 0.5f + 0.5f) * (float)H;
            if (!(fabsf(px) < 1048576.f && fabsf(py) < 1048576.f)) s.fixed = false;    // also catches NaN
            s.X[k] = (long long)lrintf(px * (float)kSub);
            s.Y[k] = (long long)lrintf(py * (float)kSub);
        }
    }
    if (s.fixed) {
        const long long area2 = (s.X[1] - s.X[0]) * (s.Y[2] - s.Y[0]) - (s.Y[1] - s.Y[0]) * (s.X[2] - s.X[0]);
        s.sign = area2 > 0 ? 1 : (area2 < 0 ? -1 : 0);
        if (s.sign == 0) { s.valid = false; return; }
        const long long mnx = min(s.X[0], min(s.X[1], s.X[2])), mxx = max(s.X[0], max(s.X[1], s.X[2]));
        const long long mny = min(s.Y[0], min(s.Y[1], s.Y[2])), mxy = max(s.Y[0], max(s.Y[1], s.Y[2]));
        // pixel centre ix*256+128 must lie in [mn, mx]:  ix >= ceil((mn-128)/256),  ix <= floor((mx-128)/256)
        const long long ax = -floor_div(128 - mnx, kSub), bx = floor_div(mxx - 128, kSub);
        const long long ay = -floor_div(128 - mny, kSub), by = floor_div(mxy - 128, kSub);
        s.x0 = (int)max(ax, 0ll); s.x1 = (int)min(bx, (long long)W - 1);
        s.y0 = (int)max(ay, 0ll); s.y1 = (int)min(by, (long long)H - 1);
    } else {
        s.sign = 1;
        s.x0 = 0; s.x1 = (int)W - 1; s.y0 = 0; s.y1 = (int)H - 1;
    }
}

// coverage + depth of pixel (ix, iy); returns true and zw when the triangle covers the pixel centre inside the depth range
__device__ __forceinline__ bool cover_pixel(const Setup& s, int ix, int iy, uint32_t H, uint32_t W, float& zw) {
    const float fx = ((float)ix + 0.5f) * (2.0f / (float)W) - 1.0f, fy = ((float)iy + 0.5f) * (2.0f / (float)H) - 1.0f;
    float b0, b1, wp;
    if (s.fixed) {
        const long long px = (long long)ix * kSub + 128, py = (long long)iy * kSub + 128;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int a = k, b = (k + 1) % 3;
            const long long dx = s.sign * (s.X[b] - s.X[a]), dy = s.sign * (s.Y[b] - s.Y[a]);
            const long long e = dx * (py - s.Y[a]) - dy * (px - s.X[a]);
            if (e < 0) return false;
            if (e == 0 && !(dy > 0 || (dy == 0 && dx > 0))) return false;      // top-left style tie rule
        }
        if (!eval_point(s.t, fx, fy, b0, b1, zw, wp)) return false;
    } else {
        if (!eval_point(s.t, fx, fy, b0, b1, zw, wp)) return false;
        const float b2 = 1.0f - b0 - b1;
        if (!(b0 >= 0.f && b1 >= 0.f && b2 >= 0.f && wp > 0.f)) return false;
    }
    return zw >= -1.0f && zw <= 1.0f;
}

__global__ void zbuf_clear_kernel(unsigned long long* __restrict__ zbuf, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) zbuf[i] = ~0ull;
}

// Coverage of a `fixed` triangle as three running edge values.  E_k(ix, iy) = e_k - (tie_k ? 0 : 1) with e_k the edge function of cover_pixel()
// at the pixel centre and tie_k its top-left rule, so "covered" is E_0, E_1, E_2 all >= 0: one OR and a sign test.  |X|, |Y| <= 2^28 by the
// `fixed` condition, so the edge deltas fit 32 bits, the 64-bit products (v_mad_i64_i32) are exact, and a step of one pixel is an exact 64-bit
// add: the same integers as cover_pixel() forms with full 64-bit multiplies at every pixel, at a tenth of the instructions.
struct EdgeOrigin {
    long long E[3];            // biased edge values at the box origin (x0, y0)
    int ndy[3], dx[3];         // per-sub-pixel steps: E_k += ndy_k per unit of px, += dx_k per unit of py
};

__device__ __forceinline__ void edge_origin(const Setup& s, EdgeOrigin& o) {
    const int px = s.x0 * kSub + 128, py = s.y0 * kSub + 128, sg = (int)s.sign;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int a = k, b = (k + 1) % 3;
        const int Xa = (int)s.X[a], Ya = (int)s.Y[a];
        const int dx = sg * ((int)s.X[b] - Xa), dy = sg * ((int)s.Y[b] - Ya);
        const long long e = (long long)dx * (long long)(py - Ya) - (long long)dy * (long long)(px - Xa);
        const bool tie = dy > 0 || (dy == 0 && dx > 0);                     // top-left style tie rule
        o.E[k] = e - (tie ? 0ll : 1ll);
        o.ndy[k] = -dy; o.dx[k] = dx;
    }
}

// depth of a covered pixel; false when the triangle is degenerate there or the depth leaves [-1, 1] (second half of cover_pixel())
__device__ __forceinline__ bool depth_at(const Setup& s, int ix, int iy, uint32_t H, uint32_t W, float& zw) {
    const float fx = ((float)ix + 0.5f) * (2.0f / (float)W) - 1.0f, fy = ((float)iy + 0.5f) * (2.0f / (float)H) - 1.0f;
    float b0, b1, wp;
    if (!eval_point(s.t, fx, fy, b0, b1, zw, wp)) return false;
    return zw >= -1.0f && zw <= 1.0f;
}

// queue[count++] = f for the threads that `want`, ONE atomic per workgroup of 256 (a returning atomic per wave -- 4 761 of them on one address
// per frame -- cost the small-triangle kernel 10 of its 30 us).  Every thread of the workgroup must call it; `slot` picks the LDS scratch.
__device__ __forceinline__ void block_append(uint32_t* __restrict__ queue, uint32_t* __restrict__ count, bool want, uint32_t f, uint32_t (&scratch)[5]) {
    const unsigned long long m = __ballot(want);
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    if (lane == 0) scratch[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = scratch[0] + scratch[1] + scratch[2] + scratch[3];
        scratch[4] = total ? atomicAdd(count, total) : 0u;
    }
    __syncthreads();
    if (want) {
        uint32_t at = scratch[4] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        for (uint32_t w = 0; w < wave; ++w) at += scratch[w];
        queue[at] = f;
    }
    __syncthreads();                      // the scratch is reused by the next call
}

// Three size classes.  A kernel of one lane per triangle lasts as long as its slowest wave, i.e. as the largest box anywhere in the frame: a
// decimated mesh seen at 1600 x 1600 has a median box of 8 pixels and a 99th percentile of 180, and the one-class kernel of rounds 1-3 spent
// 78 us on 20 us of work (the wave with the 14 x 14 boxes ran 200 iterations, the median wave 14).  So: a lane walks boxes up to kLaneBox
// pixels itself (flat loop: nested loops would cost max(width) x max(height) of the wave); boxes up to kSmallBox go to the `mid` queue (one atomic per workgroup)
// and are walked by 16 lanes each; anything larger, or touching w <= 0, goes to the `big` queue and gets a whole workgroup.
__global__ void __launch_bounds__(256)
raster_small_kernel(const float* __restrict__ pos, const int32_t* __restrict__ tri, uint32_t V, uint32_t F, uint32_t H, uint32_t W,
                    unsigned long long* __restrict__ zbuf, uint32_t* __restrict__ mid_queue, uint32_t* __restrict__ mid_count,
                    uint32_t* __restrict__ big_queue, uint32_t* __restrict__ big_count) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    Setup s;
    bool own = false, mid = false, big = false;
    uint32_t n = 0;
    if (f < F) {
        setup_tri(pos, tri, f, V, H, W, s);
        if (s.valid && s.x1 >= s.x0 && s.y1 >= s.y0) {
            const uint64_t box = (uint64_t)(s.x1 - s.x0 + 1) * (uint64_t)(s.y1 - s.y0 + 1);
            big = !s.fixed || box > kSmallBox;
            mid = !big && box > kLaneBox;
            own = !big && !mid;
            n = (uint32_t)box;
        }
    }
    __shared__ uint32_t scratch[5];
    block_append(mid_queue, mid_count, mid, f, scratch);
    if (__syncthreads_or(big)) block_append(big_queue, big_count, big, f, scratch);
    if (!own) return;
    EdgeOrigin o;
    edge_origin(s, o);
    long long row[3], e[3], sx[3], sy[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { row[k] = e[k] = o.E[k]; sx[k] = (long long)o.ndy[k] * kSub; sy[k] = (long long)o.dx[k] * kSub; }
    int ix = s.x0, iy = s.y0;
    for (uint32_t k = 0; k < n; ++k) {
        float zw;
        if ((e[0] | e[1] | e[2]) >= 0 && depth_at(s, ix, iy, H, W, zw))
            atomicMin(&zbuf[(size_t)iy * W + ix], ((unsigned long long)depth_key(zw) << 32) | f);
        if (++ix > s.x1) {
            ix = s.x0; ++iy;
#pragma unroll
            for (int q = 0; q < 3; ++q) { row[q] += sy[q]; e[q] = row[q]; }
        } else {
#pragma unroll
            for (int q = 0; q < 3; ++q) e[q] += sx[q];
        }
    }
}

// kMidLanes lanes per queued triangle (each repeats the set-up: 300 instructions against a box of 17 .. 1024 pixels), persistent groups
__global__ void __launch_bounds__(256)
raster_mid_kernel(const float* __restrict__ pos, const int32_t* __restrict__ tri, uint32_t V, uint32_t H, uint32_t W,
                  unsigned long long* __restrict__ zbuf, const uint32_t* __restrict__ mid_queue, const uint32_t* __restrict__ mid_count) {
    const uint32_t n_q = *mid_count, groups = gridDim.x * (256u / kMidLanes), sub = threadIdx.x % kMidLanes;
    for (uint32_t q = blockIdx.x * (256u / kMidLanes) + threadIdx.x / kMidLanes; q < n_q; q += groups) {
        const uint32_t f = mid_queue[q];
        Setup s;
        setup_tri(pos, tri, f, V, H, W, s);
        const uint32_t bw = (uint32_t)(s.x1 - s.x0 + 1), n = bw * (uint32_t)(s.y1 - s.y0 + 1);
        EdgeOrigin o;
        edge_origin(s, o);
        const float inv_bw = 1.0f / (float)bw;
        for (uint32_t k = sub; k < n; k += kMidLanes) {
            // k / bw for k < 1024, bw <= 1024: (k + 0.5) / bw is at least 0.5 / bw from an integer, 4 ulp of the product -- the truncation is exact
            const uint32_t qy = (uint32_t)(((float)k + 0.5f) * inv_bw), qx = k - qy * bw;
            const int ox = (int)(qx * (uint32_t)kSub), oy = (int)(qy * (uint32_t)kSub);
            long long e = 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) e |= o.E[c] + (long long)o.ndy[c] * (long long)ox + (long long)o.dx[c] * (long long)oy;
            if (e < 0) continue;
            const int ix = s.x0 + (int)qx, iy = s.y0 + (int)qy;
            float zw;
            if (depth_at(s, ix, iy, H, W, zw))
                atomicMin(&zbuf[(size_t)iy * W + ix], ((unsigned long long)depth_key(zw) << 32) | f);
        }
    }
}

// persistent workgroups: one queued triangle at a time, 256 lanes stride its bounding box
__global__ void __launch_bounds__(256)
raster_big_kernel(const float* __restrict__ pos, const int32_t* __restrict__ tri, uint32_t V, uint32_t H, uint32_t W,
                  unsigned long long* __restrict__ zbuf, const uint32_t* __restrict__ big_queue, const uint32_t* __restrict__ big_count) {
    const uint32_t n = *big_count;
    for (uint32_t q = blockIdx.x; q < n; q += gridDim.x) {
        const uint32_t f = big_queue[q];
        Setup s;
        setup_tri(pos, tri, f, V, H, W, s);
        if (!s.valid || s.x1 < s.x0 || s.y1 < s.y0) continue;
        const uint32_t bw = (uint32_t)(s.x1 - s.x0 + 1), bh = (uint32_t)(s.y1 - s.y0 + 1);
        const uint64_t total = (uint64_t)bw * bh;
        for (uint64_t i = threadIdx.x; i < total; i += 256) {
            const int iy = s.y0 + (int)(i / bw), ix = s.x0 + (int)(i % bw);
            float zw;
            if (cover_pixel(s, ix, iy, H, W, zw))
                atomicMin(&zbuf[(size_t)iy * W + ix], ((unsigned long long)depth_key(zw) << 32) | f);
        }
    }
}

__global__ void raster_resolve_kernel(const float* __restrict__ pos, const int32_t* __restrict__ tri, uint32_t V, uint32_t H, uint32_t W,
                                      const unsigned long long* __restrict__ zbuf, float* __restrict__ rast) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    const unsigned long long key = zbuf[i];
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    if (key != ~0ull) {
        const uint32_t f = (uint32_t)(key & 0xFFFFFFFFull);
        Tri t;
        if (load_tri(pos, tri, f, V, t)) {
            const int ix = (int)(i % W), iy = (int)(i / W);
            const float fx = ((float)ix + 0.5f) * (2.0f / (float)W) - 1.0f, fy = ((float)iy + 0.5f) * (2.0f / (float)H) - 1.0f;
            float b0, b1, zw, wp;
            if (eval_point(t, fx, fy, b0, b1, zw, wp))
                out = make_float4(b0, b1, fminf(fmaxf(zw, -1.f), 1.f), (float)(f + 1));
        }
    }
    reinterpret_cast<float4*>(rast)[i] = out;
}

// d(u,v)/d(pos): differentiates eval_point's b0, b1 w.r.t. x, y, w of the three vertices
__global__ void raster_backward_kernel(const float* __restrict__ pos, const int32_t* __restrict__ tri, const float* __restrict__ rast,
                                       const float* __restrict__ d_rast, uint32_t V, uint32_t H, uint32_t W, float* __restrict__ grad_pos) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    const float4 r = reinterpret_cast<const float4*>(rast)[i];
    const int f = (int)r.w - 1;
    if (f < 0) return;
    const float g0 = d_rast[4 * (size_t)i], g1 = d_rast[4 * (size_t)i + 1];
    if (g0 == 0.f && g1 == 0.f) return;
    Tri t;
    if (!load_tri(pos, tri, (uint32_t)f, V, t)) return;
    const int ix = (int)(i % W), iy = (int)(i / W);
    const float fx = ((float)ix + 0.5f) * (2.0f / (float)W) - 1.0f, fy = ((float)iy + 0.5f) * (2.0f / (float)H) - 1.0f;
    float px[3], py[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { px[k] = t.x[k] - fx * t.w[k]; py[k] = t.y[k] - fy * t.w[k]; }
    const float a0 = px[1] * py[2] - py[1] * px[2], a1 = px[2] * py[0] - py[2] * px[0], a2 = px[0] * py[1] - py[0] * px[1];
    const float S = a0 + a1 + a2;
    if (S == 0.f) return;
    const float iS = 1.0f / S, b0 = a0 * iS, b1 = a1 * iS;
    // dL/da_k
    const float da[3] = {(g0 * (1.f - b0) - g1 * b1) * iS, (g1 * (1.f - b1) - g0 * b0) * iS, (-g0 * b0 - g1 * b1) * iS};
    // a_k = px[k+1]*py[k+2] - py[k+1]*px[k+2]  =>  dL/dpx[j], dL/dpy[j]
    float dpx[3] = {0.f, 0.f, 0.f}, dpy[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int j1 = (k + 1) % 3, j2 = (k + 2) % 3;
        dpx[j1] += da[k] * py[j2];  dpy[j2] += da[k] * px[j1];
        dpy[j1] -= da[k] * px[j2];  dpx[j2] -= da[k] * py[j1];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float* gp = grad_pos + 4 * (size_t)tri[3 * f + k];
        unsafeAtomicAdd(gp + 0, dpx[k]);
        unsafeAtomicAdd(gp + 1, dpy[k]);
        unsafeAtomicAdd(gp + 3, -fx * dpx[k] - fy * dpy[k]);
    }
}

// ------------------------------------------------------------------------------------------------ interpolate
__global__ void interpolate_forward_kernel(const float* __restrict__ attr, const float* __restrict__ rast, const int32_t* __restrict__ tri,
                                           uint32_t V, uint32_t F, uint32_t A, uint32_t HW, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    const float4 r = reinterpret_cast<const float4*>(rast)[i];
    const int f = (int)r.w - 1;
    float* o = out + (size_t)i * A;
    if (f < 0 || (uint32_t)f >= F) {
        for (uint32_t a = 0; a < A; ++a) o[a] = 0.f;
        return;
    }
    const int i0 = tri[3 * f], i1 = tri[3 * f + 1], i2 = tri[3 * f + 2];
    const float b0 = r.x, b1 = r.y, b2 = 1.0f - r.x - r.y;
    for (uint32_t a = 0; a < A; ++a)
        o[a] = b0 * attr[(size_t)i0 * A + a] + b1 * attr[(size_t)i1 * A + a] + b2 * attr[(size_t)i2 * A + a];
}

__global__ void interpolate_backward_kernel(const float* __restrict__ attr, const float* __restrict__ rast, const int32_t* __restrict__ tri,
                                            const float* __restrict__ d_out, uint32_t ds, uint32_t V, uint32_t F, uint32_t A, uint32_t HW,
                                            float* __restrict__ grad_attr, float* __restrict__ grad_rast) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    const float4 r = reinterpret_cast<const float4*>(rast)[i];
    const int f = (int)r.w - 1;
    float gu = 0.f, gv = 0.f;
    if (f >= 0 && (uint32_t)f < F) {
        const int i0 = tri[3 * f], i1 = tri[3 * f + 1], i2 = tri[3 * f + 2];
        const float b0 = r.x, b1 = r.y, b2 = 1.0f - r.x - r.y;
        for (uint32_t a = 0; a < A; ++a) {
            const float g = d_out[(size_t)i * ds + a];
            if (g != 0.f && grad_attr) {
                unsafeAtomicAdd(grad_attr + (size_t)i0 * A + a, g * b0);
                unsafeAtomicAdd(grad_attr + (size_t)i1 * A + a, g * b1);
                unsafeAtomicAdd(grad_attr + (size_t)i2 * A + a, g * b2);
            }
            const float a2 = attr[(size_t)i2 * A + a];
            gu += g * (attr[(size_t)i0 * A + a] - a2);
            gv += g * (attr[(size_t)i1 * A + a] - a2);
        }
    }
    if (grad_rast) reinterpret_cast<float4*>(grad_rast)[i] = make_float4(gu, gv, 0.f, 0.f);
}

// -------------------------------------------------------------------------------------------------- antialias
struct Edge { unsigned long long key; int32_t op0, op1; };   // 16 bytes = 4 x i32 (va, vb, op0, op1)

__device__ __forceinline__ uint32_t edge_hash(uint32_t a, uint32_t b) {
    uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA6Bu;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    return h;
}

__global__ void topology_clear_kernel(Edge* __restrict__ table, uint32_t capacity) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < capacity) { table[i].key = ~0ull; table[i].op0 = -1; table[i].op1 = -1; }
}

__global__ void topology_insert_kernel(const int32_t* __restrict__ tri, uint32_t F, Edge* table, uint32_t capacity) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3 * F) return;
    const uint32_t f = t / 3, k = t - 3 * f;
    const int32_t a = tri[3 * f + k], b = tri[3 * f + (k + 1) % 3], c = tri[3 * f + (k + 2) % 3];
    if (a < 0 || b < 0 || c < 0 || a == b) return;
    const uint32_t lo = (uint32_t)min(a, b), hi = (uint32_t)max(a, b);
    const unsigned long long key = ((unsigned long long)hi << 32) | lo;
    uint32_t slot = edge_hash(lo, hi) & (capacity - 1);
    for (uint32_t probe = 0; probe < capacity; ++probe, slot = (slot + 1) & (capacity - 1)) {
        const unsigned long long old = atomicCAS(&table[slot].key, ~0ull, key);
        if (old == ~0ull || old == key) {
            if (atomicCAS(&table[slot].op0, -1, c) != -1) atomicCAS(&table[slot].op1, -1, c);
            return;
        }
    }
}

__device__ __forceinline__ int topology_other(const Edge* __restrict__ table, uint32_t capacity, int32_t a, int32_t b, int32_t c) {
    const uint32_t lo = (uint32_t)min(a, b), hi = (uint32_t)max(a, b);
    const unsigned long long key = ((unsigned long long)hi << 32) | lo;
    uint32_t slot = edge_hash(lo, hi) & (capacity - 1);
    for (uint32_t probe = 0; probe < capacity; ++probe, slot = (slot + 1) & (capacity - 1)) {
        const unsigned long long k = table[slot].key;
        if (k == key) { const int32_t o0 = table[slot].op0, o1 = table[slot].op1; return o0 != c ? o0 : o1; }
        if (k == ~0ull) return -1;
    }
    return -1;
}

struct Crossing {
    int va, vb;                // silhouette edge that crosses the pixel-pair segment
    float ax, ay, bx, by;      // its end points in pixel coordinates
    float eP, eO;              // edge function at the covered pixel P and the other pixel O
    float d;                   // crossing fraction from P towards O
};

__device__ __forceinline__ float cross2(float ux, float uy, float vx, float vy) { return ux * vy - uy * vx; }

// Examines triangle f for a silhouette edge separating pixel centres P and O.
__device__ bool find_crossing(const float* __restrict__ pos, const int32_t* __restrict__ tri, const Edge* __restrict__ table,
                              uint32_t capacity, uint32_t V, int f, float Px, float Py, float Ox, float Oy, float W, float H, Crossing& out) {
    int id[3];
    float X[3], Y[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        id[k] = tri[3 * f + k];
        if (id[k] < 0 || (uint32_t)id[k] >= V) return false;
        const float4 p = *reinterpret_cast<const float4*>(pos + 4 * (size_t)id[k]);
        if (!(p.w > 1e-12f)) return false;
        X[k] = (p.x / p.w * 0.5f + 0.5f) * W; Y[k] = (p.y / p.w * 0.5f + 0.5f) * H;
    }
    bool found = false;
    out.d = 2.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int a = k, b = (k + 1) % 3, c = (k + 2) % 3;
        const float ex = X[b] - X[a], ey = Y[b] - Y[a];
        const float sc = cross2(ex, ey, X[c] - X[a], Y[c] - Y[a]);
        if (sc == 0.f) continue;
        // the cheap conditions first (this triangle's vertices only): the edge must separate P (inside) from O (outside), cross the pixel-pair
        // segment within its own extent, and beat the crossing found so far.  One edge in three gets this far; only then the topology lookup
        // (a hash probe and the opposite vertex) decides whether it is a silhouette -- same conditions as before, three times fewer probes.
        const float eP = cross2(ex, ey, Px - X[a], Py - Y[a]), eO = cross2(ex, ey, Ox - X[a], Oy - Y[a]);
        if (!(eP * sc >= 0.f && eO * sc < 0.f)) continue;                        // P inside, O outside
        const float d = eP / (eP - eO);
        const float qx = Px + d * (Ox - Px), qy = Py + d * (Oy - Py);
        const float tt = ((qx - X[a]) * ex + (qy - Y[a]) * ey) / (ex * ex + ey * ey);
        if (!(tt >= 0.f && tt <= 1.f)) continue;
        if (!(d < out.d)) continue;
        const int other = topology_other(table, capacity, id[a], id[b], id[c]);
        bool silhouette = true;
        if (other >= 0 && (uint32_t)other < V) {
            const float4 q = *reinterpret_cast<const float4*>(pos + 4 * (size_t)other);
            if (q.w > 1e-12f) {
                const float qx2 = (q.x / q.w * 0.5f + 0.5f) * W, qy2 = (q.y / q.w * 0.5f + 0.5f) * H;
                silhouette = cross2(ex, ey, qx2 - X[a], qy2 - Y[a]) * sc > 0.f;    // both neighbours on the same side
            }
        }
        if (!silhouette) continue;
        found = true;
        out.d = d; out.va = id[a]; out.vb = id[b];
        out.ax = X[a]; out.ay = Y[a]; out.bx = X[b]; out.by = Y[b]; out.eP = eP; out.eO = eO;
    }
    return found;
}

// pair selection shared by forward and backward: returns false when nothing to do
__device__ __forceinline__ bool select_pair(const float* __restrict__ rast, uint32_t W, uint32_t H, uint32_t p, int dir, uint32_t& P, uint32_t& O, int& f) {
    const uint32_t ix = p % W, iy = p / W;
    if (dir == 0 ? ix + 1 >= W : iy + 1 >= H) return false;
    const uint32_t q = dir == 0 ? p + 1 : p + W;
    const float4 ra = reinterpret_cast<const float4*>(rast)[p], rb = reinterpret_cast<const float4*>(rast)[q];
    const int ta = (int)ra.w - 1, tb = (int)rb.w - 1;
    if (ta == tb) return false;
    const bool use_a = tb < 0 || (ta >= 0 && ra.z <= rb.z);      // the nearer surface (or the only one)
    f = use_a ? ta : tb;
    P = use_a ? p : q;
    O = use_a ? q : p;
    return true;
}

__global__ void antialias_forward_kernel(const float* __restrict__ color, const float* __restrict__ rast, const float* __restrict__ pos,
                                         const int32_t* __restrict__ tri, const Edge* __restrict__ table, uint32_t capacity, uint32_t V,
                                         uint32_t C, uint32_t H, uint32_t W, float* __restrict__ out) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
#pragma unroll
    for (int dir = 0; dir < 2; ++dir) {
        uint32_t P, O;
        int f;
        if (!select_pair(rast, W, H, p, dir, P, O, f)) continue;
        Crossing cr;
        const float Px = (float)(P % W) + 0.5f, Py = (float)(P / W) + 0.5f, Ox = (float)(O % W) + 0.5f, Oy = (float)(O / W) + 0.5f;
        if (!find_crossing(pos, tri, table, capacity, V, f, Px, Py, Ox, Oy, (float)W, (float)H, cr)) continue;
        // the pixel on the far side of the midpoint moves toward its neighbour by |0.5 - d|
        const uint32_t dst = cr.d < 0.5f ? P : O, src = cr.d < 0.5f ? O : P;
        const float wgt = fabsf(0.5f - cr.d);
        for (uint32_t c = 0; c < C; ++c)
            unsafeAtomicAdd(out + (size_t)dst * C + c, wgt * (color[(size_t)src * C + c] - color[(size_t)dst * C + c]));
    }
}

__global__ void antialias_backward_kernel(const float* __restrict__ color, const float* __restrict__ rast, const float* __restrict__ pos,
                                          const int32_t* __restrict__ tri, const Edge* __restrict__ table, uint32_t capacity,
                                          const float* __restrict__ d_out, uint32_t V, uint32_t C, uint32_t H, uint32_t W, float boost,
                                          float* __restrict__ grad_color, float* __restrict__ grad_pos) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
#pragma unroll
    for (int dir = 0; dir < 2; ++dir) {
        uint32_t P, O;
        int f;
        if (!select_pair(rast, W, H, p, dir, P, O, f)) continue;
        Crossing cr;
        const float Px = (float)(P % W) + 0.5f, Py = (float)(P / W) + 0.5f, Ox = (float)(O % W) + 0.5f, Oy = (float)(O / W) + 0.5f;
        if (!find_crossing(pos, tri, table, capacity, V, f, Px, Py, Ox, Oy, (float)W, (float)H, cr)) continue;
        const bool near_side = cr.d < 0.5f;
        const uint32_t dst = near_side ? P : O, src = near_side ? O : P;
        const float wgt = fabsf(0.5f - cr.d);
        float dot = 0.f;      // sum_c g_dst[c] * (color_src - color_dst)
        for (uint32_t c = 0; c < C; ++c) {
            const float g = d_out[(size_t)dst * C + c];
            unsafeAtomicAdd(grad_color + (size_t)src * C + c, wgt * g);
            unsafeAtomicAdd(grad_color + (size_t)dst * C + c, -wgt * g);
            dot += g * (color[(size_t)src * C + c] - color[(size_t)dst * C + c]);
        }
        if (!grad_pos || dot == 0.f) continue;
        // out_dst += |0.5-d| * (src - dst):  d|0.5-d|/dd = -1 on the near side, +1 on the far side
        const float dL_dd = (near_side ? -1.f : 1.f) * dot;
        const float den = cr.eP - cr.eO, inv2 = 1.0f / (den * den);
        const float dL_deP = dL_dd * (-cr.eO * inv2), dL_deO = dL_dd * (cr.eP * inv2);
        // E(q) = (bx-ax)(qy-ay) - (by-ay)(qx-ax)
        auto dE = [&](float qx, float qy, float g, float& gax, float& gay, float& gbx, float& gby) {
            gax += g * (cr.by - qy); gay += g * (qx - cr.bx); gbx += g * (qy - cr.ay); gby += g * (-(qx - cr.ax));
        };
        float gax = 0.f, gay = 0.f, gbx = 0.f, gby = 0.f;
        dE(Px, Py, dL_deP, gax, gay, gbx, gby);
        dE(Ox, Oy, dL_deO, gax, gay, gbx, gby);
        // pixel coordinates -> clip: X = (x/w*0.5+0.5)*W
        const int vid[2] = {cr.va, cr.vb};
        const float gX[2] = {gax, gbx}, gY[2] = {gay, gby};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(pos + 4 * (size_t)vid[k]);
            const float iw = 1.0f / v.w;
            const float gx = gX[k] * 0.5f * (float)W * iw, gy = gY[k] * 0.5f * (float)H * iw;
            float* gp = grad_pos + 4 * (size_t)vid[k];
            unsafeAtomicAdd(gp + 0, boost * gx);
            unsafeAtomicAdd(gp + 1, boost * gy);
            unsafeAtomicAdd(gp + 3, boost * (-(gx * v.x + gy * v.y) * iw));
        }
    }
}

__global__ void copy_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

}  // namespace

extern "C" int n2m_rasterize_forward(const float* pos, const int32_t* tri, uint32_t V, uint32_t F, uint32_t H, uint32_t W,
                                     unsigned long long* zbuf, float* rast, void* stream) {
    N2M_NOTNULL(zbuf); N2M_NOTNULL(rast);
    N2M_REQUIRE(H >= 1 && W >= 1 && (uint64_t)H * W < (1ull << 31), N2M_EINVAL, "rasterize: bad resolution %ux%u", H, W);
    if (F > 0) { N2M_NOTNULL(pos); N2M_NOTNULL(tri); }
    hipStream_t s = (hipStream_t)stream;
    const uint32_t HW = H * W;
    N2M_PROF(N2M_K_RASTER, s, 16.0 * V + 12.0 * F + 16.0 * HW);
    zbuf_clear_kernel<<<n2m_ceil_div(HW, 256), 256, 0, s>>>(zbuf, HW);
    if (F > 0) {
        uint32_t* queue = nullptr;          // [mid queue F | big queue F | mid count, big count]
        N2M_HIP(hipMallocAsync((void**)&queue, sizeof(uint32_t) * (2 * (size_t)F + 2), s));
        uint32_t* big_queue = queue + F;
        uint32_t* count = queue + 2 * (size_t)F;
        N2M_HIP(hipMemsetAsync(count, 0, 2 * sizeof(uint32_t), s));
        raster_small_kernel<<<n2m_ceil_div(F, 256), 256, 0, s>>>(pos, tri, V, F, H, W, zbuf, queue, count, big_queue, count + 1);
        raster_mid_kernel<<<(uint32_t)std::min<uint64_t>(n2m_ceil_div(F, 256u / kMidLanes), 8192), 256, 0, s>>>(pos, tri, V, H, W, zbuf, queue, count);
        raster_big_kernel<<<1024, 256, 0, s>>>(pos, tri, V, H, W, zbuf, big_queue, count + 1);
        N2M_HIP(hipFreeAsync(queue, s));
    }
    raster_resolve_kernel<<<n2m_ceil_div(HW, 256), 256, 0, s>>>(pos, tri, V, H, W, zbuf, rast);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_rasterize_backward(const float* pos, const int32_t* tri, const float* rast, const float* d_rast, uint32_t V,
                                      uint32_t F, uint32_t H, uint32_t W, float* grad_pos, void* stream) {
    (void)F;
    N2M_NOTNULL(pos); N2M_NOTNULL(tri); N2M_NOTNULL(rast); N2M_NOTNULL(d_rast); N2M_NOTNULL(grad_pos);
    N2M_PROF(N2M_K_RASTER_BWD, (hipStream_t)stream, 32.0 * H * W + 16.0 * V);
    raster_backward_kernel<<<n2m_ceil_div((uint64_t)H * W, 256), 256, 0, (hipStream_t)stream>>>(pos, tri, rast, d_rast, V, H, W, grad_pos);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_interpolate_forward(const float* attr, const float* rast, const int32_t* tri, uint32_t V, uint32_t F, uint32_t A,
                                       uint32_t H, uint32_t W, float* out, void* stream) {
    N2M_NOTNULL(attr); N2M_NOTNULL(rast); N2M_NOTNULL(tri); N2M_NOTNULL(out);
    N2M_REQUIRE(A >= 1, N2M_EINVAL, "interpolate: attribute count must be >= 1");
    // 16 hw (rast) + covered * (12 + 12 A) + 4 A hw (out); the host does not know the covered count: every pixel is counted as covered (upper bound)
    N2M_PROF(N2M_K_INTERP_FWD, (hipStream_t)stream, (double)H * W * (16.0 + 12.0 + 12.0 * A + 4.0 * A));
    interpolate_forward_kernel<<<n2m_ceil_div((uint64_t)H * W, 256), 256, 0, (hipStream_t)stream>>>(attr, rast, tri, V, F, A, H * W, out);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_interpolate_backward(const float* attr, const float* rast, const int32_t* tri, const float* d_out, uint32_t V,
                                        uint32_t F, uint32_t A, uint32_t H, uint32_t W, float* grad_attr, float* grad_rast,
                                        void* stream) {
    N2M_NOTNULL(attr); N2M_NOTNULL(rast); N2M_NOTNULL(tri); N2M_NOTNULL(d_out);
    N2M_PROF(N2M_K_INTERP_BWD, (hipStream_t)stream, (double)H * W * (16.0 + 12.0 + 12.0 * A + 4.0 * A + (grad_attr ? 12.0 * A : 0.0) + (grad_rast ? 16.0 : 0.0)));
    interpolate_backward_kernel<<<n2m_ceil_div((uint64_t)H * W, 256), 256, 0, (hipStream_t)stream>>>(attr, rast, tri, d_out, A, V, F, A, H * W,
                                                                                                  grad_attr, grad_rast);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_interpolate_backward_strided(const float* attr, const float* rast, const int32_t* tri, const float* d_out, uint32_t d_out_stride,
                                                uint32_t V, uint32_t F, uint32_t A, uint32_t H, uint32_t W, float* grad_attr, float* grad_rast,
                                                void* stream) {
    N2M_NOTNULL(attr); N2M_NOTNULL(rast); N2M_NOTNULL(tri); N2M_NOTNULL(d_out);
    N2M_REQUIRE(d_out_stride >= A, N2M_EINVAL, "interpolate_backward_strided: pixel stride below the attribute count");
    N2M_PROF(N2M_K_INTERP_BWD, (hipStream_t)stream, (double)H * W * (16.0 + 12.0 + 12.0 * A + 4.0 * A + (grad_attr ? 12.0 * A : 0.0) + (grad_rast ? 16.0 : 0.0)));
    interpolate_backward_kernel<<<n2m_ceil_div((uint64_t)H * W, 256), 256, 0, (hipStream_t)stream>>>(attr, rast, tri, d_out, d_out_stride, V, F, A,
                                                                                                  H * W, grad_attr, grad_rast);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_antialias_build_topology(const int32_t* tri, uint32_t F, int32_t* table, uint32_t capacity, void* stream) {
    N2M_NOTNULL(table);
    N2M_REQUIRE(capacity >= 4 && (capacity & (capacity - 1)) == 0 && (uint64_t)capacity >= 4ull * F, N2M_EINVAL,
                "antialias topology: capacity must be a power of two >= 4*F (F=%u, capacity=%u)", F, capacity);
    hipStream_t s = (hipStream_t)stream;
    topology_clear_kernel<<<n2m_ceil_div(capacity, 256), 256, 0, s>>>(reinterpret_cast<Edge*>(table), capacity);
    if (F > 0) {
        N2M_NOTNULL(tri);
        topology_insert_kernel<<<n2m_ceil_div(3ull * F, 256), 256, 0, s>>>(tri, F, reinterpret_cast<Edge*>(table), capacity);
    }
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_antialias_forward(const float* color, const float* rast, const float* pos, const int32_t* tri,
                                     const int32_t* table, uint32_t capacity, uint32_t V, uint32_t F, uint32_t C, uint32_t H,
                                     uint32_t W, float* out, void* stream) {
    (void)F;
    N2M_NOTNULL(color); N2M_NOTNULL(rast); N2M_NOTNULL(pos); N2M_NOTNULL(tri); N2M_NOTNULL(table); N2M_NOTNULL(out);
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)H * W * C;
    N2M_PROF(N2M_K_AA_FWD, s, (double)H * W * (8.0 * C + 16.0));          // SURVEY 8d (silhouette-pixel edge fetches not counted)
    copy_kernel<<<n2m_ceil_div(n, 256), 256, 0, s>>>(color, out, n);
    antialias_forward_kernel<<<n2m_ceil_div((uint64_t)H * W, 256), 256, 0, s>>>(color, rast, pos, tri, reinterpret_cast<const Edge*>(table),
                                                                              capacity, V, C, H, W, out);
    N2M_CHECK_LAUNCH();
    return 0;
}

static int antialias_backward_impl(bool seeded, const float* color, const float* rast, const float* pos, const int32_t* tri,
                                      const int32_t* table, uint32_t capacity, const float* d_out, uint32_t V, uint32_t F, uint32_t C,
                                      uint32_t H, uint32_t W, float pos_gradient_boost, float* grad_color, float* grad_pos,
                                      void* stream) {
    (void)F;
    N2M_NOTNULL(color); N2M_NOTNULL(rast); N2M_NOTNULL(pos); N2M_NOTNULL(tri); N2M_NOTNULL(table); N2M_NOTNULL(d_out);
    N2M_NOTNULL(grad_color);
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)H * W * C;
    N2M_PROF(N2M_K_AA_BWD, s, (double)H * W * (12.0 * C + 16.0));
    if (!seeded) copy_kernel<<<n2m_ceil_div(n, 256), 256, 0, s>>>(d_out, grad_color, n);     // identity part of the operator
    antialias_backward_kernel<<<n2m_ceil_div((uint64_t)H * W, 256), 256, 0, s>>>(color, rast, pos, tri, reinterpret_cast<const Edge*>(table),
                                                                               capacity, d_out, V, C, H, W, pos_gradient_boost, grad_color,
                                                                               grad_pos);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_antialias_backward(const float* color, const float* rast, const float* pos, const int32_t* tri,
                                      const int32_t* table, uint32_t capacity, const float* d_out, uint32_t V, uint32_t F, uint32_t C,
                                      uint32_t H, uint32_t W, float pos_gradient_boost, float* grad_color, float* grad_pos,
                                      void* stream) {
    return antialias_backward_impl(false, color, rast, pos, tri, table, capacity, d_out, V, F, C, H, W, pos_gradient_boost, grad_color, grad_pos, stream);
}

// grad_color already holds a copy of d_out (the producer of d_out wrote it twice): the identity part of the operator costs no pass
extern "C" int n2m_antialias_backward_seeded(const float* color, const float* rast, const float* pos, const int32_t* tri,
                                             const int32_t* table, uint32_t capacity, const float* d_out, uint32_t V, uint32_t F, uint32_t C,
                                             uint32_t H, uint32_t W, float pos_gradient_boost, float* grad_color, float* grad_pos,
                                             void* stream) {
    N2M_REQUIRE(grad_color != d_out, N2M_EINVAL, "antialias_backward_seeded: grad_color must be a second buffer (the kernel reads d_out while it adds)");
    return antialias_backward_impl(true, color, rast, pos, tri, table, capacity, d_out, V, F, C, H, W, pos_gradient_boost, grad_color, grad_pos, stream);
}
