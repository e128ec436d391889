// Training-step helpers: work the reference composes from many small torch launches, one launch each here.
//
// (1) Photometric loss head of the stage-0 training step, fused: background blend of the prediction
// (nerf/renderer.py:747 `image + (1 - weights_sum) * bg_color`), alpha compositing of the ground truth
// (nerf/utils.py:663-664), per-ray MSE over channels, mask MSE and the mean over rays (nerf/utils.py:679-683,
// `loss.mean()` :797) -- ~35 elementwise / reduction launches over [N,3] tensors in the reference's autograd graph,
// two launches here.
// (2) Ray generation for a batch of (view, pixel) pairs with the ground-truth gather (nerf/utils.py:242-290 get_rays +
// nerf/provider.py:330), ~20 launches in torch.
// Entry points are declared in include/n2m_hip.h.
#include <string.h>

#include "n2m_common.hpp"
#include "../../include/n2m_peer.h"

namespace {

struct RayTerm {
    float e[3];      // prediction - ground truth per channel
    float m;         // weights_sum - alpha
    float bg[3];
};

__device__ __forceinline__ RayTerm ray_term(const float* __restrict__ image, const float* __restrict__ wsum,
                                            const float* __restrict__ gt, const float* __restrict__ bg, float bg_scalar, uint32_t r) {
    RayTerm t;
    const float4 g = *reinterpret_cast<const float4*>(gt + (size_t)r * 4);
    const float a = g.w, ws = wsum[r];
    const float gc[3] = {g.x, g.y, g.z};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        t.bg[c] = bg ? bg[(size_t)r * 3 + c] : bg_scalar;
        const float target = gc[c] * a + t.bg[c] * (1.0f - a);
        const float pred = image[(size_t)r * 3 + c] + (1.0f - ws) * t.bg[c];
        t.e[c] = pred - target;
    }
    t.m = ws - a;
    return t;
}

__global__ void __launch_bounds__(256)
photo_loss_forward_kernel(const float* __restrict__ image, const float* __restrict__ wsum, const float* __restrict__ gt,
                          const float* __restrict__ bg, float bg_scalar, float lambda_rgb, float lambda_mask, uint32_t N,
                          float* __restrict__ partial, uint32_t* __restrict__ ticket, float* __restrict__ loss) {
    __shared__ float wave_sum[4];
    __shared__ bool last;
    const uint32_t r = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    float l = 0.0f;
    if (r < N) {
        const RayTerm t = ray_term(image, wsum, gt, bg, bg_scalar, r);
        l = lambda_rgb * ((t.e[0] * t.e[0] + t.e[1] * t.e[1] + t.e[2] * t.e[2]) / 3.0f) + lambda_mask * (t.m * t.m);
    }
    l = n2m_wave_sum(l);
    if (lane == 0) wave_sum[wid] = l;
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = (wave_sum[0] + wave_sum[1]) + (wave_sum[2] + wave_sum[3]);
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last) {   // the last workgroup to arrive sums the partials in index order: deterministic whatever the arrival order
        __threadfence();
        float s = 0.0f;
        for (uint32_t i = threadIdx.x; i < gridDim.x; i += 256) s += __hip_atomic_load(&partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s = n2m_wave_sum(s);
        if (lane == 0) wave_sum[wid] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            *loss = ((wave_sum[0] + wave_sum[1]) + (wave_sum[2] + wave_sum[3])) / (float)N;
            *ticket = 0;
        }
    }
}

__global__ void __launch_bounds__(256)
photo_loss_backward_kernel(const float* __restrict__ image, const float* __restrict__ wsum, const float* __restrict__ gt,
                           const float* __restrict__ bg, float bg_scalar, float lambda_rgb, float lambda_mask, uint32_t N,
                           const float* __restrict__ grad_loss, float* __restrict__ d_image, float* __restrict__ d_wsum) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= N) return;
    const float g = *grad_loss / (float)N;
    const RayTerm t = ray_term(image, wsum, gt, bg, bg_scalar, r);
    float dw = g * lambda_mask * 2.0f * t.m;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float dp = g * lambda_rgb * (2.0f * t.e[c] / 3.0f);
        d_image[(size_t)r * 3 + c] = dp;
        dw -= dp * t.bg[c];
    }
    d_wsum[r] = dw;
}

// Stage-1 image head in ONE launch, forward and backward (nerf/renderer.py:886-913 after the two antialias calls + the loss of
// nerf/utils.py:708-721): clamp(alpha), clamp(rgb) -> image = alpha * rgb, depth = alpha * z/w, T = 1 - alpha -> ssaa reduction (bilinear
// minification by an integer factor 2 = the mean of the 2 x 2 block in the association torch's kernel uses; nearest for the triangle id) ->
// image + T * bg, weights_sum = 1 - T -> per-pixel loss lambda_rgb * mean_c (image - gt_rgb)^2 + lambda_mask * (weights_sum - gt_a)^2 with
// gt_rgb = gt * gt_a + bg * (1 - gt_a) -> its mean (per-workgroup partials).  The gradient of that mean w.r.t. the two antialias OUTPUTS
// (through the clamps: passed on [0, 1] inclusive like torch.clamp) is written in the same pass -- like the stage-0 head it does not
// depend on the loss value.  The reference spends ~40 full-image elementwise / resize launches (and their autograd nodes) on this.
template <int S>
__global__ void __launch_bounds__(256)
stage1_head_kernel(const float* __restrict__ aa_alpha /*[h0 S, w0 S]*/, const float* __restrict__ aa_rgb /*[h0 S, w0 S, 3]*/,
                   const float* __restrict__ rast /*[h0 S, w0 S, 4]*/, uint32_t h0, uint32_t w0, const float* __restrict__ gt /*[h0 w0, 4]*/,
                   const float* __restrict__ bg /*[h0 w0, 3] or NULL*/, float bg_scalar, float lambda_rgb, float lambda_mask,
                   float* __restrict__ image, float* __restrict__ depth, float* __restrict__ wsum, float* __restrict__ trig_id,
                   float* __restrict__ loss_px, float* __restrict__ d_alpha, float* __restrict__ d_rgb, float* __restrict__ partial,
                   float* __restrict__ tri_err, float* __restrict__ tri_cnt, uint32_t sa /*floats per pixel of aa_alpha / d_alpha: 1, or 4 = channel 3 of an RGBA image*/,
                   uint32_t sr /*of aa_rgb / d_rgb: 3, or 4*/, const float* __restrict__ seed /*factor on the gradients (the loss scale), or NULL*/,
                   float* __restrict__ d_copy /*packed layout only: a second copy of the RGBA gradient image, or NULL*/) {
    __shared__ float wave_sum[4];
    const uint32_t n = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    const uint32_t N = h0 * w0, w = w0 * S;
    float l = 0.0f;
    if (n < N) {
        const uint32_t y = n / w0, x = n - y * w0;
        float a[S * S], c[S * S][3], z[S * S];
        bool pass_a[S * S], pass_c[S * S][3];
#pragma unroll
        for (int j = 0; j < S; ++j)
#pragma unroll
            for (int i = 0; i < S; ++i) {
                const size_t p = (size_t)(y * S + j) * w + (x * S + i);
                const int k = j * S + i;
                const float ra = aa_alpha[p * sa];
                pass_a[k] = ra >= 0.0f && ra <= 1.0f;
                a[k] = fminf(fmaxf(ra, 0.0f), 1.0f);
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const float rc = aa_rgb[p * sr + ch];
                    pass_c[k][ch] = rc >= 0.0f && rc <= 1.0f;
                    c[k][ch] = fminf(fmaxf(rc, 0.0f), 1.0f);
                }
                z[k] = rast[p * 4 + 2];
            }
        // reduction of a quantity q over the block: S = 1 identity; S = 2: 0.5 (0.5 q00 + 0.5 q01) + 0.5 (0.5 q10 + 0.5 q11)
        auto down = [&](auto q) -> float {
            if (S == 1) return q(0);
            return 0.5f * (0.5f * q(0) + 0.5f * q(1)) + 0.5f * (0.5f * q(2) + 0.5f * q(3));
        };
        float im[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) im[ch] = down([&](int k) { return a[k] * c[k][ch]; });
        const float dp = down([&](int k) { return a[k] * z[k]; });
        const float T = down([&](int k) { return 1.0f - a[k]; });
        const float ws = 1.0f - T;
        const float4 g = *reinterpret_cast<const float4*>(gt + (size_t)n * 4);
        const float gc[3] = {g.x, g.y, g.z}, ga = g.w;
        float e[3], bgc[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            bgc[ch] = bg ? bg[(size_t)n * 3 + ch] : bg_scalar;
            im[ch] = im[ch] + T * bgc[ch];
            const float target = gc[ch] * ga + bgc[ch] * (1.0f - ga);
            e[ch] = im[ch] - target;
            image[(size_t)n * 3 + ch] = im[ch];
        }
        const float m = ws - ga;
        l = lambda_rgb * ((e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) / 3.0f) + lambda_mask * (m * m);
        depth[n] = dp; wsum[n] = ws; loss_px[n] = l;
        trig_id[n] = rast[((size_t)(y * S) * w + (size_t)(x * S)) * 4 + 3] - 1.0f;      // nearest: the block's first sub-pixel
        if (tri_err) {      // update_triangles_errors (nerf/renderer.py:924-943): the pixel's loss onto the face visible at it
            const float t = trig_id[n];
            if (t >= 0.0f) { unsafeAtomicAdd(tri_err + (uint32_t)t, l); unsafeAtomicAdd(tri_cnt + (uint32_t)t, 1.0f); }
        }
        if (d_alpha) {
            // d mean / d image_c, d mean / d weights_sum (seed 1 / N; the caller scales by the incoming gradient)
            const float inv = (seed ? *seed : 1.0f) / (float)N;
            float gi[3], gT = -(inv * lambda_mask * 2.0f * m);                         // weights_sum = 1 - T
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) { gi[ch] = inv * lambda_rgb * (2.0f * e[ch] / 3.0f); gT += gi[ch] * bgc[ch]; }
            const float share = S == 1 ? 1.0f : 0.25f;
#pragma unroll
            for (int j = 0; j < S; ++j)
#pragma unroll
                for (int i = 0; i < S; ++i) {
                    const size_t p = (size_t)(y * S + j) * w + (x * S + i);
                    const int k = j * S + i;
                    float da = -(share * gT), dc[3];                                    // T_sub = 1 - alpha
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        const float dis = share * gi[ch];                               // d image_sub
                        da += dis * c[k][ch];
                        dc[ch] = pass_c[k][ch] ? dis * a[k] : 0.0f;
                    }
                    da = pass_a[k] ? da : 0.0f;
                    if (sr == 4u && sa == 4u) {                                         // one RGBA image: a 16-byte store (and its copy)
                        const float4 g4 = make_float4(dc[0], dc[1], dc[2], da);
                        reinterpret_cast<float4*>(d_rgb)[p] = g4;
                        if (d_copy) reinterpret_cast<float4*>(d_copy)[p] = g4;
                    } else {
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) d_rgb[p * sr + ch] = dc[ch];
                        d_alpha[p * sa] = da;
                    }
                }
        }
    }
    l = n2m_wave_sum(l);
    if (lane == 0) wave_sum[wid] = l;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (wave_sum[0] + wave_sum[1]) + (wave_sum[2] + wave_sum[3]);
}

// ------------------------------------------------------------------------------------------------ SDF head (config 5)
// The caller-side arithmetic of the reference's SDF branch (nerf/renderer.py:724-739, nerf/network.py:143-154, nerf/utils.py:740-743) as
// three kernels for the step executor; the torch statement in nerf2mesh_amd/{renderer,network}.py stays the parity baseline.
//   offsets : pts[m, k] = clamp(x[m] +- eps e_axis, -bound, bound) for the six finite-difference copies (k = 2 axis + (minus ? 1 : 0)),
//             and the same points normalised to [0,1] as grid.py:156 does ((p + bound) / (2 bound)).  The copies of a sample are ADJACENT
//             (sample-major [M, 6]; the torch statement stacks them [6, M]): they fall into the same cell on almost every level, so the
//             density encoder's gathers coalesce and its backward merges the six updates of a vertex into one entry
//   forward : normal = 0.5 (s+ - s-) / eps; cos = dir^ . normal^ (safe_normalize both); iter_cos = -(relu(0.5 - 0.5 cos)(1 - car) +
//             relu(-cos) car); inv_s = clip(exp(10 variance), 1e-6, 1e6); p = sigmoid((sdf - iter_cos dt / 2) inv_s), q = sigmoid((sdf +
//             iter_cos dt / 2) inv_s); alpha = clip((p - q + 1e-5) / (p + 1e-5), 0, 1); eikonal partial sums of (|normal| - 1)^2
//   backward: d alpha -> d sdf, d s+-, d variance (per-workgroup partials) + the eikonal term's gradient lambda * 2 (|n| - 1) n / |n| / M
// one thread per OUTPUT value (sample m, copy k, axis a): both stores of a wave are 256 contiguous bytes (one thread per sample wrote 18 values
// at a 72-byte stride per array: 41 us for 38 MB, measured)
__global__ void __launch_bounds__(256)
sdf_offsets_kernel(const float* __restrict__ xyz, uint32_t M, float eps, float bound, float* __restrict__ pts, float* __restrict__ pts01) {
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= (size_t)M * 18u) return;
    const uint32_t m = (uint32_t)(j / 18u), r = (uint32_t)(j - (size_t)m * 18u), k = r / 3u, a = r - k * 3u;
    const float off = (a == (k >> 1)) ? ((k & 1u) ? -eps : eps) : 0.0f;
    const float p = fminf(fmaxf(xyz[3 * (size_t)m + a] + off, -bound), bound);
    pts[j] = p;
    if (pts01) pts01[j] = (p + bound) / (2.0f * bound);
}

struct SdfSample { float n[3], nn, dh[3], nh[3], tc, a, b, ic, s, p, q, raw, sdf, dt; bool s_free; };
__device__ __forceinline__ SdfSample sdf_sample(const float* __restrict__ sdf, const float* __restrict__ s6, const float* __restrict__ dirs,
                                                const float* __restrict__ ts, uint32_t M, uint32_t m, float var, float eps, float car) {
    SdfSample r;
#pragma unroll
    for (int a = 0; a < 3; ++a) r.n[a] = 0.5f * (s6[(size_t)m * 6 + 2 * a] - s6[(size_t)m * 6 + 2 * a + 1]) / eps;
    const float d[3] = {dirs[3 * (size_t)m], dirs[3 * (size_t)m + 1], dirs[3 * (size_t)m + 2]};
    const float dd = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2];
    r.nn = (r.n[0] * r.n[0] + r.n[1] * r.n[1]) + r.n[2] * r.n[2];
    const float dl = sqrtf(fmaxf(dd, 1e-20f)), nl = sqrtf(fmaxf(r.nn, 1e-20f));
#pragma unroll
    for (int a = 0; a < 3; ++a) { r.dh[a] = d[a] / dl; r.nh[a] = r.n[a] / nl; }
    r.tc = (r.dh[0] * r.nh[0] + r.dh[1] * r.nh[1]) + r.dh[2] * r.nh[2];
    r.a = fmaxf(-r.tc * 0.5f + 0.5f, 0.0f);
    r.b = fmaxf(-r.tc, 0.0f);
    r.ic = -(r.a * (1.0f - car) + r.b * car);
    const float e = expf(var * 10.0f);
    r.s = fminf(fmaxf(e, 1e-6f), 1e6f);
    r.s_free = e >= 1e-6f && e <= 1e6f;
    r.sdf = sdf[m];
    r.dt = ts[2 * (size_t)m + 1];
    const float h = r.ic * r.dt * 0.5f;
    r.p = 1.0f / (1.0f + expf(-((r.sdf - h) * r.s)));
    r.q = 1.0f / (1.0f + expf(-((r.sdf + h) * r.s)));
    r.raw = (r.p - r.q + 1e-5f) / (r.p + 1e-5f);
    return r;
}

__global__ void __launch_bounds__(256)
sdf_alpha_forward_kernel(const float* __restrict__ sdf, const float* __restrict__ s6, const float* __restrict__ dirs, const float* __restrict__ ts,
                         uint32_t M, const float* __restrict__ variance, float eps, float car, float* __restrict__ alpha,
                         float* __restrict__ normal, float* __restrict__ eik_partial) {
    __shared__ float wave_sum[4];
    const uint32_t m = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    float e = 0.0f;
    if (m < M) {
        const SdfSample r = sdf_sample(sdf, s6, dirs, ts, M, m, *variance, eps, car);
        alpha[m] = fminf(fmaxf(r.raw, 0.0f), 1.0f);
        if (normal) { normal[3 * (size_t)m] = r.n[0]; normal[3 * (size_t)m + 1] = r.n[1]; normal[3 * (size_t)m + 2] = r.n[2]; }
        const float t = sqrtf(r.nn) - 1.0f;
        e = t * t;
    }
    e = n2m_wave_sum(e);
    if (lane == 0) wave_sum[wid] = e;
    __syncthreads();
    if (threadIdx.x == 0 && eik_partial) eik_partial[blockIdx.x] = (wave_sum[0] + wave_sum[1]) + (wave_sum[2] + wave_sum[3]);
}

__global__ void __launch_bounds__(256)
sdf_alpha_backward_kernel(const float* __restrict__ d_alpha, const float* __restrict__ sdf, const float* __restrict__ s6,
                          const float* __restrict__ dirs, const float* __restrict__ ts, uint32_t M, const float* __restrict__ variance, float eps,
                          float car, const float* __restrict__ seed, float eik_coef /* lambda_eikonal * 2 / M */, float* __restrict__ d_sdf,
                          float* __restrict__ d_s6, float* __restrict__ var_partial) {
    __shared__ float wave_sum[4];
    const uint32_t m = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    float dvar = 0.0f;
    if (m < M) {
        const SdfSample r = sdf_sample(sdf, s6, dirs, ts, M, m, *variance, eps, car);
        const float g = d_alpha[m];
        const float draw = (r.raw >= 0.0f && r.raw <= 1.0f) ? g : 0.0f;             // clip passes the gradient on [0, 1] inclusive
        const float pe = r.p + 1e-5f;
        const float dp = draw * (r.q / (pe * pe)), dq = -draw / pe;
        const float du = dp * r.p * (1.0f - r.p), dv = dq * r.q * (1.0f - r.q);     // through the two sigmoids
        const float h = r.ic * r.dt * 0.5f;
        d_sdf[m] = r.s * (du + dv);
        const float dh = r.s * (dv - du);
        const float ds = du * (r.sdf - h) + dv * (r.sdf + h);
        dvar = r.s_free ? ds * 10.0f * r.s : 0.0f;
        const float dic = dh * r.dt * 0.5f;
        const float da = -(1.0f - car) * dic, db = -car * dic;
        const float dtc = (r.a > 0.0f ? -0.5f * da : 0.0f) + (r.b > 0.0f ? -db : 0.0f);
        // cos = dir^ . normal^ : d normal^ = dtc dir^ ; through safe_normalize (x / sqrt(max(x.x, 1e-20)))
        float dn[3];
        const float nl = sqrtf(fmaxf(r.nn, 1e-20f));
        const float proj = dtc * r.tc;                                             // normal^ . d normal^
#pragma unroll
        for (int a = 0; a < 3; ++a) dn[a] = (r.nn > 1e-20f) ? (dtc * r.dh[a] - r.nh[a] * proj) / nl : dtc * r.dh[a] / nl;
        // eikonal: lambda mean (|n| - 1)^2, torch.linalg.norm (no clamp; zero vector: zero gradient)
        const float nrm = sqrtf(r.nn);
        if (eik_coef != 0.0f && nrm > 0.0f) {
            const float k = *seed * eik_coef * (nrm - 1.0f) / nrm;
#pragma unroll
            for (int a = 0; a < 3; ++a) dn[a] += k * r.n[a];
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float t = 0.5f * dn[a] / eps;
            d_s6[(size_t)m * 6 + 2 * a] = t;
            d_s6[(size_t)m * 6 + 2 * a + 1] = -t;
        }
    }
    dvar = n2m_wave_sum(dvar);
    if (lane == 0) wave_sum[wid] = dvar;
    __syncthreads();
    if (threadIdx.x == 0) var_partial[blockIdx.x] = (wave_sum[0] + wave_sum[1]) + (wave_sum[2] + wave_sum[3]);
}

// out[0] (+)= sum(partial[0 .. n)) in a fixed order; raises found_inf when the sum is not finite
__global__ void __launch_bounds__(256) sum_partials_kernel(const float* __restrict__ partial, uint32_t n, float* __restrict__ out, int add,
                                                           float* __restrict__ found_inf) {
    __shared__ float wave_sum[4];
    float acc = 0.0f;
    for (uint32_t i = threadIdx.x; i < n; i += 256) acc += partial[i];
    acc = n2m_wave_sum(acc);
    if ((threadIdx.x & 63u) == 0) wave_sum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (wave_sum[0] + wave_sum[1]) + (wave_sum[2] + wave_sum[3]);
        out[0] = add ? out[0] + t : t;
        if (found_inf && !(fabsf(t) <= 3.0e38f)) *found_inf = 1.0f;
    }
}

// rays of pixels pix[n] (flat index j*W + i) of views cam[n]: directions ((i+0.5-cx)/fx, -(j+0.5-cy)/fy, -1) rotated by
// the pose (NOT normalised: t is then z-depth, nerf/utils.py:285), origin = pose translation; rgba = images[cam, pix]
__global__ void __launch_bounds__(256)
get_rays_kernel(const float* __restrict__ poses /*[V,4,4]*/, const int64_t* __restrict__ cam, const int64_t* __restrict__ pix, uint32_t N,
                uint32_t W, uint64_t HW, float fx, float fy, float cx, float cy, const float* __restrict__ images /*[V,HW,4] or NULL*/,
                float* __restrict__ rays_o, float* __restrict__ rays_d, float* __restrict__ rgba) {
    const uint32_t n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int64_t v = cam[n], p = pix[n];
    const float i = (float)(p % W) + 0.5f, j = (float)(p / W) + 0.5f;
    const float d0 = (i - cx) / fx, d1 = -(j - cy) / fy, d2 = -1.0f;
    const float* __restrict__ P = poses + (size_t)v * 16;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        rays_d[(size_t)n * 3 + k] = (d0 * P[4 * k] + d1 * P[4 * k + 1]) + d2 * P[4 * k + 2];
        rays_o[(size_t)n * 3 + k] = P[4 * k + 3];
    }
    if (images) *reinterpret_cast<float4*>(rgba + (size_t)n * 4) = *reinterpret_cast<const float4*>(images + ((size_t)v * HW + (size_t)p) * 4);
}


// ------------------------------------------------------------------------------------------------ Adam + loss scaling
// torch.optim.Adam(fused=True) + GradScaler, restated for this step: ONE launch updates every parameter tensor (fp32 master,
// exp_avg, exp_avg_sq; gradients fp32 or fp16, still multiplied by the loss scale), skips everything when *found_inf != 0 and
// can refresh an fp16 shadow copy of a table in the same pass (the colour table is consumed as fp16 by the encoder, grid.py:45).
// A one-thread kernel then does GradScaler.update() and the step count.  Math as in torch's fused kernel:
//   g = grad / scale; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= (lr / (1-b1^t)) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
// Access form of the partner tensor (the [rows,1] table a [rows,2] table's threads update alongside) and of the gradients:
//   0: scalar, cache-allocating (rounds 2-5)              Adam 99.3 us, lookup 71.7 us in the step
//   1: scalar, every pure stream non-temporal              Adam 112 us (+15), lookup 66.0 us (-5.5): a loss of 10 us -- profiles/r06_adam_nt_ab.txt
//   2: the partner's two rows as ONE 8-byte streaming access per array, gradients streaming (round 6 default)
//                                                          Adam 93.5 us (-5.8), lookup 71.1: bit-identical, tests/test_optim.py
#ifndef N2M_ADAM_NT_ALL
#define N2M_ADAM_NT_ALL 2
#endif
struct AdamTensors {
    uint64_t p[N2M_ADAM_MAX], g[N2M_ADAM_MAX], m[N2M_ADAM_MAX], v[N2M_ADAM_MAX], shadow[N2M_ADAM_MAX];
    uint32_t n[N2M_ADAM_MAX], first_block[N2M_ADAM_MAX + 1];
    float lr[N2M_ADAM_MAX];
    uint8_t shadow_mode[N2M_ADAM_MAX];
    uint8_t slot[N2M_ADAM_MAX];       // 0: bias[0..1] (one step count for everything); s > 0: bias[2s..2s+1], this tensor's own count
    int8_t partner[N2M_ADAM_MAX];     // mode-3 tensor: index of the mode-2 tensor packed into the same table (updated by the same threads), or -1
    uint32_t count, g_half_mask, clear_mask;    // clear_mask: fp32 gradients this launch resets to zero once consumed
};

// refresh the working copy of elements i0 .. i0+cnt-1 (cnt <= 4): plain fp16 copy, or one column of a packed table whose rows
// are 8 bytes {fp32 from a [rows,1] table, half2 from a [rows,2] table}
__device__ __forceinline__ void shadow_store(_Float16* S, uint32_t mode, uint32_t i0, const float (&p)[4], uint32_t cnt) {
    if (mode == 2u) {
        float* F = reinterpret_cast<float*>(S);
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e)
            if (e < cnt) F[(size_t)(i0 + e) * 2u] = p[e];
    } else if (mode == 3u) {                               // i0 is a multiple of 4: elements (2r, 2r+1) are row r
        typedef _Float16 h2v __attribute__((ext_vector_type(2)));
        uint32_t* U = reinterpret_cast<uint32_t*>(S);
#pragma unroll
        for (uint32_t e = 0; e < 4; e += 2) {
            if (e + 1 < cnt) {
                h2v o;
                o.x = (_Float16)p[e]; o.y = (_Float16)p[e + 1];
                U[(size_t)((i0 + e) >> 1) * 2u + 1u] = __builtin_bit_cast(uint32_t, o);
            }
        }
    } else if (cnt == 4u) {
        typedef _Float16 h4v __attribute__((ext_vector_type(4)));
        h4v o;
        o.x = (_Float16)p[0]; o.y = (_Float16)p[1]; o.z = (_Float16)p[2]; o.w = (_Float16)p[3];
        *reinterpret_cast<h4v*>(S + i0) = o;
    } else {
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e)
            if (e < cnt) S[i0 + e] = (_Float16)p[e];
    }
}

// Peer-store form (n2m_adam_step_peer, include/n2m_peer.h): a table entry's gradient is the rank-order sum of W staging slots (what
// n2m_peer_reduce_slices would have written for n2m_adam_step to read), and every packed row the pass refreshes is also stored into the other
// ranks' packed tables (what n2m_peer_copy would have sent afterwards).  Same arithmetic, element for element; two passes over the rows fewer.
struct AdamPeerK {
    uint32_t world, n_remote;
    int8_t entry[N2M_ADAM_MAX];                  // tensor k -> row of `slots`, or -1 (reads t.g[k])
    uint64_t slots[4][N2M_PEER_MAX];             // the W slots of that entry, in this rank's own staging memory
    long long remote_delta[N2M_PEER_MAX];        // byte offset from this rank's packed table to rank r's mapping of its own
};

__device__ __forceinline__ uint32_t peer_load32(const void* at) {      // the slots were written by other agents: bypass the caches
    return __hip_atomic_load(reinterpret_cast<const uint32_t*>(at), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Wave-uniform state of the step (found_inf, bias corrections, loss scale) through the scalar data cache: in a kernel that also WRITES those
// words (adam_kernel_with_scaler's last workgroup) the compiler must otherwise fetch them with vector loads -- three dependent L2 round trips at the
// top of every one of 18 000 short-lived workgroups (measured: the pass 136 us instead of 93).  Nobody writes them before every workgroup has read them.
__device__ __forceinline__ float uniform_f32(const float* p) {
    return *(const __attribute__((address_space(4))) float*)(p);
}

template <bool PEER>
__device__ __forceinline__ void
adam_body(const AdamTensors& t, float beta1, float beta2, float omb1, float omb2, float eps, const float* scale,
          const float* found_inf, const float* bias /*[2]: 1-b1^t, sqrt(1-b2^t) of THIS step*/, const AdamPeerK& pe) {
    uint32_t k = 0;
    while (k + 1 < t.count && blockIdx.x >= t.first_block[k + 1]) ++k;
    const uint32_t i0 = ((blockIdx.x - t.first_block[k]) * 256u + threadIdx.x) * 4u;
    const uint32_t n = t.n[k];
    if (i0 >= n) return;
    const bool clear_g = (t.clear_mask >> k) & 1u;                     // consume-and-clear: the producer accumulates into a persistent buffer
    if (found_inf && uniform_f32(found_inf) != 0.0f) {                     // GradScaler: skip the whole step (the gradients are still consumed)
        if (clear_g)
            for (uint32_t e = 0; e < 4u && i0 + e < n; ++e) reinterpret_cast<float*>(t.g[k])[i0 + e] = 0.0f;
        return;
    }
    const float bc1 = uniform_f32(bias + 2u * t.slot[k]), bc2_sqrt = uniform_f32(bias + 2u * t.slot[k] + 1u);
    const float step_size = t.lr[k] / bc1;
    const float inv_scale = scale ? 1.0f / uniform_f32(scale) : 1.0f;
    float* __restrict__ P = reinterpret_cast<float*>(t.p[k]);
    float* __restrict__ M = reinterpret_cast<float*>(t.m[k]);
    float* __restrict__ V = reinterpret_cast<float*>(t.v[k]);
    _Float16* __restrict__ S = reinterpret_cast<_Float16*>(t.shadow[k]);
    const uint32_t smode = t.shadow_mode[k];
    const bool g_half = (t.g_half_mask >> k) & 1u;
    float p[4], m[4], v[4], g[4];
    const bool full = i0 + 4u <= n;                                   // tensors are 16-byte aligned (torch allocations)
    // the partner tensor's two rows (see below) are requested together with this tensor's values: one memory round trip per thread, not two
    const int pk = t.partner[k];
    const uint32_t r0 = i0 >> 1, n1 = pk >= 0 ? t.n[pk] : 0u;
    float p1[2] = {0.f, 0.f}, m1v[2] = {0.f, 0.f}, v1v[2] = {0.f, 0.f}, g1v[2] = {0.f, 0.f};
    if (pk >= 0) {
        const float* __restrict__ P1 = reinterpret_cast<const float*>(t.p[pk]);
        const float* __restrict__ M1 = reinterpret_cast<const float*>(t.m[pk]);
        const float* __restrict__ V1 = reinterpret_cast<const float*>(t.v[pk]);
        const bool g1_half = (t.g_half_mask >> pk) & 1u;
        bool pair8 = false;
#if N2M_ADAM_NT_ALL == 2
        // the partner's two rows as ONE 8-byte streaming access per array (r0 is even: 8-byte aligned whenever the slice starts on an even row)
        typedef float f2v __attribute__((ext_vector_type(2)));
        pair8 = r0 + 2u <= n1 && ((t.p[pk] | t.m[pk] | t.v[pk] | t.g[pk]) & 7u) == 0u && !g1_half && !(PEER && pe.entry[pk] >= 0);
        if (pair8) {
            const f2v a = __builtin_nontemporal_load(reinterpret_cast<const f2v*>(P1 + r0)), b = __builtin_nontemporal_load(reinterpret_cast<const f2v*>(M1 + r0)),
                      c = __builtin_nontemporal_load(reinterpret_cast<const f2v*>(V1 + r0)),
                      d = __builtin_nontemporal_load(reinterpret_cast<const f2v*>(reinterpret_cast<const float*>(t.g[pk]) + r0));
            p1[0] = a.x; p1[1] = a.y; m1v[0] = b.x; m1v[1] = b.y; v1v[0] = c.x; v1v[1] = c.y; g1v[0] = d.x; g1v[1] = d.y;
        }
#endif
#pragma unroll
        for (uint32_t e = 0; e < 2; ++e) {
            if (pair8 || r0 + e >= n1) continue;
#if N2M_ADAM_NT_ALL == 1      // every pure stream of the pass bypasses the caches; only the packed rows (what the next lookup gathers) allocate
            p1[e] = __builtin_nontemporal_load(P1 + r0 + e); m1v[e] = __builtin_nontemporal_load(M1 + r0 + e); v1v[e] = __builtin_nontemporal_load(V1 + r0 + e);
#else
            p1[e] = P1[r0 + e]; m1v[e] = M1[r0 + e]; v1v[e] = V1[r0 + e];
#endif
            if (PEER && pe.entry[pk] >= 0) {               // (fp32 [rows,1] table: one value per row and slot)
                float a = 0.0f;
                for (uint32_t sl = 0; sl < pe.world; ++sl) a += __uint_as_float(peer_load32(reinterpret_cast<const float*>(pe.slots[pe.entry[pk]][sl]) + r0 + e));
                g1v[e] = a;
                continue;
            }
#if N2M_ADAM_NT_ALL == 1
            g1v[e] = g1_half ? (float)__builtin_nontemporal_load(reinterpret_cast<const _Float16*>(t.g[pk]) + r0 + e)
                             : __builtin_nontemporal_load(reinterpret_cast<const float*>(t.g[pk]) + r0 + e);
#else
            g1v[e] = g1_half ? (float)reinterpret_cast<const _Float16*>(t.g[pk])[r0 + e] : reinterpret_cast<const float*>(t.g[pk])[r0 + e];
#endif
        }
    }
    typedef float f4v __attribute__((ext_vector_type(4)));
    if (full) {
        // streaming (non-temporal) accesses for the optimizer state: 540 MB pass through once per step, and the marcher that runs beside
        // this kernel lives on a 256 KB bit field it wants to keep in L2
        const f4v pp = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(P + i0)), mm = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(M + i0)),
                  vv = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(V + i0));
        p[0] = pp.x; p[1] = pp.y; p[2] = pp.z; p[3] = pp.w;
        m[0] = mm.x; m[1] = mm.y; m[2] = mm.z; m[3] = mm.w;
        v[0] = vv.x; v[1] = vv.y; v[2] = vv.z; v[3] = vv.w;
        if (PEER && pe.entry[k] >= 0) {
            // rank-order sum of the W slots; an fp16 gradient is summed in fp32 and rounded to half ONCE, like n2m_peer_reduce_slices
            typedef _Float16 h2p __attribute__((ext_vector_type(2)));
            float a[4] = {0.f, 0.f, 0.f, 0.f};
            for (uint32_t sl = 0; sl < pe.world; ++sl) {
                const char* base = reinterpret_cast<const char*>(pe.slots[pe.entry[k]][sl]);
                if (g_half) {
                    const h2p u0 = __builtin_bit_cast(h2p, peer_load32(base + (size_t)i0 * 2u)), u1 = __builtin_bit_cast(h2p, peer_load32(base + (size_t)i0 * 2u + 4u));
                    a[0] += (float)u0.x; a[1] += (float)u0.y; a[2] += (float)u1.x; a[3] += (float)u1.y;
                } else {
#pragma unroll
                    for (uint32_t e = 0; e < 4; ++e) a[e] += __uint_as_float(peer_load32(base + ((size_t)i0 + e) * 4u));
                }
            }
#pragma unroll
            for (uint32_t e = 0; e < 4; ++e) g[e] = g_half ? (float)(_Float16)a[e] : a[e];
        } else if (g_half) {
            typedef _Float16 h4v __attribute__((ext_vector_type(4)));
#if N2M_ADAM_NT_ALL
            const h4v gg = __builtin_nontemporal_load(reinterpret_cast<const h4v*>(reinterpret_cast<const _Float16*>(t.g[k]) + i0));
#else
            const h4v gg = *reinterpret_cast<const h4v*>(reinterpret_cast<const _Float16*>(t.g[k]) + i0);
#endif
            g[0] = (float)gg.x; g[1] = (float)gg.y; g[2] = (float)gg.z; g[3] = (float)gg.w;
        } else {
#if N2M_ADAM_NT_ALL
            const f4v gg = clear_g ? *reinterpret_cast<const f4v*>(reinterpret_cast<const float*>(t.g[k]) + i0)
                                   : __builtin_nontemporal_load(reinterpret_cast<const f4v*>(reinterpret_cast<const float*>(t.g[k]) + i0));
#else
            const float4 gg = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(t.g[k]) + i0);
#endif
            g[0] = gg.x; g[1] = gg.y; g[2] = gg.z; g[3] = gg.w;
        }
    } else {
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e) {
            const bool ok = i0 + e < n;
            p[e] = ok ? P[i0 + e] : 0.f; m[e] = ok ? M[i0 + e] : 0.f; v[e] = ok ? V[i0 + e] : 0.f;
            g[e] = !ok ? 0.f : (g_half ? (float)reinterpret_cast<const _Float16*>(t.g[k])[i0 + e] : reinterpret_cast<const float*>(t.g[k])[i0 + e]);
        }
    }
#pragma unroll
    for (uint32_t e = 0; e < 4; ++e) {
        const float gr = g[e] * inv_scale;
        m[e] = beta1 * m[e] + omb1 * gr;              // omb = 1 - beta rounded from double: 1.0f - 0.999f is off by 5e-5 relative
        v[e] = beta2 * v[e] + omb2 * gr * gr;
        const float denom = sqrtf(v[e]) / bc2_sqrt + eps;
        p[e] -= step_size * m[e] / denom;
    }
    if (clear_g)
        for (uint32_t e = 0; e < 4u && i0 + e < n; ++e) reinterpret_cast<float*>(t.g[k])[i0 + e] = 0.0f;
    if (full) {
        const f4v po = {p[0], p[1], p[2], p[3]}, mo = {m[0], m[1], m[2], m[3]}, vo = {v[0], v[1], v[2], v[3]};
        __builtin_nontemporal_store(po, reinterpret_cast<f4v*>(P + i0));
        __builtin_nontemporal_store(mo, reinterpret_cast<f4v*>(M + i0));
        __builtin_nontemporal_store(vo, reinterpret_cast<f4v*>(V + i0));
        if (S && pk < 0) shadow_store(S, smode, i0, p, 4u);
    } else {
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e)
            if (i0 + e < n) { P[i0 + e] = p[e]; M[i0 + e] = m[e]; V[i0 + e] = v[e]; }
        if (S && pk < 0) shadow_store(S, smode, i0, p, n - i0);
    }
    if (pk >= 0) {
        // this [rows,2] tensor shares a packed table with the [rows,1] tensor `pk`: the same thread updates rows r0, r0+1 of that
        // tensor too and writes the two complete 8-byte packed rows with one 16-byte store (separate column writes at an 8-byte
        // stride cost 22 us more, measured)
        float* __restrict__ P1 = reinterpret_cast<float*>(t.p[pk]);
        float* __restrict__ M1 = reinterpret_cast<float*>(t.m[pk]);
        float* __restrict__ V1 = reinterpret_cast<float*>(t.v[pk]);
        const float bc1p = uniform_f32(bias + 2u * t.slot[pk]), bc2p_sqrt = uniform_f32(bias + 2u * t.slot[pk] + 1u);      // the partner tensor's own step count
        const float step1 = t.lr[pk] / bc1p;
        float q[2] = {0.f, 0.f};
#if N2M_ADAM_NT_ALL == 2
        typedef float f2s __attribute__((ext_vector_type(2)));
        const bool st8 = r0 + 2u <= n1 && ((t.p[pk] | t.m[pk] | t.v[pk]) & 7u) == 0u;
        float mo[2] = {0.f, 0.f}, vo[2] = {0.f, 0.f};
#endif
#pragma unroll
        for (uint32_t e = 0; e < 2; ++e) {
            if (r0 + e >= n1) continue;
            const float gr = g1v[e] * inv_scale;
            const float m1 = beta1 * m1v[e] + omb1 * gr;
            const float v1 = beta2 * v1v[e] + omb2 * gr * gr;
            q[e] = p1[e] - step1 * m1 / (sqrtf(v1) / bc2p_sqrt + eps);
#if N2M_ADAM_NT_ALL == 2
            mo[e] = m1; vo[e] = v1;
            if (!st8) { P1[r0 + e] = q[e]; M1[r0 + e] = m1; V1[r0 + e] = v1; }
#elif N2M_ADAM_NT_ALL == 1
            __builtin_nontemporal_store(q[e], P1 + r0 + e); __builtin_nontemporal_store(m1, M1 + r0 + e); __builtin_nontemporal_store(v1, V1 + r0 + e);
#else
            P1[r0 + e] = q[e]; M1[r0 + e] = m1; V1[r0 + e] = v1;
#endif
        }
#if N2M_ADAM_NT_ALL == 2
        if (st8) {
            __builtin_nontemporal_store((f2s){q[0], q[1]}, reinterpret_cast<f2s*>(P1 + r0));
            __builtin_nontemporal_store((f2s){mo[0], mo[1]}, reinterpret_cast<f2s*>(M1 + r0));
            __builtin_nontemporal_store((f2s){vo[0], vo[1]}, reinterpret_cast<f2s*>(V1 + r0));
        }
#endif
        typedef _Float16 h2v __attribute__((ext_vector_type(2)));
        h2v c0, c1;
        c0.x = (_Float16)p[0]; c0.y = (_Float16)p[1]; c1.x = (_Float16)p[2]; c1.y = (_Float16)p[3];
        uint32_t* U = reinterpret_cast<uint32_t*>(S) + (size_t)r0 * 2u;
        if (full && r0 + 2u <= n1) {
            const uint4 rows2 = make_uint4(__float_as_uint(q[0]), __builtin_bit_cast(uint32_t, c0), __float_as_uint(q[1]), __builtin_bit_cast(uint32_t, c1));
            *reinterpret_cast<uint4*>(U) = rows2;
            if (PEER)
                for (uint32_t r = 0; r < pe.n_remote; ++r) *reinterpret_cast<uint4*>(reinterpret_cast<char*>(U) + pe.remote_delta[r]) = rows2;
        } else {
            if (i0 + 1u < n) { U[1] = __builtin_bit_cast(uint32_t, c0); if (r0 < n1) U[0] = __float_as_uint(q[0]); }
            if (i0 + 3u < n) { U[3] = __builtin_bit_cast(uint32_t, c1); if (r0 + 1u < n1) U[2] = __float_as_uint(q[1]); }
        }
    }
}

// GradScaler.update() (torch/amp/grad_scaler.py:_amp_update_scale_) + step count + found_inf reset, one thread
__global__ void scaler_update_kernel(float* scale, float* growth_tracker, float* found_inf, float* step, float* bias, double beta1,
                                     double beta2, float growth_factor, float backoff_factor, float growth_interval) {
    if (*found_inf != 0.0f) {
        if (scale) *scale *= backoff_factor;
        if (growth_tracker) *growth_tracker = 0.0f;
    } else {
        if (step) *step += 1.0f;
        if (growth_tracker) {
            const float ok = *growth_tracker + 1.0f;
            if (ok >= growth_interval) {
                const float ns = *scale * growth_factor;
                if (scale && ns <= 3.0e38f) *scale = ns;            // do not grow into inf
                *growth_tracker = 0.0f;
            } else {
                *growth_tracker = ok;
            }
        }
    }
    *found_inf = 0.0f;
    if (step && bias) {                       // bias corrections of the NEXT step, in double like torch's host-side arithmetic
        const double t = (double)*step + 1.0;
        bias[0] = (float)(1.0 - pow(beta1, t));
        bias[1] = (float)sqrt(1.0 - pow(beta2, t));
    }
}

}  // namespace

extern "C" int n2m_get_rays(const float* poses, const int64_t* cam, const int64_t* pix, uint32_t N, uint32_t H, uint32_t W, float fx, float fy,
                            float cx, float cy, const float* images, float* rays_o, float* rays_d, float* rgba, void* stream) {
    N2M_REQUIRE(poses && cam && pix && rays_o && rays_d, N2M_ENULL, "get_rays: NULL tensor");
    N2M_REQUIRE((images == nullptr) == (rgba == nullptr), N2M_EINVAL, "get_rays: images and rgba go together");
    N2M_REQUIRE(W > 0 && H > 0 && fx != 0.f && fy != 0.f, N2M_EINVAL, "get_rays: bad intrinsics");
    if (N == 0) return 0;
    get_rays_kernel<<<n2m_ceil_div(N, 256), 256, 0, (hipStream_t)stream>>>(poses, cam, pix, N, W, (uint64_t)H * W, fx, fy, cx, cy, images, rays_o,
                                                                          rays_d, rgba);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_photo_loss_forward(const float* image, const float* weights_sum, const float* gt_rgba, const float* bg, float bg_scalar,
                                      float lambda_rgb, float lambda_mask, uint32_t N, float* partial, uint32_t* ticket, float* loss,
                                      void* stream) {
    N2M_REQUIRE(image && weights_sum && gt_rgba && partial && ticket && loss, N2M_ENULL, "photo_loss_forward: NULL tensor");
    N2M_REQUIRE(N > 0, N2M_EINVAL, "photo_loss_forward: N must be positive");
    hipStream_t s = (hipStream_t)stream;
    photo_loss_forward_kernel<<<n2m_ceil_div(N, 256), 256, 0, s>>>(image, weights_sum, gt_rgba, bg, bg_scalar, lambda_rgb, lambda_mask, N, partial,
                                                                    ticket, loss);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_photo_loss_backward(const float* image, const float* weights_sum, const float* gt_rgba, const float* bg, float bg_scalar,
                                       float lambda_rgb, float lambda_mask, uint32_t N, const float* grad_loss, float* d_image,
                                       float* d_weights_sum, void* stream) {
    N2M_REQUIRE(image && weights_sum && gt_rgba && grad_loss && d_image && d_weights_sum, N2M_ENULL, "photo_loss_backward: NULL tensor");
    if (N == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    photo_loss_backward_kernel<<<n2m_ceil_div(N, 256), 256, 0, s>>>(image, weights_sum, gt_rgba, bg, bg_scalar, lambda_rgb, lambda_mask, N, grad_loss,
                                                                     d_image, d_weights_sum);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_sdf_offsets(const float* xyz, uint32_t M, float eps, float bound, float* pts, float* pts01, void* stream) {
    N2M_REQUIRE(xyz && pts, N2M_ENULL, "sdf_offsets: NULL tensor");      // pts01 = NULL: a caller that encodes from its own lists (n2m_sdf_fold_*)
    N2M_REQUIRE(eps > 0.0f && bound > 0.0f, N2M_EINVAL, "sdf_offsets: eps and bound must be positive");
    if (M == 0) return 0;
    sdf_offsets_kernel<<<n2m_ceil_div((uint64_t)M * 18u, 256), 256, 0, (hipStream_t)stream>>>(xyz, M, eps, bound, pts, pts01);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_sdf_alpha_forward(const float* sdf, const float* sdf6, const float* dirs, const float* ts, uint32_t M, const float* variance,
                                     float eps, float cos_anneal_ratio, float* alpha, float* normal, float* eik_partial, void* stream) {
    N2M_REQUIRE(sdf && sdf6 && dirs && ts && variance && alpha, N2M_ENULL, "sdf_alpha_forward: NULL tensor");
    if (M == 0) return 0;
    sdf_alpha_forward_kernel<<<n2m_ceil_div(M, 256), 256, 0, (hipStream_t)stream>>>(sdf, sdf6, dirs, ts, M, variance, eps, cos_anneal_ratio, alpha,
                                                                                  normal, eik_partial);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_sdf_alpha_backward(const float* d_alpha, const float* sdf, const float* sdf6, const float* dirs, const float* ts, uint32_t M,
                                      const float* variance, float eps, float cos_anneal_ratio, const float* seed, float eik_coef, float* d_sdf,
                                      float* d_sdf6, float* var_partial, float* d_variance, float* found_inf, void* stream) {
    N2M_REQUIRE(d_alpha && sdf && sdf6 && dirs && ts && variance && d_sdf && d_sdf6 && var_partial && d_variance, N2M_ENULL,
                "sdf_alpha_backward: NULL tensor");
    N2M_REQUIRE(eik_coef == 0.0f || seed, N2M_ENULL, "sdf_alpha_backward: the eikonal term needs the seed gradient");
    hipStream_t s = (hipStream_t)stream;
    const uint32_t nb = n2m_ceil_div(M, 256);
    if (M > 0) sdf_alpha_backward_kernel<<<nb, 256, 0, s>>>(d_alpha, sdf, sdf6, dirs, ts, M, variance, eps, cos_anneal_ratio, seed, eik_coef, d_sdf,
                                                           d_sdf6, var_partial);
    sum_partials_kernel<<<1, 256, 0, s>>>(var_partial, M > 0 ? nb : 0u, d_variance, 0, found_inf);
    N2M_CHECK_LAUNCH();
    return 0;
}

// ---- world -> clip space of the stage-1 mesh (nerf/renderer.py:858: [v, 1] @ mvp^T), value and gradient w.r.t. the vertices, one launch each.
// Forward keeps the association of the broadcast form it replaces, ((v0 m_0 + v1 m_1) + v2 m_2) + m_3 per output column, no contraction.
__global__ void __launch_bounds__(256)
to_clip_kernel(const float* __restrict__ v, const float* __restrict__ mvp /*[4,4] row-major*/, uint32_t V, float* __restrict__ clip) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= V) return;
    const float x = v[(size_t)i * 3u], y = v[(size_t)i * 3u + 1], z = v[(size_t)i * 3u + 2];
    float4 o;
    o.x = ((x * mvp[0] + y * mvp[1]) + z * mvp[2]) + mvp[3];
    o.y = ((x * mvp[4] + y * mvp[5]) + z * mvp[6]) + mvp[7];
    o.z = ((x * mvp[8] + y * mvp[9]) + z * mvp[10]) + mvp[11];
    o.w = ((x * mvp[12] + y * mvp[13]) + z * mvp[14]) + mvp[15];
    *reinterpret_cast<float4*>(clip + (size_t)i * 4u) = o;
}

__global__ void __launch_bounds__(256)
to_clip_backward_kernel(const float* __restrict__ d_clip, const float* __restrict__ mvp, uint32_t V, float* __restrict__ d_v) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= V) return;
    const float4 g = *reinterpret_cast<const float4*>(d_clip + (size_t)i * 4u);
#pragma unroll
    for (uint32_t a = 0; a < 3; ++a)
        d_v[(size_t)i * 3u + a] = ((g.x * mvp[a] + g.y * mvp[4 + a]) + g.z * mvp[8 + a]) + g.w * mvp[12 + a];
}

extern "C" int n2m_to_clip(const float* vertices, const float* mvp, uint32_t V, float* clip, void* stream) {
    N2M_REQUIRE(vertices && mvp && clip, N2M_ENULL, "to_clip: NULL tensor");
    N2M_REQUIRE(((uintptr_t)clip & 15u) == 0, N2M_EINVAL, "to_clip: clip must be 16-byte aligned");
    if (V == 0) return 0;
    to_clip_kernel<<<n2m_ceil_div(V, 256), 256, 0, (hipStream_t)stream>>>(vertices, mvp, V, clip);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_to_clip_backward(const float* d_clip, const float* mvp, uint32_t V, float* d_vertices, void* stream) {
    N2M_REQUIRE(d_clip && mvp && d_vertices, N2M_ENULL, "to_clip_backward: NULL tensor");
    N2M_REQUIRE(((uintptr_t)d_clip & 15u) == 0, N2M_EINVAL, "to_clip_backward: d_clip must be 16-byte aligned");
    if (V == 0) return 0;
    to_clip_backward_kernel<<<n2m_ceil_div(V, 256), 256, 0, (hipStream_t)stream>>>(d_clip, mvp, V, d_vertices);
    N2M_CHECK_LAUNCH();
    return 0;
}

// ---- the two mesh regularisers of stage 1 in one launch each way.  Uniform-Laplacian smoothness (nerf/utils.py:176-221):
// mean_i || deg_i v_i - sum_{j in N(i)} v_j ||_2 over the unique directed edges, as CSR (row_ptr [V + 1], col [E], neighbours of a vertex in
// ascending order: a fixed summation order); offset penalty (nerf/utils.py:772-789): mean_i |off_i|^2, with bound > 1 the inner mesh's mean +
// 0.1 x the outer meshes' (vertices [0, n_in) are the inner ones).  forward: L = D v - A v, its norms, per-workgroup sums of
// lam_lap / V * norm_i + w_i |off_i|^2 (w = lam_off / count of the vertex's group [x 0.1]); backward: d v_i = deg_i gL_i - sum_{j in N(i)} gL_j
// with gL_k = g lam_lap / V * L_k / ||L_k|| (0 where the norm is 0, like torch's norm backward; the adjacency is symmetric, so the adjoint is
// the same walk) and d off_i = g w_i 2 off_i.
__global__ void __launch_bounds__(256)
laplacian_forward_kernel(const float* __restrict__ v, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col, uint32_t V,
                         const float* __restrict__ off, float lam_lap, float w_in, float w_out, uint32_t n_in,
                         float* __restrict__ Lv, float* __restrict__ norm, float* __restrict__ partial) {
    __shared__ float red[4];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    float term = 0.0f;
    if (i < V) {
        const int32_t b = row_ptr[i], e = row_ptr[i + 1];
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int32_t k = b; k < e; ++k) {
            const size_t j = (size_t)col[k] * 3u;
            s0 += v[j]; s1 += v[j + 1]; s2 += v[j + 2];
        }
        const float d = (float)(e - b);
        const float l0 = v[(size_t)i * 3u] * d - s0, l1 = v[(size_t)i * 3u + 1] * d - s1, l2 = v[(size_t)i * 3u + 2] * d - s2;
        const float n = sqrtf((l0 * l0 + l1 * l1) + l2 * l2);
        Lv[(size_t)i * 3u] = l0; Lv[(size_t)i * 3u + 1] = l1; Lv[(size_t)i * 3u + 2] = l2;
        norm[i] = n;
        term = n * (lam_lap / (float)V);
        if (off) {
            const float o0 = off[(size_t)i * 3u], o1 = off[(size_t)i * 3u + 1], o2 = off[(size_t)i * 3u + 2];
            term += ((o0 * o0 + o1 * o1) + o2 * o2) * (i < n_in ? w_in : w_out);
        }
    }
    const float w = n2m_wave_sum(term);
    if ((threadIdx.x & 63u) == 0u) red[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0u) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

template <bool ACC>
__global__ void __launch_bounds__(256)
laplacian_backward_kernel(const float* __restrict__ Lv, const float* __restrict__ norm, const int32_t* __restrict__ row_ptr,
                          const int32_t* __restrict__ col, uint32_t V, const float* __restrict__ g, float lam_lap,
                          const float* __restrict__ off, float w_in, float w_out, uint32_t n_in, float* __restrict__ d_v, float* __restrict__ d_off,
                          float* __restrict__ found_inf) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= V) return;
    const float gs = *g * (lam_lap / (float)V);
    auto gl = [&](size_t k, float (&o)[3]) {
        const float n = norm[k];
        const float f = n > 0.0f ? gs / n : 0.0f;
        o[0] = Lv[k * 3u] * f; o[1] = Lv[k * 3u + 1] * f; o[2] = Lv[k * 3u + 2] * f;
    };
    const int32_t b = row_ptr[i], e = row_ptr[i + 1];
    float s[3] = {0.f, 0.f, 0.f}, t[3];
    for (int32_t k = b; k < e; ++k) {
        gl((size_t)col[k], t);
        s[0] += t[0]; s[1] += t[1]; s[2] += t[2];
    }
    gl((size_t)i, t);
    const float d = (float)(e - b);
    if (ACC) {
        // d_v holds the rendering gradient of the vertices: the sum ((d_v + smoothness) + offset penalty), in that order, and the non-finite flag
        const float f = *g * (i < n_in ? w_in : w_out) * 2.0f;
        bool bad = false;
#pragma unroll
        for (uint32_t a = 0; a < 3; ++a) {
            const float r = (d_v[(size_t)i * 3u + a] + (t[a] * d - s[a])) + off[(size_t)i * 3u + a] * f;
            d_v[(size_t)i * 3u + a] = r;
            bad |= !isfinite(r);
        }
        if (bad && found_inf) *found_inf = 1.0f;
        return;
    }
    d_v[(size_t)i * 3u] = t[0] * d - s[0]; d_v[(size_t)i * 3u + 1] = t[1] * d - s[1]; d_v[(size_t)i * 3u + 2] = t[2] * d - s[2];
    if (d_off) {
        const float f = *g * (i < n_in ? w_in : w_out) * 2.0f;
#pragma unroll
        for (uint32_t a = 0; a < 3; ++a) d_off[(size_t)i * 3u + a] = off[(size_t)i * 3u + a] * f;
    }
}

extern "C" int n2m_laplacian_forward(const float* verts, const int32_t* row_ptr, const int32_t* col, uint32_t V, const float* offsets, float lam_lap,
                                     float w_in, float w_out, uint32_t n_in, float* Lv, float* norm, float* partial, void* stream) {
    N2M_REQUIRE(verts && row_ptr && col && Lv && norm && partial, N2M_ENULL, "laplacian_forward: NULL tensor");
    if (V == 0) return 0;
    laplacian_forward_kernel<<<n2m_ceil_div(V, 256), 256, 0, (hipStream_t)stream>>>(verts, row_ptr, col, V, offsets, lam_lap, w_in, w_out, n_in, Lv, norm,
                                                                                   partial);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_laplacian_backward(const float* Lv, const float* norm, const int32_t* row_ptr, const int32_t* col, uint32_t V, const float* grad,
                                      float lam_lap, const float* offsets, float w_in, float w_out, uint32_t n_in, float* d_verts, float* d_offsets,
                                      void* stream) {
    N2M_REQUIRE(Lv && norm && row_ptr && col && grad && d_verts, N2M_ENULL, "laplacian_backward: NULL tensor");
    N2M_REQUIRE((offsets == nullptr) == (d_offsets == nullptr), N2M_ENULL, "laplacian_backward: offsets and d_offsets come together");
    if (V == 0) return 0;
    laplacian_backward_kernel<false><<<n2m_ceil_div(V, 256), 256, 0, (hipStream_t)stream>>>(Lv, norm, row_ptr, col, V, grad, lam_lap, offsets, w_in, w_out,
                                                                                           n_in, d_verts, d_offsets, nullptr);
    N2M_CHECK_LAUNCH();
    return 0;
}

// The same gradients ADDED onto `d_verts` (which already holds the rendering gradient of the vertex positions): with vertices = base + offsets all
// three land on the offsets (nerf/renderer.py:855, nerf/utils.py:761-789), so d_verts <- (d_verts + smoothness) + offset penalty in one pass, and
// `found_inf` (optional) is raised when the sum is not finite -- the check torch.amp's unscale_ would make on that gradient.
extern "C" int n2m_laplacian_backward_acc(const float* Lv, const float* norm, const int32_t* row_ptr, const int32_t* col, uint32_t V, const float* grad,
                                          float lam_lap, const float* offsets, float w_in, float w_out, uint32_t n_in, float* d_verts, float* found_inf,
                                          void* stream) {
    N2M_REQUIRE(Lv && norm && row_ptr && col && grad && d_verts && offsets, N2M_ENULL, "laplacian_backward_acc: NULL tensor");
    if (V == 0) return 0;
    laplacian_backward_kernel<true><<<n2m_ceil_div(V, 256), 256, 0, (hipStream_t)stream>>>(Lv, norm, row_ptr, col, V, grad, lam_lap, offsets, w_in, w_out,
                                                                                          n_in, d_verts, nullptr, found_inf);
    N2M_CHECK_LAUNCH();
    return 0;
}

// rows of a [N, C] fp32 array by index (stage 1: the covered pixels of a frame, nerf/renderer.py:864-881): out[k] = x[idx[k]] and its mirror
// dst[idx[k]] = src[k] (idx unique: no atomics).  torch's index kernels spend 50 us per call on 0.7 M rows of three floats.
template <bool SCATTER>
__global__ void __launch_bounds__(256)
rows_by_index_kernel(const float* __restrict__ in, const int64_t* __restrict__ idx, uint32_t K, uint32_t C, uint32_t in_stride, uint32_t out_stride,
                     float* __restrict__ out) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= K) return;
    const size_t r = (size_t)idx[k];
    const float* src = in + (SCATTER ? (size_t)k : r) * in_stride;
    float* dst = out + (SCATTER ? r : (size_t)k) * out_stride;
    if (C == 3u) { const float a = src[0], b = src[1], c = src[2]; dst[0] = a; dst[1] = b; dst[2] = c; }
    else for (uint32_t c = 0; c < C; ++c) dst[c] = src[c];
}

extern "C" int n2m_gather_rows(const float* x, const int64_t* idx, uint32_t K, uint32_t C, float* out, void* stream) {
    N2M_REQUIRE(x && idx && out && C >= 1, N2M_ENULL, "gather_rows: NULL tensor");
    if (K == 0) return 0;
    rows_by_index_kernel<false><<<n2m_ceil_div(K, 256), 256, 0, (hipStream_t)stream>>>(x, idx, K, C, C, C, out);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_scatter_rows(const float* src, const int64_t* idx, uint32_t K, uint32_t C, float* dst, void* stream) {
    N2M_REQUIRE(src && idx && dst && C >= 1, N2M_ENULL, "scatter_rows: NULL tensor");
    if (K == 0) return 0;
    rows_by_index_kernel<true><<<n2m_ceil_div(K, 256), 256, 0, (hipStream_t)stream>>>(src, idx, K, C, C, C, dst);
    N2M_CHECK_LAUNCH();
    return 0;
}

// the same with row strides: C leading floats of rows that are `x_stride` / `out_stride` floats apart (the RGB of an RGBA image, say)
extern "C" int n2m_gather_rows_strided(const float* x, const int64_t* idx, uint32_t K, uint32_t C, uint32_t x_stride, float* out, uint32_t out_stride,
                                       void* stream) {
    N2M_REQUIRE(x && idx && out && C >= 1, N2M_ENULL, "gather_rows_strided: NULL tensor");
    N2M_REQUIRE(x_stride >= C && out_stride >= C, N2M_EINVAL, "gather_rows_strided: a row stride below the row length");
    if (K == 0) return 0;
    rows_by_index_kernel<false><<<n2m_ceil_div(K, 256), 256, 0, (hipStream_t)stream>>>(x, idx, K, C, x_stride, out_stride, out);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_scatter_rows_strided(const float* src, const int64_t* idx, uint32_t K, uint32_t C, uint32_t src_stride, float* dst, uint32_t dst_stride,
                                        void* stream) {
    N2M_REQUIRE(src && idx && dst && C >= 1, N2M_ENULL, "scatter_rows_strided: NULL tensor");
    N2M_REQUIRE(src_stride >= C && dst_stride >= C, N2M_EINVAL, "scatter_rows_strided: a row stride below the row length");
    if (K == 0) return 0;
    rows_by_index_kernel<true><<<n2m_ceil_div(K, 256), 256, 0, (hipStream_t)stream>>>(src, idx, K, C, src_stride, dst_stride, dst);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_stage1_head(const float* aa_alpha, const float* aa_rgb, const float* rast, uint32_t h0, uint32_t w0, uint32_t ssaa,
                               const float* gt_rgba, const float* bg, float bg_scalar, float lambda_rgb, float lambda_mask, float* image,
                               float* depth, float* weights_sum, float* trig_id, float* loss_px, float* d_alpha, float* d_rgb, float* partial,
                               float* tri_err, float* tri_cnt, int packed_rgba, const float* seed, float* d_copy, void* stream) {
    N2M_REQUIRE(aa_alpha && aa_rgb && rast && gt_rgba && image && depth && weights_sum && trig_id && loss_px && partial, N2M_ENULL,
                "stage1_head: NULL tensor");
    N2M_REQUIRE((d_alpha == nullptr) == (d_rgb == nullptr), N2M_ENULL, "stage1_head: d_alpha and d_rgb come together");
    N2M_REQUIRE((tri_err == nullptr) == (tri_cnt == nullptr), N2M_ENULL, "stage1_head: tri_err and tri_cnt come together");
    N2M_REQUIRE(ssaa == 1 || ssaa == 2, N2M_EUNSUPPORTED, "stage1_head: ssaa 1 or 2 (the reduction is the exact 2 x 2 mean of torch's bilinear minification)");
    N2M_REQUIRE(h0 > 0 && w0 > 0 && (uint64_t)h0 * w0 < (1ull << 31), N2M_EINVAL, "stage1_head: bad image size");
    N2M_REQUIRE(!packed_rgba || !d_rgb || (d_alpha == d_rgb + 3 && ((uintptr_t)d_rgb & 15u) == 0), N2M_EINVAL,
                "stage1_head: packed gradients are ONE 16-byte aligned [h, w, 4] image (d_alpha = d_rgb + 3)");
    N2M_REQUIRE(!d_copy || (packed_rgba && d_rgb && ((uintptr_t)d_copy & 15u) == 0), N2M_EINVAL, "stage1_head: d_copy goes with the packed gradient image");
    hipStream_t s = (hipStream_t)stream;
    const uint32_t N = h0 * w0;
    if (ssaa == 1)
        stage1_head_kernel<1><<<n2m_ceil_div(N, 256), 256, 0, s>>>(aa_alpha, aa_rgb, rast, h0, w0, gt_rgba, bg, bg_scalar, lambda_rgb, lambda_mask,
                                                                  image, depth, weights_sum, trig_id, loss_px, d_alpha, d_rgb, partial, tri_err, tri_cnt, packed_rgba ? 4u : 1u,
                                                                  packed_rgba ? 4u : 3u, seed, d_copy);
    else
        stage1_head_kernel<2><<<n2m_ceil_div(N, 256), 256, 0, s>>>(aa_alpha, aa_rgb, rast, h0, w0, gt_rgba, bg, bg_scalar, lambda_rgb, lambda_mask,
                                                                  image, depth, weights_sum, trig_id, loss_px, d_alpha, d_rgb, partial, tri_err, tri_cnt, packed_rgba ? 4u : 1u,
                                                                  packed_rgba ? 4u : 3u, seed, d_copy);
    N2M_CHECK_LAUNCH();
    return 0;
}

template <bool PEER>
__global__ void __launch_bounds__(256)
adam_kernel(AdamTensors t, float beta1, float beta2, float omb1, float omb2, float eps, const float* scale, const float* found_inf,
            const float* bias, AdamPeerK pe) {
    adam_body<PEER>(t, beta1, beta2, omb1, omb2, eps, scale, found_inf, bias, pe);
}

// n2m_adam_step_scaler: the optimizer pass whose LAST workgroup to finish also does the GradScaler / step-count / loss-value bookkeeping that
// n2m_scaler_update_slots_loss3 does as a one-workgroup launch behind it (scaler_update_slots_body: the same code walking the partials in the same
// order -- identical bits).  Every wave reads found_inf / bias before its update and leaves an arrival mark when it is done; the one wave that waits
// for all marks knows that nobody will read that state again in this launch and rewrites it for the next step.  No fence: the bookkeeping reads
// nothing the other workgroups of this launch wrote.  Saves the step one launch on its critical path (a single workgroup that
// waits for a free CU beside the marcher: ~10 us + the queue gap in front of the next lookup).
struct ScalerTailK {
    float* scale; float* growth_tracker; float* found_inf; float* steps; float* bias;
    uint32_t participants; double beta1, beta2; float growth_factor, backoff_factor, growth_interval;
    const float* loss_partial; uint32_t n_partial; float inv_rays; float* loss; float* loss_sum;
    const float* extra_partial; uint32_t n_extra; float extra_scale;
    const float* extra2_partial; uint32_t n_extra2; float extra2_scale;
    uint32_t* ticket;
};
template <bool ONE_WAVE>
__device__ __forceinline__ void scaler_update_slots_body(float* scale, float* growth_tracker, float* found_inf, float* steps, float* bias,
                                                         uint32_t participants, double beta1, double beta2, float growth_factor, float backoff_factor,
                                                         float growth_interval, const float* __restrict__ loss_partial, uint32_t n_partial, float inv_rays,
                                                         float* __restrict__ loss, float* __restrict__ loss_sum, const float* __restrict__ extra_partial,
                                                         uint32_t n_extra, float extra_scale, const float* __restrict__ extra2_partial, uint32_t n_extra2,
                                                         float extra2_scale);

constexpr uint32_t kTailSlots = 64;        // arrival counters, one 128-byte line each (N2M_TAIL_TICKET_WORDS = 32 x this; 1 024 of them: no faster)
// (the kernel's argument block as the hardware lays it out: where `tail` sits in the kernarg segment)
struct AdamTailArgs {
    AdamTensors t; float beta1, beta2, omb1, omb2, eps; const float* scale; const float* found_inf; const float* bias; ScalerTailK tail;
};

__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(48)))      // (the bookkeeping's scalars spill; the update keeps the pass's occupancy)
adam_kernel_with_scaler(AdamTensors t, float beta1, float beta2, float omb1, float omb2, float eps, const float* scale, const float* found_inf,
                        const float* bias, ScalerTailK tail) {
    adam_body<false>(t, beta1, beta2, omb1, omb2, eps, scale, found_inf, bias, AdamPeerK{});
    // Arrival without a round trip: every wave but ONE leaves a fire-and-forget add on counter (workgroup % 64) -- its own 128-byte line -- behind
    // its reads of found_inf / bias / scale (consumed long before: the update depends on them), and is gone.  Wave 0 of the workgroup with the highest
    // index does its own rows, then waits until the counters hold the other 4 x gridDim - 1 arrivals: that workgroup is dispatched last and nobody
    // waits for it, so the wait blocks no one.  Then it does the bookkeeping alone (scaler_update_slots_body<true>).  No LDS and no barrier anywhere in
    // this kernel; SGPRs capped at the update's own need (amdgpu_num_sgpr: the bookkeeping's scalars spill); the bookkeeping's arguments are read
    // from the kernarg segment through a laundered offset inside that one wave's branch, not at the top of every wave.
    // MEASURED, NOT ADOPTED (the step executor keeps the separate launch, N2M_ADAM_TAIL=1 selects this kernel).  The pass alone takes
    // 93 us.  In one kernel with the bookkeeping: 182 us with a RETURNING ticket per workgroup + barrier on one counter (18 000 same-address round
    // trips), 136 us on 64 counters; 135 us with this arrival scheme while the bookkeeping used LDS / __syncthreads -- and still 135 us with the update
    // ALONE followed by an immediate return, as long as the rest of the code was in the kernel: it raised the kernel's SGPR count from 45 to 100 and
    // with it the pass lost waves per SIMD.  Capped at 48 SGPRs: 116 us (121 us with 1 024 counters).  The 22 us that remain are more than the 10 us +
    // queue gap the separate one-workgroup launch costs: step 0.543 -> 0.548 ms.
    if (blockIdx.x != gridDim.x - 1u || threadIdx.x >= 64u) {
        if ((threadIdx.x & 63u) == 0u)
            (void)__hip_atomic_fetch_add(tail.ticket + 32u * (blockIdx.x & (kTailSlots - 1u)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    uint32_t off = (uint32_t)offsetof(AdamTailArgs, tail);
    asm volatile("" : "+s"(off));                          // (not a constant any more: the loads below stay inside this branch)
    const ScalerTailK* tp = reinterpret_cast<const ScalerTailK*>((const char*)__builtin_amdgcn_kernarg_segment_ptr() + off);
    {
        const uint32_t expected = 4u * gridDim.x - 1u;
        uint32_t* mine = tp->ticket + 32u * threadIdx.x;
        for (;;) {
            uint32_t v = 0u;
#pragma unroll
            for (uint32_t k = 0; k < kTailSlots / 64u; ++k) v += __hip_atomic_load(mine + 32u * 64u * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (n2m_wave_sum_u32(v) >= expected) break;
            __builtin_amdgcn_s_sleep(8);
        }
#pragma unroll
        for (uint32_t k = 0; k < kTailSlots / 64u; ++k) __hip_atomic_store(mine + 32u * 64u * k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    scaler_update_slots_body<true>(tp->scale, tp->growth_tracker, tp->found_inf, tp->steps, tp->bias, tp->participants, tp->beta1, tp->beta2,
                                   tp->growth_factor, tp->backoff_factor, tp->growth_interval, tp->loss_partial, tp->n_partial, tp->inv_rays,
                                   tp->loss, tp->loss_sum, tp->extra_partial, tp->n_extra, tp->extra_scale, tp->extra2_partial, tp->n_extra2,
                                   tp->extra2_scale);
}

static int adam_step_impl(const N2mAdamDesc* d, double beta1, double beta2, float eps, const float* scale, const float* found_inf,
                          const float* bias, const N2mAdamPeer* peer, void* stream, const N2mScalerTail* tail = nullptr) {
    N2M_REQUIRE(d != nullptr && bias != nullptr, N2M_ENULL, "adam_step: NULL descriptor / bias");
    N2M_REQUIRE(d->count >= 1 && d->count <= N2M_ADAM_MAX, N2M_EINVAL, "adam_step: 1..%d tensors per call (got %u)", N2M_ADAM_MAX, d->count);
    AdamTensors t;
    memset(&t, 0, sizeof(t));
    uint32_t blocks = 0;
    for (uint32_t k = 0; k < d->count; ++k) {
        N2M_REQUIRE(d->param[k] && d->grad[k] && d->exp_avg[k] && d->exp_avg_sq[k], N2M_ENULL, "adam_step: NULL tensor %u", k);
        // the kernel moves float4 / half4 vectors; gfx950 global accesses need element (dword / half-pair) alignment only, so views at
        // any element offset of a flat gradient buffer are fine (tests/test_optim.py drives a 12-byte-offset view) -- but not less
        N2M_REQUIRE((((uintptr_t)d->param[k] | (uintptr_t)d->grad[k] | (uintptr_t)d->exp_avg[k] | (uintptr_t)d->exp_avg_sq[k]) & 3u) == 0,
                    N2M_EINVAL, "adam_step: tensor %u is not 4-byte aligned (param/grad/exp_avg/exp_avg_sq)", k);
        t.p[k] = (uint64_t)d->param[k]; t.g[k] = (uint64_t)d->grad[k]; t.m[k] = (uint64_t)d->exp_avg[k]; t.v[k] = (uint64_t)d->exp_avg_sq[k];
        t.shadow[k] = (uint64_t)d->half_shadow[k];
        t.n[k] = d->numel[k];
        t.lr[k] = d->lr[k];
        t.first_block[k] = blocks;
        blocks += n2m_ceil_div(d->numel[k], 1024);
        if (d->grad_is_half[k]) t.g_half_mask |= 1u << k;
        if (d->clear_grad[k] && !d->grad_is_half[k]) t.clear_mask |= 1u << k;
        t.shadow_mode[k] = (uint8_t)(d->half_shadow[k] ? (d->shadow_mode[k] ? d->shadow_mode[k] : 1) : 0);
        N2M_REQUIRE(d->slot[k] >= 0 && d->slot[k] <= N2M_ADAM_MAX, N2M_EINVAL, "adam_step: slot of tensor %u out of range", k);
        t.slot[k] = (uint8_t)d->slot[k];
        N2M_REQUIRE(t.shadow_mode[k] <= 3 && !(t.shadow_mode[k] == 3 && (d->numel[k] & 1u)), N2M_EINVAL, "adam_step: bad shadow mode for tensor %u", k);
    }
    // a [rows,1] tensor (mode 2) and a [rows,2] tensor (mode 3) writing columns of the SAME packed table are updated together by the
    // threads of the second one: complete 8-byte rows, coalesced.  The first tensor then gets no blocks of its own.
    for (uint32_t k = 0; k < d->count; ++k) t.partner[k] = -1;
    for (uint32_t k = 0; k < d->count; ++k) {
        if (t.shadow_mode[k] != 3) continue;
        for (uint32_t j = 0; j < d->count; ++j)
            if (t.shadow_mode[j] == 2 && d->half_shadow[j] == d->half_shadow[k] && d->numel[k] == 2 * d->numel[j] && ((uintptr_t)d->half_shadow[k] & 15u) == 0) {
                // the paired update neither clears the partner's gradient nor zeroes it on a skipped step: a consume-and-clear gradient
                // buffer for that tensor would silently keep its values
                N2M_REQUIRE(!((t.clear_mask >> j) & 1u), N2M_EUNSUPPORTED,
                            "adam_step: tensor %u shares a packed table with tensor %u and cannot use clear_grad", j, k);
                t.partner[k] = (int8_t)j;
                break;
            }
    }
    blocks = 0;
    for (uint32_t k = 0; k < d->count; ++k) {
        bool is_partner = false;
        for (uint32_t j = 0; j < d->count; ++j) is_partner |= t.partner[j] == (int8_t)k;
        t.first_block[k] = blocks;
        if (!is_partner) blocks += n2m_ceil_div(d->numel[k], 1024);
    }
    t.first_block[d->count] = blocks;
    t.count = d->count;
    if (blocks == 0) return 0;
    {   // algorithmic bytes: param + both moments read and written (24 B), gradient read (4 or 2 B), working copy written (2 or 4 B / value)
        double bytes = 0;
        for (uint32_t k = 0; k < d->count; ++k)
            bytes += (double)d->numel[k] * (24.0 + (d->grad_is_half[k] ? 2.0 : 4.0) + (t.shadow_mode[k] == 1 ? 2.0 : (t.shadow_mode[k] == 2 ? 4.0 : (t.shadow_mode[k] == 3 ? 2.0 : 0.0))));
        AdamPeerK pe;
        memset(&pe, 0, sizeof(pe));
        for (uint32_t k = 0; k < N2M_ADAM_MAX; ++k) pe.entry[k] = -1;
        if (peer) {
            N2M_REQUIRE(peer->world >= 1 && peer->world <= N2M_PEER_MAX && peer->n_remote < N2M_PEER_MAX, N2M_EINVAL, "adam_step_peer: 1..%d ranks", N2M_PEER_MAX);
            N2M_REQUIRE(peer->n_remote == 0 || peer->packed_local != nullptr, N2M_ENULL, "adam_step_peer: remote packed tables need the local base");
            pe.world = peer->world;
            pe.n_remote = peer->n_remote;
            for (uint32_t r = 0; r < peer->n_remote; ++r) {
                N2M_REQUIRE(peer->packed_remote[r] != nullptr && (((uintptr_t)peer->packed_remote[r] ^ (uintptr_t)peer->packed_local) & 15u) == 0, N2M_EINVAL,
                            "adam_step_peer: remote packed table %u NULL or aligned differently from the local one", r);
                pe.remote_delta[r] = (long long)((intptr_t)peer->packed_remote[r] - (intptr_t)peer->packed_local);
            }
            uint32_t rows_used = 0;
            for (uint32_t k = 0; k < d->count; ++k) {
                if (peer->slots[k][0] == nullptr) continue;
                N2M_REQUIRE(rows_used < 4u, N2M_EUNSUPPORTED, "adam_step_peer: at most four tensors take their gradient from slots");
                // the fused form covers what the sharded tables are at an even split: whole float4 / half4 vectors, and the paired packed-row
                // store (a [rows,1] + a [rows,2] tensor on one 16-byte aligned packed slice); anything else keeps the separate passes
                N2M_REQUIRE(d->numel[k] % 4u == 0u && !((t.clear_mask >> k) & 1u), N2M_EUNSUPPORTED,
                            "adam_step_peer: tensor %u: element count not a multiple of 4, or clear_grad set", k);
                for (uint32_t sl = 0; sl < peer->world; ++sl) {
                    N2M_REQUIRE(peer->slots[k][sl] != nullptr && ((uintptr_t)peer->slots[k][sl] & 3u) == 0, N2M_EINVAL, "adam_step_peer: slot %u of tensor %u", sl, k);
                    pe.slots[rows_used][sl] = (uint64_t)peer->slots[k][sl];
                }
                pe.entry[k] = (int8_t)rows_used++;
            }
            for (uint32_t k = 0; k < d->count; ++k) {
                if (t.shadow_mode[k] == 2 || t.shadow_mode[k] == 3) {
                    bool paired = t.partner[k] >= 0;
                    for (uint32_t j = 0; j < d->count; ++j) paired |= t.partner[j] == (int8_t)k;
                    N2M_REQUIRE(peer->n_remote == 0 || paired, N2M_EUNSUPPORTED,
                                "adam_step_peer: tensor %u refreshes a packed table outside the paired 16-byte form (odd or unaligned slice)", k);
                    if (t.partner[k] >= 0) N2M_REQUIRE(d->numel[k] == 2u * d->numel[t.partner[k]], N2M_EUNSUPPORTED, "adam_step_peer: tensor %u: unequal row counts", k);
                }
            }
        }
        N2M_PROF_K(N2M_K_ADAM, (hipStream_t)stream, bytes);
        if (tail) {
            N2M_REQUIRE(!peer, N2M_EUNSUPPORTED, "adam_step_scaler: not with the peer-store form");
            N2M_REQUIRE(tail->ticket && found_inf && tail->steps && tail->loss_partial && tail->n_rays > 0, N2M_ENULL,
                        "adam_step_scaler: NULL ticket / found_inf / steps / loss partials (or no rays)");
            const ScalerTailK tk{const_cast<float*>(scale), tail->growth_tracker, const_cast<float*>(found_inf), tail->steps, const_cast<float*>(bias),
                                 tail->participants, beta1, beta2, tail->growth_factor, tail->backoff_factor, tail->growth_interval, tail->loss_partial,
                                 tail->n_partial, 1.0f / (float)tail->n_rays, tail->loss, tail->loss_sum, tail->extra_partial,
                                 tail->extra_partial ? tail->n_extra : 0u, tail->extra_scale, tail->extra2_partial,
                                 tail->extra2_partial ? tail->n_extra2 : 0u, tail->extra2_scale, tail->ticket};
            N2M_LAUNCH(adam_kernel_with_scaler, blocks, 256, 0, (hipStream_t)stream, t, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), eps,
                       scale, found_inf, bias, tk);
        } else if (peer)
            N2M_LAUNCH((adam_kernel<true>), blocks, 256, 0, (hipStream_t)stream, t, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), eps, scale,
                                                                        found_inf, bias, pe);
        else
            N2M_LAUNCH((adam_kernel<false>), blocks, 256, 0, (hipStream_t)stream, t, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), eps, scale,
                                                                         found_inf, bias, pe);
    }
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_adam_step(const N2mAdamDesc* d, double beta1, double beta2, float eps, const float* scale, const float* found_inf,
                             const float* bias, void* stream) {
    return adam_step_impl(d, beta1, beta2, eps, scale, found_inf, bias, nullptr, stream);
}

extern "C" int n2m_adam_step_scaler(const N2mAdamDesc* d, double beta1, double beta2, float eps, float* scale, float* found_inf, float* bias,
                                    const N2mScalerTail* tail, void* stream) {
    N2M_REQUIRE(tail != nullptr, N2M_ENULL, "adam_step_scaler: NULL tail (use n2m_adam_step + n2m_scaler_update_slots_loss3)");
    return adam_step_impl(d, beta1, beta2, eps, scale, found_inf, bias, nullptr, stream, tail);
}

extern "C" int n2m_adam_step_peer(const N2mAdamDesc* d, double beta1, double beta2, float eps, const float* scale, const float* found_inf,
                                  const float* bias, const N2mAdamPeer* peer, void* stream) {
    N2M_REQUIRE(peer != nullptr, N2M_ENULL, "adam_step_peer: NULL peer description (use n2m_adam_step)");
    return adam_step_impl(d, beta1, beta2, eps, scale, found_inf, bias, peer, stream);
}

// The same with one step count PER TENSOR SLOT (torch.optim.Adam keeps `state[p]["step"]` per parameter: a parameter that gets its
// first gradient late -- nerf2mesh's specular head after `diffuse_step` -- starts its bias corrections at t = 1).  Thread 0 does the
// scaler bookkeeping and the global count (slot 0), thread s the count and the next-step corrections of slot s.
// (one body for the stand-alone launch below -- a workgroup of `blockDim.x` threads, partial sums through LDS -- and for ONE WAVE at the end of
//  adam_kernel_with_scaler above, ONE_WAVE = true: no LDS and no barrier in that kernel -- a kernel that has either gets its workgroups placed as
//  units, which costs the optimizer pass of 18 000 short-lived workgroups 40 us, measured -- the wave walks the partials in the order a
//  256-thread workgroup does: virtual wave w, lane l takes partials 64 w + l, + 256, ...; the same wave sums, added in wave order: identical bits)
template <bool ONE_WAVE>
__device__ __forceinline__ void scaler_update_slots_body(float* scale, float* growth_tracker, float* found_inf, float* steps /*[1+MAX]*/,
                                                         float* bias /*[1+MAX][2]*/, uint32_t participants, double beta1, double beta2,
                                                         float growth_factor, float backoff_factor, float growth_interval,
                                                         const float* __restrict__ loss_partial, uint32_t n_partial, float inv_rays,
                                                         float* __restrict__ loss, float* __restrict__ loss_sum,
                                                         const float* __restrict__ extra_partial, uint32_t n_extra, float extra_scale,
                                                         const float* __restrict__ extra2_partial, uint32_t n_extra2, float extra2_scale) {
    const uint32_t s = ONE_WAVE ? (threadIdx.x & 63u) : threadIdx.x;
    float tot_part = 0.0f, tot_extra = 0.0f, tot_extra2 = 0.0f;       // thread 0: the three sums
    bool ok;
    if constexpr (ONE_WAVE) {
        // (the 32 values of a chunk of 2 048 partials are requested together and added in order afterwards: one memory round trip per chunk -- the
        //  plain loop's one round trip per value made this tail 45 us long)
        auto sum256 = [&](const float* __restrict__ part, uint32_t n) {
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            for (uint32_t base = 0; base < n; base += 2048u) {
                float x[8][4];
#pragma unroll
                for (uint32_t k = 0; k < 8u; ++k)
#pragma unroll
                    for (uint32_t w = 0; w < 4u; ++w) {
                        const uint32_t i = base + 256u * k + 64u * w + s;
                        x[k][w] = i < n ? part[i] : 0.0f;
                    }
#pragma unroll
                for (uint32_t k = 0; k < 8u; ++k)
#pragma unroll
                    for (uint32_t w = 0; w < 4u; ++w)
                        if (base + 256u * k + 64u * w + s < n) acc[w] += x[k][w];
            }
            float tot = 0.0f;
#pragma unroll
            for (uint32_t w = 0; w < 4u; ++w) tot += n2m_wave_sum(acc[w]);
            return tot;
        };
        if (extra2_partial) tot_extra2 = sum256(extra2_partial, n_extra2);
        if (extra_partial) tot_extra = sum256(extra_partial, n_extra);
        if (loss_partial) tot_part = sum256(loss_partial, n_partial);
        ok = *found_inf == 0.0f;                          // (one wave: every lane has its verdict before lane 0's store below is issued)
    } else {
        __shared__ float wave_part[16], wave_extra[16], wave_extra2[16];
        if (extra2_partial) {     // a third term (SDF recipe: the eikonal loss, lambda / M x sum of (|normal| - 1)^2)
            float acc = 0.0f;
            for (uint32_t i = s; i < n_extra2; i += blockDim.x) acc += extra2_partial[i];
            acc = n2m_wave_sum(acc);
            if ((s & 63u) == 0u) wave_extra2[s >> 6] = acc;
        }
        if (extra_partial) {      // a second term of the loss value with its own normalisation (specular regulariser: lambda / M x sum of squares)
            float acc = 0.0f;
            for (uint32_t i = s; i < n_extra; i += blockDim.x) acc += extra_partial[i];
            acc = n2m_wave_sum(acc);
            if ((s & 63u) == 0u) wave_extra[s >> 6] = acc;
        }
        if (loss_partial) {       // the step's loss VALUE (nobody on the GPU waits for it): per-workgroup partials of n2m_composite_loss_train,
            float acc = 0.0f;     // summed in a fixed order: thread j takes partials j, j + blockDim, ...; lanes in scan order; waves 0, 1, ...
            for (uint32_t i = s; i < n_partial; i += blockDim.x) acc += loss_partial[i];      // (one wave did this alone: 17 serial round trips, 8 us)
            acc = n2m_wave_sum(acc);
            if ((s & 63u) == 0u) wave_part[s >> 6] = acc;
        }
        ok = *found_inf == 0.0f;
        __syncthreads();                                  // everyone has read the verdict before thread 0 clears it
        if (s == 0) {
            for (uint32_t w = 0; w < (blockDim.x + 63u) / 64u; ++w) {
                if (loss_partial) tot_part += wave_part[w];
                if (extra_partial) tot_extra += wave_extra[w];
                if (extra2_partial) tot_extra2 += wave_extra2[w];
            }
        }
    }
    if (loss_partial && s == 0) {
        float v = tot_part * inv_rays;
        if (extra_partial) v += tot_extra * extra_scale;
        if (extra2_partial) v += tot_extra2 * extra2_scale;
        if (loss) *loss = v;
        if (loss_sum) *loss_sum += v;
    }
    if (s == 0) {
        if (!ok) {
            if (scale) *scale *= backoff_factor;
            if (growth_tracker) *growth_tracker = 0.0f;
        } else if (growth_tracker) {
            const float g = *growth_tracker + 1.0f;
            if (g >= growth_interval) {
                const float ns = *scale * growth_factor;
                if (scale && ns <= 3.0e38f) *scale = ns;            // do not grow into inf
                *growth_tracker = 0.0f;
            } else {
                *growth_tracker = g;
            }
        }
        *found_inf = 0.0f;
    }
    if (s <= N2M_ADAM_MAX) {
        if (ok && (s == 0 || ((participants >> (s - 1u)) & 1u))) steps[s] += 1.0f;
        const double t = (double)steps[s] + 1.0;      // corrections of this slot's NEXT step, in double like torch's host arithmetic
        bias[2u * s] = (float)(1.0 - pow(beta1, t));
        bias[2u * s + 1u] = (float)sqrt(1.0 - pow(beta2, t));
    }
}

__global__ void scaler_update_slots_kernel(float* scale, float* growth_tracker, float* found_inf, float* steps /*[1+MAX]*/,
                                           float* bias /*[1+MAX][2]*/, uint32_t participants, double beta1, double beta2,
                                           float growth_factor, float backoff_factor, float growth_interval,
                                           const float* __restrict__ loss_partial, uint32_t n_partial, float inv_rays,
                                           float* __restrict__ loss, float* __restrict__ loss_sum,
                                           const float* __restrict__ extra_partial, uint32_t n_extra, float extra_scale,
                                           const float* __restrict__ extra2_partial = nullptr, uint32_t n_extra2 = 0, float extra2_scale = 0.0f) {
    __builtin_amdgcn_s_setprio(3);
    scaler_update_slots_body<false>(scale, growth_tracker, found_inf, steps, bias, participants, beta1, beta2, growth_factor, backoff_factor, growth_interval,
                             loss_partial, n_partial, inv_rays, loss, loss_sum, extra_partial, n_extra, extra_scale, extra2_partial, n_extra2, extra2_scale);
}

extern "C" int n2m_scaler_update_slots(float* scale, float* growth_tracker, float* found_inf, float* steps, float* bias,
                                       uint32_t participants, double beta1, double beta2, float growth_factor, float backoff_factor,
                                       float growth_interval, void* stream) {
    N2M_REQUIRE(found_inf != nullptr && steps != nullptr && bias != nullptr, N2M_ENULL, "scaler_update_slots: NULL found_inf / steps / bias");
    scaler_update_slots_kernel<<<1, 64, 0, (hipStream_t)stream>>>(scale, growth_tracker, found_inf, steps, bias, participants, beta1, beta2,
                                                                  growth_factor, backoff_factor, growth_interval, nullptr, 0u, 0.0f, nullptr, nullptr, nullptr, 0u, 0.0f);
    N2M_CHECK_LAUNCH();
    return 0;
}

// The same + the final reduction of the loss value from the per-workgroup partials n2m_composite_loss_train leaves when it is given no
// ticket (loss = sum(partial[0..n_partial)) / n_rays; *loss_sum += loss): keeps the arrival ticket -- one __threadfence + one same-address
// atomic per workgroup -- out of the compositing kernel, whose loss VALUE nothing on the device waits for.
extern "C" int n2m_scaler_update_slots_loss3(float* scale, float* growth_tracker, float* found_inf, float* steps, float* bias,
                                            uint32_t participants, double beta1, double beta2, float growth_factor, float backoff_factor,
                                            float growth_interval, const float* loss_partial, uint32_t n_partial, uint32_t n_rays, float* loss,
                                            float* loss_sum, const float* extra_partial, uint32_t n_extra, float extra_scale, const float* extra2_partial,
                                             uint32_t n_extra2, float extra2_scale, void* stream) {
    N2M_REQUIRE(found_inf != nullptr && steps != nullptr && bias != nullptr, N2M_ENULL, "scaler_update_slots: NULL found_inf / steps / bias");
    N2M_REQUIRE(loss_partial != nullptr && n_rays > 0, N2M_EINVAL, "scaler_update_slots_loss: needs the loss partials and the ray count");
    // 256 threads, not 1024: the launch sits on the main stream between Adam and the next forward while the side stream's marcher fills
    // every CU, and a 16-wave workgroup then waits (measured: up to 27 us) for one CU to free 16 wave slots at once
    scaler_update_slots_kernel<<<1, 256, 0, (hipStream_t)stream>>>(scale, growth_tracker, found_inf, steps, bias, participants, beta1, beta2,
                                                                  growth_factor, backoff_factor, growth_interval, loss_partial, n_partial,
                                                                  1.0f / (float)n_rays, loss, loss_sum, extra_partial, extra_partial ? n_extra : 0u, extra_scale,
                                                                  extra2_partial, extra2_partial ? n_extra2 : 0u, extra2_scale);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_scaler_update_slots_loss2(float* scale, float* growth_tracker, float* found_inf, float* steps, float* bias,
                                             uint32_t participants, double beta1, double beta2, float growth_factor, float backoff_factor,
                                             float growth_interval, const float* loss_partial, uint32_t n_partial, uint32_t n_rays, float* loss,
                                             float* loss_sum, const float* extra_partial, uint32_t n_extra, float extra_scale, void* stream) {
    return n2m_scaler_update_slots_loss3(scale, growth_tracker, found_inf, steps, bias, participants, beta1, beta2, growth_factor, backoff_factor,
                                         growth_interval, loss_partial, n_partial, n_rays, loss, loss_sum, extra_partial, n_extra, extra_scale,
                                         nullptr, 0u, 0.0f, stream);
}

extern "C" int n2m_scaler_update_slots_loss(float* scale, float* growth_tracker, float* found_inf, float* steps, float* bias,
                                            uint32_t participants, double beta1, double beta2, float growth_factor, float backoff_factor,
                                            float growth_interval, const float* loss_partial, uint32_t n_partial, uint32_t n_rays, float* loss,
                                            float* loss_sum, void* stream) {
    return n2m_scaler_update_slots_loss2(scale, growth_tracker, found_inf, steps, bias, participants, beta1, beta2, growth_factor, backoff_factor,
                                         growth_interval, loss_partial, n_partial, n_rays, loss, loss_sum, nullptr, 0u, 0.0f, stream);
}

extern "C" int n2m_scaler_update(float* scale, float* growth_tracker, float* found_inf, float* step, float* bias, double beta1,
                                 double beta2, float growth_factor, float backoff_factor, float growth_interval, void* stream) {
    N2M_REQUIRE(found_inf != nullptr, N2M_ENULL, "scaler_update: found_inf is NULL");
    scaler_update_kernel<<<1, 1, 0, (hipStream_t)stream>>>(scale, growth_tracker, found_inf, step, bias, beta1, beta2, growth_factor,
                                                           backoff_factor, growth_interval);
    N2M_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------ EMA of the parameters
// torch_ema.ExponentialMovingAverage.update() (un-vendored dependency of the reference; nerf/utils.py:544-545 builds it with decay 0.95,
// :1213-1214 updates it once per epoch) for up to N2M_EMA_MAX tensors in ONE launch: shadow -= (1 - decay) * (shadow - param), in exactly
// that association (the library forms tmp = shadow - param, scales it in place and subtracts; no FMA: -ffp-contract=off), so the shadow is
// bit-identical to what the torch ops produce.  16 bytes per lane, streaming accesses: a once-per-epoch pass over 73 MB of parameters.
struct EmaTensors {
    uint64_t shadow[N2M_EMA_MAX], param[N2M_EMA_MAX];
    uint32_t n[N2M_EMA_MAX], first_block[N2M_EMA_MAX + 1];
    uint32_t count;
};

__global__ void __launch_bounds__(256) ema_update_kernel(EmaTensors t, float one_minus_decay) {
    uint32_t k = 0;
    while (k + 1 < t.count && blockIdx.x >= t.first_block[k + 1]) ++k;
    const uint32_t i0 = ((blockIdx.x - t.first_block[k]) * 256u + threadIdx.x) * 4u;
    const uint32_t n = t.n[k];
    if (i0 >= n) return;
    float* __restrict__ S = reinterpret_cast<float*>(t.shadow[k]);
    const float* __restrict__ P = reinterpret_cast<const float*>(t.param[k]);
    typedef float f4v __attribute__((ext_vector_type(4)));
    if (i0 + 4u <= n && (((t.shadow[k] | t.param[k]) & 15u) == 0)) {
        f4v s = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(S + i0));
        const f4v p = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(P + i0));
        s.x = s.x - (s.x - p.x) * one_minus_decay; s.y = s.y - (s.y - p.y) * one_minus_decay;
        s.z = s.z - (s.z - p.z) * one_minus_decay; s.w = s.w - (s.w - p.w) * one_minus_decay;
        __builtin_nontemporal_store(s, reinterpret_cast<f4v*>(S + i0));
    } else {
        for (uint32_t e = 0; e < 4u && i0 + e < n; ++e) S[i0 + e] = S[i0 + e] - (S[i0 + e] - P[i0 + e]) * one_minus_decay;
    }
}

extern "C" int n2m_ema_update(const N2mEmaDesc* d, float one_minus_decay, void* stream) {
    N2M_REQUIRE(d != nullptr, N2M_ENULL, "ema_update: NULL descriptor");
    N2M_REQUIRE(d->count >= 1 && d->count <= N2M_EMA_MAX, N2M_EINVAL, "ema_update: 1..%d tensors per call (got %u)", N2M_EMA_MAX, d->count);
    N2M_REQUIRE(one_minus_decay >= 0.0f && one_minus_decay <= 1.0f, N2M_EINVAL, "ema_update: 1 - decay = %g outside [0, 1]", (double)one_minus_decay);
    EmaTensors t;
    t.count = d->count;
    uint32_t blocks = 0;
    for (uint32_t k = 0; k < d->count; ++k) {
        N2M_REQUIRE(d->shadow[k] && d->param[k], N2M_ENULL, "ema_update: NULL tensor %u", k);
        N2M_REQUIRE((((uintptr_t)d->shadow[k] | (uintptr_t)d->param[k]) & 3u) == 0, N2M_EINVAL, "ema_update: tensor %u is not 4-byte aligned", k);
        N2M_REQUIRE(d->shadow[k] != d->param[k], N2M_EINVAL, "ema_update: tensor %u: shadow and parameter are the same memory", k);
        t.shadow[k] = (uint64_t)(uintptr_t)d->shadow[k];
        t.param[k] = (uint64_t)(uintptr_t)d->param[k];
        t.n[k] = d->numel[k];
        t.first_block[k] = blocks;
        blocks += (d->numel[k] + 1023u) / 1024u;
    }
    t.first_block[d->count] = blocks;
    if (blocks == 0) return 0;
    ema_update_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(t, one_minus_decay);
    N2M_CHECK_LAUNCH();
    return 0;
}
