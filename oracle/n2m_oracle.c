/*
 * n2m_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference algorithm for nerf2mesh's
 * hot path (ray marching, compositing, hash-grid encoding, SH encoding).
 *
 * Who may use this: tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg -- as the CHECKER /
 * the timed CPU baseline, never as the product.  The product path (nerf2mesh_amd/) does not import, link or
 * fall back to anything in oracle/.
 *
 * Pinning: every function here is checked bit-for-bit (integer outputs and, for the fp32 paths, float
 * outputs too) against the reference's own kernels compiled for the host (oracle/_ref, see build_ref.py) in
 * tests/test_oracle_vs_reference.py, and against the committed fixtures in tests/golden/ that were generated
 * from those reference builds (tests/golden/make_golden.py).
 *
 * Canonical arithmetic (decisions the CUDA build leaves to nvcc): no FMA contraction
 * (compile with -ffp-contract=off), 1/x is an IEEE division, expf instead of __expf, rsqrt = 1/sqrtf.
 * All "file:line" citations are relative to the reference checkout.
 *
 * Parallelism: OpenMP over rays / samples / levels where the result does not depend on the order.
 * Scatter-adds are parallel over LEVELS only (levels own disjoint table ranges), serial inside a level, so
 * sums are reproducible and equal to a serial execution of the reference.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define N2M_MAX_LEVELS 32
#define N2M_MAX_D 5
#define N2M_MAX_C 8

/* ---------------------------------------------------------------------------------------------- helpers */

static inline float clampf(float v, float lo, float hi) { return fminf(hi, fmaxf(lo, v)); } /* raymarching.cu:34-36 */
static inline float sign1(float v) { return copysignf(1.0f, v); }                           /* raymarching.cu:30-32 */

/* 10 bits per axis -> 30-bit interleave, x in bit 0 (raymarching.cu:56-71). */
static inline uint32_t spread3(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
static inline uint32_t morton_encode(uint32_t x, uint32_t y, uint32_t z) {
    return spread3(x) | (spread3(y) << 1) | (spread3(z) << 2);
}
/* inverse of spread3 on every third bit (raymarching.cu:73-81) */
static inline uint32_t gather3(uint32_t v) {
    v &= 0x49249249u;
    v = (v | (v >> 2)) & 0xC30C30C3u;
    v = (v | (v >> 4)) & 0x0F00F00Fu;
    v = (v | (v >> 8)) & 0xFF0000FFu;
    v = (v | (v >> 16)) & 0x0000FFFFu;
    return v;
}

/* cascade level from the position / from the step size (raymarching.cu:42-54):
 * exponent e of frexpf (|v| in [2^(e-1), 2^e)), clamped to [0, C-1]; computed in float like the reference. */
static inline int level_from_exponent(float mx, float cascades) {
    int e;
    (void)frexpf(mx, &e);
    return (int)fminf(cascades - 1.0f, fmaxf(0.0f, (float)e));
}
static inline int level_from_pos(float x, float y, float z, float cascades) {
    return level_from_exponent(fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z))), cascades);
}
static inline int level_from_dt(float dt, float H, float cascades) {
    /* `dt * H * 0.5`: float product, then a double multiply by the literal, narrowed to float (:50) */
    const float mx = (float)((double)(dt * H) * 0.5);
    return level_from_exponent(mx, cascades);
}

/* IEEE binary16 <-> binary32, round-to-nearest-even; what at::Half's converting constructor does. */
static inline float h2f(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else { /* subnormal: renormalise */
            int s = 0;
            while (!(man & 0x400u)) { man <<= 1; ++s; }
            man &= 0x3FFu;
            bits = sign | ((uint32_t)(113 - s) << 23) | (man << 13);
        }
    } else if (exp == 31) bits = sign | 0x7F800000u | (man << 13);
    else bits = sign | ((exp + 112u) << 23) | (man << 13);
    float f; memcpy(&f, &bits, 4); return f;
}
static inline uint16_t f2h(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    x &= 0x7FFFFFFFu;
    if (x >= 0x7F800000u) return (uint16_t)(sign | (x > 0x7F800000u ? 0x7E00u : 0x7C00u));
    if (x >= 0x477FF000u) { /* rounds to >= 65520 -> inf */
        return (uint16_t)(sign | 0x7C00u);
    }
    if (x < 0x38800000u) { /* result is subnormal or zero: value < 2^-14 */
        if (x < 0x33000000u) return sign; /* < 2^-25 -> 0 (ties-to-even at exactly 2^-25 also -> 0) */
        const int e = (int)(x >> 23);                 /* biased float exponent, 102..112 */
        uint32_t man = (x & 0x7FFFFFu) | 0x800000u;   /* 24-bit significand */
        const int shift = 126 - e;                    /* 14..24: bits to drop to land on 2^-24 units */
        const uint32_t q = man >> shift, rem = man & ((1u << shift) - 1u), half = 1u << (shift - 1);
        uint32_t r = q;
        if (rem > half || (rem == half && (q & 1u))) ++r;
        return (uint16_t)(sign | r);
    }
    /* normal range */
    uint32_t e = (x >> 23) - 112u, man = x & 0x7FFFFFu;
    uint32_t r = (e << 10) | (man >> 13);
    const uint32_t rem = man & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r; /* carry may bump the exponent: still correct */
    return (uint16_t)(sign | r);
}
static inline uint16_t hadd(uint16_t a, uint16_t b) { return f2h(h2f(a) + h2f(b)); }
static inline uint16_t hsub(uint16_t a, uint16_t b) { return f2h(h2f(a) - h2f(b)); }
static inline uint16_t hmul(uint16_t a, uint16_t b) { return f2h(h2f(a) * h2f(b)); }

void n2m_oracle_f32_to_f16(const float* in, uint16_t* out, uint64_t n) { for (uint64_t i = 0; i < n; ++i) out[i] = f2h(in[i]); }
void n2m_oracle_f16_to_f32(const uint16_t* in, float* out, uint64_t n) { for (uint64_t i = 0; i < n; ++i) out[i] = h2f(in[i]); }

/* ------------------------------------------------------------------------------------- small ray utilities */

/* raymarching.cu:91-145: slab test, axis by axis, with the early-outs of the reference. */
void n2m_oracle_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N,
                                   float min_near, float* nears, float* fars) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; ++n) {
        const float* o = rays_o + 3 * n;
        const float* d = rays_d + 3 * n;
        float tn = 0.f, tf = 0.f;
        int hit = 1;
        for (int a = 0; a < 3 && hit; ++a) {
            const float inv = 1.0f / d[a];
            float lo = (aabb[a] - o[a]) * inv, hi = (aabb[a + 3] - o[a]) * inv;
            if (lo > hi) { const float t = lo; lo = hi; hi = t; }
            if (a == 0) { tn = lo; tf = hi; continue; }
            if (tn > hi || lo > tf) { hit = 0; break; }
            if (lo > tn) tn = lo;
            if (hi < tf) tf = hi;
        }
        if (!hit) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (tn < min_near) tn = min_near;
        nears[n] = tn; fars[n] = tf;
    }
}

/* raymarching.cu:162-198 */
void n2m_oracle_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords) {
    const float rpi = 0.3183098861837907f;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; ++n) {
        const float ox = rays_o[3 * n], oy = rays_o[3 * n + 1], oz = rays_o[3 * n + 2];
        const float dx = rays_d[3 * n], dy = rays_d[3 * n + 1], dz = rays_d[3 * n + 2];
        const float A = dx * dx + dy * dy + dz * dz;
        const float Bh = ox * dx + oy * dy + oz * dz;
        const float Cc = ox * ox + oy * oy + oz * oz - radius * radius;
        const float t = (-Bh + sqrtf(Bh * Bh - A * Cc)) / A;
        const float x = ox + t * dx, y = oy + t * dy, z = oz + t * dz;
        const float theta = atan2f(sqrtf(x * x + z * z), y);
        const float phi = atan2f(z, x);
        coords[2 * n] = 2 * theta * rpi - 1;
        coords[2 * n + 1] = phi * rpi;
    }
}

void n2m_oracle_morton3D(const int32_t* coords, uint32_t N, int32_t* indices) { /* :214-226 */
    for (uint32_t n = 0; n < N; ++n)
        indices[n] = (int32_t)morton_encode((uint32_t)coords[3 * n], (uint32_t)coords[3 * n + 1], (uint32_t)coords[3 * n + 2]);
}
void n2m_oracle_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords) { /* :237-254 */
    for (uint32_t n = 0; n < N; ++n) {
        const int32_t v = indices[n]; /* arithmetic shifts on the signed value, as the reference */
        coords[3 * n] = (int32_t)gather3((uint32_t)(v >> 0));
        coords[3 * n + 1] = (int32_t)gather3((uint32_t)(v >> 1));
        coords[3 * n + 2] = (int32_t)gather3((uint32_t)(v >> 2));
    }
}

void n2m_oracle_packbits(const float* grid, uint32_t N, float thresh, uint8_t* bitfield) { /* :267-289 */
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; ++n) {
        unsigned bits = 0;
        for (int i = 0; i < 8; ++i) bits |= (grid[8 * n + i] > thresh) ? (1u << i) : 0u;
        bitfield[n] = (uint8_t)bits;
    }
}

void n2m_oracle_flatten_rays(const int32_t* rays, uint32_t N, uint32_t M, int32_t* res) { /* :303-319 */
    (void)M;
    for (uint32_t n = 0; n < N; ++n) {
        const uint32_t off = (uint32_t)rays[2 * n], cnt = (uint32_t)rays[2 * n + 1];
        for (uint32_t i = 0; i < cnt; ++i) res[off + i] = (int32_t)n;
    }
}

/* -------------------------------------------------------------------------------------------- the marcher */

typedef struct {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
    float bound, dt_gamma, dt_min, dt_max, rH, H3f, Hf, Cf;
    uint32_t H;
    int contract;
    const uint8_t* bits;
} march_ctx;

typedef struct { float cx, cy, cz, t_after, dt; } march_sample;

/* One iteration of the DDA loop body (raymarching.cu:396-465 == :759-827).  Advances *t.  Returns 1 and
 * fills `s` when a sample is kept, 0 when an empty voxel was skipped. */
static inline int march_step(const march_ctx* c, float* t_io, march_sample* s) {
    float t = *t_io;
    const float x = clampf(c->ox + t * c->dx, -c->bound, c->bound);
    const float y = clampf(c->oy + t * c->dy, -c->bound, c->bound);
    const float z = clampf(c->oz + t * c->dz, -c->bound, c->bound);
    float dt = clampf(t * c->dt_gamma, c->dt_min, c->dt_max);

    const int lp = level_from_pos(x, y, z, c->Cf), ld = level_from_dt(dt, c->Hf, c->Cf);
    const int level = lp > ld ? lp : ld;
    const float mip_bound = fminf(scalbnf(1.0f, level), c->bound);
    const float mip_rbound = 1.0f / mip_bound;

    float cx = x, cy = y, cz = z;
    const float mag = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    const int outside = c->contract && mag > 1.0f;
    if (outside) { /* L-inf contraction (:413-419) */
        const float k = (2.0f - 1.0f / mag) / mag;
        cx *= k; cy *= k; cz *= k;
    }
    /* voxel coordinates: the `0.5 *` literal makes the product double; clamp() narrows it back to float,
     * then the assignment to int truncates (:422-424) */
    const float top = (float)(c->H - 1);
    const int nx = (int)clampf((float)(0.5 * (double)(cx * mip_rbound + 1.0f) * (double)c->H), 0.0f, top);
    const int ny = (int)clampf((float)(0.5 * (double)(cy * mip_rbound + 1.0f) * (double)c->H), 0.0f, top);
    const int nz = (int)clampf((float)(0.5 * (double)(cz * mip_rbound + 1.0f) * (double)c->H), 0.0f, top);
    /* bit index is formed in float: level * H3 + morton (:379,426) */
    const uint32_t index = (uint32_t)((float)level * c->H3f + (float)morton_encode((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
    const int occ = (c->bits[index >> 3] >> (index & 7u)) & 1;

    if (occ || outside) {
        t += dt;
        s->cx = cx; s->cy = cy; s->cz = cz; s->t_after = t; s->dt = dt;
        *t_io = t;
        return 1;
    }
    /* skip to the exit face of this voxel, in whole steps (:454-463) */
    const float tx = (((nx + 0.5f + 0.5f * sign1(c->dx)) * c->rH * 2 - 1) * mip_bound - cx) * c->rdx;
    const float ty = (((ny + 0.5f + 0.5f * sign1(c->dy)) * c->rH * 2 - 1) * mip_bound - cy) * c->rdy;
    const float tz = (((nz + 0.5f + 0.5f * sign1(c->dz)) * c->rH * 2 - 1) * mip_bound - cz) * c->rdz;
    const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    do {
        dt = clampf(t * c->dt_gamma, c->dt_min, c->dt_max);
        t += dt;
    } while (t < tt);
    *t_io = t;
    return 0;
}

static inline void march_ctx_init(march_ctx* c, const float* o, const float* d, float eps, const uint8_t* bits,
                                  float bound, int contract, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H) {
    c->ox = o[0]; c->oy = o[1]; c->oz = o[2];
    c->dx = d[0]; c->dy = d[1]; c->dz = d[2];
    /* train: 1/d (:377); inference: 1/(d + 1e-10f) (:744) */
    c->rdx = 1.0f / (c->dx + eps); c->rdy = 1.0f / (c->dy + eps); c->rdz = 1.0f / (c->dz + eps);
    c->bound = bound; c->dt_gamma = dt_gamma; c->contract = contract; c->bits = bits; c->H = H;
    c->Hf = (float)H; c->Cf = (float)C;
    c->rH = 1.0f / (float)H;
    c->H3f = (float)(H * H * H);
    c->dt_min = 2 * 1.7320508075688772f / (float)max_steps; /* :385 */
    c->dt_max = 2 * 1.7320508075688772f * bound / (float)H; /* :386 */
}

/* raymarching.cu:337-475.  Pass 1 when xyzs == NULL. */
void n2m_oracle_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound,
                                 int contract, float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C,
                                 uint32_t H, const float* nears, const float* fars, float* xyzs, float* dirs,
                                 float* ts, int32_t* rays, int32_t* counter, const float* noises) {
    const int first_pass = (xyzs == NULL);
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t n = 0; n < (int64_t)N; ++n) {
        march_ctx c;
        march_ctx_init(&c, rays_o + 3 * n, rays_d + 3 * n, 0.0f, grid, bound, contract, dt_gamma, max_steps, C, H);
        uint32_t budget = max_steps;
        float *px = NULL, *pd = NULL, *pt = NULL;
        if (!first_pass) {
            const uint32_t off = (uint32_t)rays[2 * n];
            budget = (uint32_t)rays[2 * n + 1];
            px = xyzs + 3 * (size_t)off; pd = dirs + 3 * (size_t)off; pt = ts + 2 * (size_t)off;
        }
        const float far = fars[n];
        float t = nears[n];
        t += clampf(t * dt_gamma, c.dt_min, c.dt_max) * noises[n]; /* :389-391 */
        uint32_t kept = 0;
        march_sample s;
        while (t < far && kept < budget) {
            if (!march_step(&c, &t, &s)) continue;
            if (!first_pass) {
                px[0] = s.cx; px[1] = s.cy; px[2] = s.cz;
                pd[0] = c.dx; pd[1] = c.dy; pd[2] = c.dz;
                pt[0] = s.t_after; pt[1] = s.dt;
                px += 3; pd += 3; pt += 2;
            }
            ++kept;
        }
        if (first_pass) rays[2 * n + 1] = (int32_t)kept;
    }
    if (first_pass) { /* offsets: exclusive scan in ray order from the counter's entry value (:470-474) */
        uint32_t run = (uint32_t)counter[0];
        for (uint32_t n = 0; n < N; ++n) { rays[2 * n] = (int32_t)run; run += (uint32_t)rays[2 * n + 1]; }
        counter[0] = (int32_t)run;
    }
}

/* raymarching.cu:712-828 */
void n2m_oracle_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                           const float* rays_o, const float* rays_d, float bound, int contract, float dt_gamma,
                           uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid, const float* nears,
                           const float* fars, float* xyzs, float* dirs, float* ts, const float* noises) {
    (void)nears;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t n = 0; n < (int64_t)n_alive; ++n) {
        const int32_t ray = rays_alive[n];
        march_ctx c;
        march_ctx_init(&c, rays_o + 3 * (size_t)ray, rays_d + 3 * (size_t)ray, 1e-10f, grid, bound, contract,
                       dt_gamma, max_steps, C, H);
        float* px = xyzs + 3 * (size_t)n * n_step;
        float* pd = dirs + 3 * (size_t)n * n_step;
        float* pt = ts + 2 * (size_t)n * n_step;
        const float far = fars[ray];
        float t = rays_t[ray];
        t += clampf(t * dt_gamma, c.dt_min, c.dt_max) * noises[n];
        uint32_t kept = 0;
        march_sample s;
        while (t < far && kept < n_step) {
            if (!march_step(&c, &t, &s)) continue;
            px[0] = s.cx; px[1] = s.cy; px[2] = s.cz;
            pd[0] = c.dx; pd[1] = c.dy; pd[2] = c.dz;
            pt[0] = s.t_after; pt[1] = s.dt;
            px += 3; pd += 3; pt += 2;
            ++kept;
        }
    }
}

/* ------------------------------------------------------------------------------------------- compositing */

/* raymarching.cu:500-578 */
void n2m_oracle_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* ts,
                                             const int32_t* rays, uint32_t M, uint32_t N, float T_thresh,
                                             int alpha_mode, float* weights, float* weights_sum, float* depth,
                                             float* image) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t n = 0; n < (int64_t)N; ++n) {
        const uint32_t off = (uint32_t)rays[2 * n], cnt = (uint32_t)rays[2 * n + 1];
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, d = 0;
        if (cnt != 0 && off + cnt <= M) {
            for (uint32_t i = off; i < off + cnt; ++i) {
                const float alpha = alpha_mode ? sigmas[i] : (1.0f - expf(-sigmas[i] * ts[2 * i + 1]));
                const float w = alpha * T;
                weights[i] = w;
                r += w * rgbs[3 * i]; g += w * rgbs[3 * i + 1]; b += w * rgbs[3 * i + 2];
                ws += w;
                d += w * ts[2 * i];
                T *= 1.0f - alpha;
                if (T < T_thresh) break; /* the sample that crosses the threshold is still counted (:554-557) */
            }
        }
        weights_sum[n] = ws; depth[n] = d;
        image[3 * n] = r; image[3 * n + 1] = g; image[3 * n + 2] = b;
    }
}

/* raymarching.cu:604-694 */
void n2m_oracle_composite_rays_train_backward(const float* grad_weights, const float* grad_weights_sum,
                                              const float* grad_depth, const float* grad_image,
                                              const float* sigmas, const float* rgbs, const float* ts,
                                              const int32_t* rays, const float* weights_sum, const float* depth,
                                              const float* image, uint32_t M, uint32_t N, float T_thresh,
                                              int alpha_mode, float* grad_sigmas, float* grad_rgbs) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t n = 0; n < (int64_t)N; ++n) {
        const uint32_t off = (uint32_t)rays[2 * n], cnt = (uint32_t)rays[2 * n + 1];
        if (cnt == 0 || off + cnt > M) continue;
        const float gi0 = grad_image[3 * n], gi1 = grad_image[3 * n + 1], gi2 = grad_image[3 * n + 2];
        const float gws = grad_weights_sum[n], gd = grad_depth[n];
        const float rF = image[3 * n], gF = image[3 * n + 1], bF = image[3 * n + 2], wsF = weights_sum[n], dF = depth[n];
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, d = 0;
        for (uint32_t i = off; i < off + cnt; ++i) {
            const float alpha = alpha_mode ? sigmas[i] : (1.0f - expf(-sigmas[i] * ts[2 * i + 1]));
            const float w = alpha * T;
            r += w * rgbs[3 * i]; g += w * rgbs[3 * i + 1]; b += w * rgbs[3 * i + 2];
            ws += w;
            d += w * ts[2 * i];
            T *= 1.0f - alpha; /* T is now the transmittance AFTER sample i */
            grad_rgbs[3 * i] = gi0 * w; grad_rgbs[3 * i + 1] = gi1 * w; grad_rgbs[3 * i + 2] = gi2 * w;
            const float scale = alpha_mode ? (1.0f / (1.0f - alpha)) : ts[2 * i + 1];
            grad_sigmas[i] = scale * (gi0 * (T * rgbs[3 * i] - (rF - r)) + gi1 * (T * rgbs[3 * i + 1] - (gF - g)) +
                                      gi2 * (T * rgbs[3 * i + 2] - (bF - b)) + (gws + grad_weights[i]) * (T - (wsF - ws)) +
                                      gd * (T * ts[2 * i] - (dF - d)));
            if (T < T_thresh) break;
        }
    }
}

/* raymarching.cu:841-924 */
void n2m_oracle_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int alpha_mode,
                               int32_t* rays_alive, float* rays_t, const float* sigmas, const float* rgbs,
                               const float* ts, float* weights_sum, float* depth, float* image) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)n_alive; ++n) {
        const int32_t ray = rays_alive[n];
        const size_t base = (size_t)n * n_step;
        float t = 0.0f; /* the reference leaves it uninitialised; it is only stored when step == n_step */
        float d = depth[ray], r = image[3 * ray], g = image[3 * ray + 1], b = image[3 * ray + 2], ws = weights_sum[ray];
        uint32_t step = 0;
        while (step < n_step) {
            const size_t i = base + step;
            if (ts[2 * i] == 0) break; /* empty slot: ray finished (:877) */
            const float alpha = alpha_mode ? sigmas[i] : (1.0f - expf(-sigmas[i] * ts[2 * i + 1]));
            const float T = 1 - ws; /* transmittance from the running alpha sum (:887) */
            const float w = alpha * T;
            ws += w;
            t = ts[2 * i];
            d += w * t;
            r += w * rgbs[3 * i]; g += w * rgbs[3 * i + 1]; b += w * rgbs[3 * i + 2];
            if (T < T_thresh) break;
            ++step;
        }
        if (step < n_step) rays_alive[n] = -1; else rays_t[ray] = t;
        weights_sum[ray] = ws; depth[ray] = d;
        image[3 * ray] = r; image[3 * ray + 1] = g; image[3 * ray + 2] = b;
    }
}

/* order-preserving compaction of the alive list (== rays_alive[rays_alive >= 0], nerf/renderer.py:798) */
void n2m_oracle_compact_alive(const int32_t* rays_alive, uint32_t n_alive, int32_t* out, int32_t* n_out) {
    int32_t k = 0;
    for (uint32_t i = 0; i < n_alive; ++i) if (rays_alive[i] >= 0) out[k++] = rays_alive[i];
    *n_out = k;
}

/* ---------------------------------------------------------------------------------------------- hash grid */

typedef struct { float scale; uint32_t resolution, size, offset; } level_geom;

/* per-level geometry (gridencoder.cu:137-139): scale = 2^(l*S)*H - 1, resolution = ceil(scale) + 1 */
static inline level_geom level_setup(const int32_t* offsets, uint32_t level, float S, uint32_t H) {
    level_geom g;
    g.offset = (uint32_t)offsets[level];
    g.size = (uint32_t)(offsets[level + 1] - offsets[level]);
    g.scale = exp2f((float)level * S) * (float)H - 1.0f;
    g.resolution = (uint32_t)ceilf(g.scale) + 1u;
    return g;
}

void n2m_oracle_level_geometry(const int32_t* offsets, uint32_t L, float S, uint32_t H, float* scales, uint32_t* resolutions) {
    for (uint32_t l = 0; l < L; ++l) { const level_geom g = level_setup(offsets, l, S, H); scales[l] = g.scale; resolutions[l] = g.resolution; }
}

/* row index of a grid vertex (gridencoder.cu:50-84): dense stride index while the running stride fits the
 * level's table, else (hash type) xor-of-primes; always reduced modulo the table size. */
static inline uint32_t vertex_row(const uint32_t* v, uint32_t D, uint32_t gridtype, int align_corners,
                                  uint32_t size, uint32_t resolution) {
    static const uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
    uint32_t stride = 1, index = 0;
    for (uint32_t d = 0; d < D && stride <= size; ++d) {
        index += v[d] * stride;
        stride *= align_corners ? resolution : resolution + 1u;
    }
    if (gridtype == 0 && stride > size) {
        index = 0;
        for (uint32_t d = 0; d < D; ++d) index ^= v[d] * primes[d];
    }
    return index % size;
}

static inline int out_of_unit_cube(const float* x, uint32_t D) {
    for (uint32_t d = 0; d < D; ++d) if (x[d] < 0 || x[d] > 1) return 1;
    return 0;
}

/* cell + interpolation fractions of a point at one level (gridencoder.cu:146-158) */
static inline void locate(const float* x, uint32_t D, float scale, int align_corners, uint32_t interp,
                          uint32_t* cell, float* frac, float* dfrac) {
    for (uint32_t d = 0; d < D; ++d) {
        float p = x[d] * scale + (align_corners ? 0.0f : 0.5f);
        const float fl = floorf(p);
        cell[d] = (uint32_t)fl;
        p -= (float)cell[d];
        if (interp == 1) { dfrac[d] = 6 * p * (1.0f - p); p = p * p * (3.0f - 2.0f * p); }
        else dfrac[d] = 1.0f;
        frac[d] = p;
    }
}

#define GRID_AT(ptr, is_half, i) ((is_half) ? h2f(((const uint16_t*)(ptr))[i]) : ((const float*)(ptr))[i])

/* gridencoder.cu:87-244.  outputs are level-major [L,B,C]; `sample_major` != 0 writes [B,L*C] instead
 * (the layout gridencoder/grid.py:63 derives with a permute) and zero-fills levels >= max_level. */
static void grid_forward_impl(const float* inputs, const void* emb, const int32_t* offsets, void* outputs,
                              uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level, float S,
                              uint32_t H, void* dy_dx, uint32_t gridtype, int align_corners, uint32_t interp,
                              int is_half, int sample_major) {
    for (uint32_t level = 0; level < L; ++level) {
        if (level >= max_level) {
            if (sample_major)
                for (uint32_t b = 0; b < B; ++b) for (uint32_t ch = 0; ch < C; ++ch) {
                    const size_t o = (size_t)b * L * C + (size_t)level * C + ch;
                    if (is_half) ((uint16_t*)outputs)[o] = 0; else ((float*)outputs)[o] = 0.0f;
                }
            continue;
        }
        const level_geom g = level_setup(offsets, level, S, H);
#pragma omp parallel for schedule(static)
        for (int64_t b = 0; b < (int64_t)B; ++b) {
            const float* x = inputs + (size_t)b * D;
            const size_t obase = sample_major ? ((size_t)b * L * C + (size_t)level * C) : ((size_t)level * B * C + (size_t)b * C);
            const size_t gbase = (size_t)b * L * D * C + (size_t)level * D * C;
            if (out_of_unit_cube(x, D)) {
                for (uint32_t ch = 0; ch < C; ++ch) { if (is_half) ((uint16_t*)outputs)[obase + ch] = 0; else ((float*)outputs)[obase + ch] = 0.0f; }
                if (dy_dx) for (uint32_t k = 0; k < D * C; ++k) { if (is_half) ((uint16_t*)dy_dx)[gbase + k] = 0; else ((float*)dy_dx)[gbase + k] = 0.0f; }
                continue;
            }
            uint32_t cell[N2M_MAX_D], v[N2M_MAX_D];
            float frac[N2M_MAX_D], dfrac[N2M_MAX_D];
            locate(x, D, g.scale, align_corners, interp, cell, frac, dfrac);

            float accf[N2M_MAX_C] = {0};
            uint16_t acch[N2M_MAX_C] = {0};
            for (uint32_t corner = 0; corner < (1u << D); ++corner) {
                float w = 1.0f;
                for (uint32_t d = 0; d < D; ++d) {
                    if (corner & (1u << d)) { w *= frac[d]; v[d] = cell[d] + 1; }
                    else { w *= 1 - frac[d]; v[d] = cell[d]; }
                }
                const size_t row = (size_t)(g.offset + vertex_row(v, D, gridtype, align_corners, g.size, g.resolution)) * C;
                for (uint32_t ch = 0; ch < C; ++ch) {
                    if (is_half) acch[ch] = hadd(acch[ch], f2h(w * h2f(((const uint16_t*)emb)[row + ch]))); /* half accumulator (:163,186) */
                    else accf[ch] += w * ((const float*)emb)[row + ch];
                }
            }
            for (uint32_t ch = 0; ch < C; ++ch) { if (is_half) ((uint16_t*)outputs)[obase + ch] = acch[ch]; else ((float*)outputs)[obase + ch] = accf[ch]; }

            if (!dy_dx) continue;
            /* analytic d(out)/d(x_gd) (:200-243): corners of the other D-1 axes, right minus left on axis gd */
            for (uint32_t gd = 0; gd < D; ++gd) {
                float gf[N2M_MAX_C] = {0};
                uint16_t gh[N2M_MAX_C] = {0};
                for (uint32_t corner = 0; corner < (1u << (D - 1)); ++corner) {
                    float w = g.scale;
                    for (uint32_t nd = 0; nd < D - 1; ++nd) {
                        const uint32_t d = nd >= gd ? nd + 1 : nd;
                        if (corner & (1u << nd)) { w *= frac[d]; v[d] = cell[d] + 1; }
                        else { w *= 1 - frac[d]; v[d] = cell[d]; }
                    }
                    v[gd] = cell[gd];
                    const size_t left = (size_t)(g.offset + vertex_row(v, D, gridtype, align_corners, g.size, g.resolution)) * C;
                    v[gd] = cell[gd] + 1;
                    const size_t right = (size_t)(g.offset + vertex_row(v, D, gridtype, align_corners, g.size, g.resolution)) * C;
                    for (uint32_t ch = 0; ch < C; ++ch) {
                        if (is_half) {
                            const uint16_t diff = hsub(((const uint16_t*)emb)[right + ch], ((const uint16_t*)emb)[left + ch]);
                            gh[ch] = hadd(gh[ch], f2h(w * h2f(diff) * dfrac[gd]));
                        } else gf[ch] += w * (((const float*)emb)[right + ch] - ((const float*)emb)[left + ch]) * dfrac[gd];
                    }
                }
                for (uint32_t ch = 0; ch < C; ++ch) {
                    if (is_half) ((uint16_t*)dy_dx)[gbase + gd * C + ch] = gh[ch]; else ((float*)dy_dx)[gbase + gd * C + ch] = gf[ch];
                }
            }
        }
    }
}

void n2m_oracle_grid_encode_forward(const float* inputs, const void* emb, const int32_t* offsets, void* outputs,
                                    uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level, float S,
                                    uint32_t H, void* dy_dx, uint32_t gridtype, int align_corners, uint32_t interp,
                                    int dtype) {
    grid_forward_impl(inputs, emb, offsets, outputs, B, D, C, L, max_level, S, H, dy_dx, gridtype, align_corners, interp, dtype == 1, 0);
}
void n2m_oracle_grid_encode_forward_bm(const float* inputs, const void* emb, const int32_t* offsets, void* outputs,
                                       uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level, float S,
                                       uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, int dtype) {
    grid_forward_impl(inputs, emb, offsets, outputs, B, D, C, L, max_level, S, H, NULL, gridtype, align_corners, interp, dtype == 1, 1);
}

/* gridencoder.cu:247-339 (+ :342-368).  The reference visits (thread -> sample b, channel pair) in thread
 * order inside each level; a serial sweep over b, then channel group, reproduces its summation order.
 * N_C = min(2, C) channels per thread (:404); half2-packed adds when half and N_C even (:324-330). */
static void grid_backward_impl(const void* grad, const float* inputs, const void* emb, const int32_t* offsets,
                               void* grad_emb, uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level,
                               float S, uint32_t H, const void* dy_dx, void* grad_inputs, uint32_t gridtype,
                               int align_corners, uint32_t interp, int is_half, int sample_major) {
    (void)emb;
    const uint32_t NC = C < 2 ? C : 2;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t lv = 0; lv < (int64_t)max_level; ++lv) {
        const uint32_t level = (uint32_t)lv;
        const level_geom g = level_setup(offsets, level, S, H);
        for (uint32_t b = 0; b < B; ++b) {
            const float* x = inputs + (size_t)b * D;
            if (out_of_unit_cube(x, D)) continue;
            uint32_t cell[N2M_MAX_D], v[N2M_MAX_D];
            float frac[N2M_MAX_D], dfrac[N2M_MAX_D];
            locate(x, D, g.scale, align_corners, interp, cell, frac, dfrac);
            const size_t gb = sample_major ? ((size_t)b * L * C + (size_t)level * C) : ((size_t)level * B * C + (size_t)b * C);
            for (uint32_t ch0 = 0; ch0 < C; ch0 += NC) {
                for (uint32_t corner = 0; corner < (1u << D); ++corner) {
                    float w = 1.0f;
                    for (uint32_t d = 0; d < D; ++d) {
                        if (corner & (1u << d)) { w *= frac[d]; v[d] = cell[d] + 1; }
                        else { w *= 1 - frac[d]; v[d] = cell[d]; }
                    }
                    const size_t row = (size_t)(g.offset + vertex_row(v, D, gridtype, align_corners, g.size, g.resolution)) * C + ch0;
                    for (uint32_t c = 0; c < NC; ++c) {
                        if (is_half) {
                            uint16_t* dst = (uint16_t*)grad_emb + row + c;
                            *dst = hadd(*dst, f2h(w * h2f(((const uint16_t*)grad)[gb + ch0 + c])));
                        } else ((float*)grad_emb)[row + c] += w * ((const float*)grad)[gb + ch0 + c];
                    }
                }
            }
        }
    }
    if (dy_dx && grad_inputs) { /* :342-368, accumulates in scalar_t over ALL L levels */
#pragma omp parallel for schedule(static)
        for (int64_t t = 0; t < (int64_t)B * D; ++t) {
            const uint32_t b = (uint32_t)(t / D), d = (uint32_t)(t - (int64_t)b * D);
            float rf = 0; uint16_t rh = 0;
            for (uint32_t l = 0; l < L; ++l) for (uint32_t ch = 0; ch < C; ++ch) {
                const size_t gi = (size_t)l * B * C + (size_t)b * C + ch;
                const size_t di = (size_t)b * L * D * C + (size_t)l * D * C + (size_t)d * C + ch;
                if (is_half) rh = hadd(rh, hmul(((const uint16_t*)grad)[gi], ((const uint16_t*)dy_dx)[di]));
                else rf += ((const float*)grad)[gi] * ((const float*)dy_dx)[di];
            }
            if (is_half) ((uint16_t*)grad_inputs)[t] = rh; else ((float*)grad_inputs)[t] = rf;
        }
    }
}

void n2m_oracle_grid_encode_backward(const void* grad, const float* inputs, const void* emb, const int32_t* offsets,
                                     void* grad_emb, uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level,
                                     float S, uint32_t H, const void* dy_dx, void* grad_inputs, uint32_t gridtype,
                                     int align_corners, uint32_t interp, int dtype) {
    grid_backward_impl(grad, inputs, emb, offsets, grad_emb, B, D, C, L, max_level, S, H, dy_dx, grad_inputs, gridtype, align_corners, interp, dtype == 1, 0);
}
void n2m_oracle_grid_encode_backward_bm(const void* grad, const float* inputs, const void* emb, const int32_t* offsets,
                                        void* grad_emb, uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level,
                                        float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, int dtype) {
    grid_backward_impl(grad, inputs, emb, offsets, grad_emb, B, D, C, L, max_level, S, H, NULL, NULL, gridtype, align_corners, interp, dtype == 1, 1);
}

/* Exact-sum form of the table backward, for tight parity bars on kernels that do NOT follow the reference's (order-dependent)
 * summation: per row and channel, the sum IN DOUBLE of the terms the reference adds -- for half tables each term is first rounded
 * like the reference rounds it, (__half)(w * grad) with the product rounded to fp32 in between (gridencoder.cu:326); for fp32
 * tables the fp32 product -- plus the sum of their magnitudes (for rounding bounds) and the number of terms.  Same traversal as
 * grid_backward_impl; level-major grad [L,B,C]. */
void n2m_oracle_grid_encode_backward_exact(const void* grad, const float* inputs, const int32_t* offsets, double* sum, double* abs_sum,
                                           uint32_t* count, uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t max_level, float S,
                                           uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, int dtype) {
    const int is_half = dtype == 1;
    (void)L;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t lv = 0; lv < (int64_t)max_level; ++lv) {
        const uint32_t level = (uint32_t)lv;
        const level_geom g = level_setup(offsets, level, S, H);
        for (uint32_t b = 0; b < B; ++b) {
            const float* x = inputs + (size_t)b * D;
            if (out_of_unit_cube(x, D)) continue;
            uint32_t cell[N2M_MAX_D], v[N2M_MAX_D];
            float frac[N2M_MAX_D], dfrac[N2M_MAX_D];
            locate(x, D, g.scale, align_corners, interp, cell, frac, dfrac);
            const size_t gb = (size_t)level * B * C + (size_t)b * C;
            for (uint32_t corner = 0; corner < (1u << D); ++corner) {
                float w = 1.0f;
                for (uint32_t d = 0; d < D; ++d) {
                    if (corner & (1u << d)) { w *= frac[d]; v[d] = cell[d] + 1; }
                    else { w *= 1 - frac[d]; v[d] = cell[d]; }
                }
                const size_t row = (size_t)(g.offset + vertex_row(v, D, gridtype, align_corners, g.size, g.resolution)) * C;
                for (uint32_t c = 0; c < C; ++c) {
                    volatile float p = is_half ? w * h2f(((const uint16_t*)grad)[gb + c]) : w * ((const float*)grad)[gb + c];
                    const double t = is_half ? (double)h2f(f2h(p)) : (double)p;
                    sum[row + c] += t;
                    abs_sum[row + c] += t < 0 ? -t : t;
                    if (t != 0.0) count[row + c] += 1;
                }
            }
        }
    }
}

/* gridencoder.cu:505-609, fp32 tables only (the reference's half path ends in an empty atomicAdd stub,
 * gridencoder.cu:22-26, and grid.py:170 runs TV with autocast disabled). */
void n2m_oracle_grad_total_variation(const float* inputs, const float* emb, float* grad, const int32_t* offsets,
                                     float weight, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                     uint32_t gridtype, int align_corners) {
    const float w = weight / (float)(2 * D);
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t lv = 0; lv < (int64_t)L; ++lv) {
        const level_geom g = level_setup(offsets, (uint32_t)lv, S, H);
        const float* tab = emb + (size_t)g.offset * C;
        float* gtab = grad + (size_t)g.offset * C;
        for (uint32_t b = 0; b < B; ++b) {
            const float* x = inputs + (size_t)b * D;
            if (out_of_unit_cube(x, D)) continue;
            uint32_t cell[N2M_MAX_D];
            for (uint32_t d = 0; d < D; ++d) cell[d] = (uint32_t)floorf(x[d] * g.scale + (align_corners ? 0.0f : 0.5f));
            const size_t here = (size_t)vertex_row(cell, D, gridtype, align_corners, g.size, g.resolution) * C;
            float sum[N2M_MAX_C] = {0}, sq[N2M_MAX_C] = {0};
            for (uint32_t d = 0; d < D; ++d) {
                const uint32_t cur = cell[d];
                if (cur < g.resolution) { /* +1 neighbour (:571) */
                    cell[d] = cur + 1;
                    const size_t nb = (size_t)vertex_row(cell, D, gridtype, align_corners, g.size, g.resolution) * C;
                    for (uint32_t ch = 0; ch < C; ++ch) { const float dv = tab[here + ch] - tab[nb + ch]; sum[ch] += dv; sq[ch] += dv * dv; }
                }
                if (cur > 0) { /* -1 neighbour (:585) */
                    cell[d] = cur - 1;
                    const size_t nb = (size_t)vertex_row(cell, D, gridtype, align_corners, g.size, g.resolution) * C;
                    for (uint32_t ch = 0; ch < C; ++ch) { const float dv = tab[here + ch] - tab[nb + ch]; sum[ch] += dv; sq[ch] += dv * dv; }
                }
                cell[d] = cur;
            }
            for (uint32_t ch = 0; ch < C; ++ch) gtab[here + ch] += w * sum[ch] * (1.0f / sqrtf(sq[ch] + 1e-9f));
        }
    }
}

/* ------------------------------------------------------------------------------------ spherical harmonics */

/* shencoder.cu:27-356 writes the 64 real SH basis functions (degree <= 8) of a unit vector as expanded
 * polynomials, index l*l + l + m, sign (-1)^m (Condon-Shortley), e.g. out[1] = -c1*y, out[2] = c1*z,
 * out[3] = -c1*x (:52-54).  Every entry factors as  N(l,m) * Pbar(l,|m|)(z) * {Re,Im}((x+iy)^|m|)  with
 * Pbar = P_l^m(z) / (1-z^2)^(m/2) (a polynomial in z); the reference's expanded forms are exactly these
 * products (e.g. :76 `0.473*(x2-y2)*(7*z2-1)`), also away from the unit sphere.  This restatement evaluates
 * that factorisation with the standard three-term recurrences in DOUBLE and rounds once, so it agrees with
 * the reference's fp32 polynomial evaluation to fp32 rounding error (not bit-for-bit: the association order
 * differs; tolerance stated in tests/test_oracle_vs_reference.py).
 * dy_dx [B,3,deg^2] holds the partial derivatives of those polynomials (:125-354):
 *   d/dx Re_m = m Re_{m-1}, d/dx Im_m = m Im_{m-1}, d/dy Re_m = -m Im_{m-1}, d/dy Im_m = m Re_{m-1},
 *   d/dz Pbar(l,m) = Pbar(l,m+1). */
void n2m_oracle_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t degree, float* dy_dx) {
    const uint32_t C2 = degree * degree;
    const double PI = 3.14159265358979323846;
    /* normalisation N(l,m) = sqrt((2l+1)/(4 pi) * (l-m)!/(l+m)!), times sqrt(2) and (-1)^m for m > 0 */
    double Nlm[8][8];
    for (uint32_t l = 0; l < degree; ++l) for (uint32_t m = 0; m <= l; ++m) {
        double ratio = 1.0; /* (l-m)! / (l+m)! */
        for (uint32_t k = l - m + 1; k <= l + m; ++k) ratio /= (double)k;
        double v = sqrt((2.0 * l + 1.0) / (4.0 * PI) * ratio);
        if (m > 0) v *= sqrt(2.0) * ((m & 1u) ? -1.0 : 1.0);
        Nlm[l][m] = v;
    }
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < (int64_t)B; ++b) {
        const double x = inputs[(size_t)b * D], y = inputs[(size_t)b * D + 1], z = inputs[(size_t)b * D + 2];
        double re[9], im[9];   /* (x+iy)^m, m = 0..8 */
        re[0] = 1; im[0] = 0;
        for (uint32_t m = 1; m <= 8; ++m) { re[m] = re[m - 1] * x - im[m - 1] * y; im[m] = re[m - 1] * y + im[m - 1] * x; }
        double P[9][10]; /* Pbar(l,m), with m up to l+1 = 0 padding for the z-derivative */
        memset(P, 0, sizeof(P));
        for (uint32_t m = 0; m <= 8; ++m) {
            double pmm = 1.0;
            for (uint32_t k = 1; k <= m; ++k) pmm *= (2.0 * k - 1.0);
            P[m][m] = pmm;
            if (m + 1 <= 8) P[m + 1][m] = (2.0 * m + 1.0) * z * pmm;
            for (uint32_t l = m + 2; l <= 8; ++l)
                P[l][m] = ((2.0 * l - 1.0) * z * P[l - 1][m] - (double)(l + m - 1) * P[l - 2][m]) / (double)(l - m);
        }
        float* out = outputs + (size_t)b * C2;
        float* gx = dy_dx ? dy_dx + (size_t)b * D * C2 : NULL;
        float* gy = gx ? gx + C2 : NULL;
        float* gz = gy ? gy + C2 : NULL;
        for (uint32_t l = 0; l < degree; ++l) {
            for (uint32_t m = 0; m <= l; ++m) {
                const double n = Nlm[l][m], p = P[l][m], dp = (m + 1 <= l) ? P[l][m + 1] : 0.0;
                const uint32_t ip = l * l + l + m, in = l * l + l - m;
                out[ip] = (float)(n * p * re[m]);
                if (m) out[in] = (float)(n * p * im[m]);
                if (!gx) continue;
                const double dre_dx = m ? m * re[m - 1] : 0.0, dre_dy = m ? -(double)m * im[m - 1] : 0.0;
                const double dim_dx = m ? m * im[m - 1] : 0.0, dim_dy = m ? m * re[m - 1] : 0.0;
                gx[ip] = (float)(n * p * dre_dx); gy[ip] = (float)(n * p * dre_dy); gz[ip] = (float)(n * dp * re[m]);
                if (m) { gx[in] = (float)(n * p * dim_dx); gy[in] = (float)(n * p * dim_dy); gz[in] = (float)(n * dp * im[m]); }
            }
        }
    }
}

/* shencoder.cu:358-382: grad_inputs[b,d] += sum_ch grad[b,ch] * dy_dx[b,d,ch], fp32 running sum in memory order */
void n2m_oracle_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t degree,
                                   const float* dy_dx, float* grad_inputs) {
    (void)inputs;
    const uint32_t C2 = degree * degree;
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < (int64_t)B * D; ++t) {
        const size_t b = (size_t)(t / D);
        const float* g = grad + b * C2;
        const float* j = dy_dx + (size_t)t * C2;
        float acc = grad_inputs[t];
        for (uint32_t ch = 0; ch < C2; ++ch) acc += g[ch] * j[ch];
        grad_inputs[t] = acc;
    }
}

/* ------------------------------------------------------------------------------------------------------------------
 * freqencoder (freqencoder/src/freqencoder.cu:30-94): outputs [B, C], C = D + 2*deg*D: column block 0 is the input, block
 * 1 + 2f is sin(2^f x), block 2 + 2f is sin(2^f x + pi/2) (the reference's cosine, :58-60; __sinf there, sinf here).
 * backward (:66-94): d/dx = g_self + sum_f 2^f (g_sin cos - g_cos sin), using the stored outputs. */
void n2m_oracle_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float* outputs) {
    const float PI_F = 3.141592653589793f;
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < (int64_t)B * C; ++t) {
        const uint32_t b = (uint32_t)(t / C), c = (uint32_t)(t - (int64_t)b * C);
        const float* in = inputs + (size_t)b * D;
        if (c < D) { outputs[t] = in[c]; continue; }
        const uint32_t col = c / D - 1, d = c % D, freq = col / 2;
        const float phase_shift = (float)(col % 2) * (PI_F / 2);
        outputs[t] = sinf(scalbnf(in[d], (int)freq) + phase_shift);
    }
    (void)deg;
}

void n2m_oracle_freq_encode_backward(const float* grad, const float* outputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                                     float* grad_inputs) {
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < (int64_t)B * D; ++t) {
        const uint32_t b = (uint32_t)(t / D), d = (uint32_t)(t - (int64_t)b * D);
        const float* g = grad + (size_t)b * C;
        const float* o = outputs + (size_t)b * C;
        float result = g[d];
        g += D; o += D;
        for (uint32_t f = 0; f < deg; ++f) {
            result += scalbnf(1.0f, (int)f) * (g[d] * o[D + d] - g[D + d] * o[d]);
            g += 2 * D; o += 2 * D;
        }
        grad_inputs[t] = result;
    }
}

/* ------------------------------------------------------------------------------------------------ marching cubes
 * PARITY UNPINNED (PyMCubes, the reference's `mcubes.marching_cubes` of nerf/renderer.py:524-527, is an un-vendored dependency that is
 * absent here): a plain-C restatement of the extraction documented in oracle/marching_cubes.py, for volumes too large for that file's
 * pure-Python emission loop.  The 256-case table is NOT in this file: the caller passes the one oracle/marching_cubes.py derives
 * (num_tris[256], tris[256][stride] edge ids; edge e = 4 * axis + (u | v << 1), owned by the node at its lower corner).
 * solid = !(value < iso) compared in double; vertex = lower corner + (iso - f1) / (f2 - f1) in double, mapped ((p / div) * mul) + add,
 * then cast to float; vertices by node (C order) then axis, triangles by cell (C order) then table order.
 * counts[0] = vertices, counts[1] = triangles; entries beyond cap_v / cap_t are counted but not written. */
void n2m_oracle_marching_cubes(const float* f, uint32_t R0, uint32_t R1, uint32_t R2, double iso, const uint8_t* num_tris,
                               const uint8_t* tris, uint32_t stride, double div, double mul, double add, float* vertices, uint64_t cap_v,
                               int32_t* triangles, uint64_t cap_t, uint64_t* counts) {
    const size_t plane = (size_t)R1 * R2, N = (size_t)R0 * plane;
    int64_t* vid = (int64_t*)malloc(N * 3 * sizeof(int64_t));     /* vertex id of the edge leaving node n along axis a, or -1 */
    uint64_t nv = 0, nt = 0;
    const size_t step[3] = {plane, R2, 1};
    for (size_t n = 0; n < N; ++n) {
        const uint32_t i = (uint32_t)(n / plane), j = (uint32_t)((n % plane) / R2), k = (uint32_t)(n % R2);
        const uint32_t pos[3] = {i, j, k}, ext[3] = {R0, R1, R2};
        const double f1 = (double)f[n];
        const int s1 = !(f1 < iso);
        for (int a = 0; a < 3; ++a) {
            vid[n * 3 + a] = -1;
            if (pos[a] + 1 >= ext[a]) continue;
            const double f2 = (double)f[n + step[a]];
            if ((!(f2 < iso)) == s1) continue;
            double p[3] = {(double)i, (double)j, (double)k};
            p[a] = p[a] + (iso - f1) / (f2 - f1);
            if (nv < cap_v)
                for (int d = 0; d < 3; ++d) vertices[nv * 3 + d] = (float)(((p[d] / div) * mul) + add);
            vid[n * 3 + a] = (int64_t)nv++;
        }
    }
    if (R0 >= 2 && R1 >= 2 && R2 >= 2) {
        for (uint32_t i = 0; i + 1 < R0; ++i)
            for (uint32_t j = 0; j + 1 < R1; ++j)
                for (uint32_t k = 0; k + 1 < R2; ++k) {
                    const size_t n = (size_t)i * plane + (size_t)j * R2 + k;
                    uint32_t cas = 0;
                    for (uint32_t c = 0; c < 8; ++c) {
                        const size_t m = n + ((c & 1u) ? plane : 0) + ((c & 2u) ? R2 : 0) + ((c & 4u) ? 1 : 0);
                        cas |= (uint32_t)(!((double)f[m] < iso)) << c;
                    }
                    for (uint32_t t = 0; t < num_tris[cas]; ++t, ++nt) {
                        if (nt >= cap_t) continue;
                        for (int m3 = 0; m3 < 3; ++m3) {
                            const uint32_t e = tris[(size_t)cas * stride + 3u * t + m3];
                            const uint32_t a = e >> 2, u = e & 1u, v = (e >> 1) & 1u;
                            const uint32_t dx = a == 0 ? 0 : u, dy = a == 0 ? u : (a == 1 ? 0 : v), dz = a == 2 ? 0 : v;
                            triangles[nt * 3 + m3] = (int32_t)vid[(n + dx * plane + (size_t)dy * R2 + dz) * 3 + a];
                        }
                    }
                }
    }
    counts[0] = nv;
    counts[1] = nt;
    free(vid);
}
