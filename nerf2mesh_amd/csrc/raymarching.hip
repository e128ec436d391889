// Ray marching + compositing for gfx950 (wave64).  Replaces the reference's _raymarching_mob extension
// (raymarching/src/raymarching.cu); entry points and their reference counterparts are listed in
// include/n2m_hip.h.
//
// Design notes (MI355X):
//  * marchers: one ray per lane, 64-thread workgroups so that even a 4096-ray batch spreads over 64 CUs with
//    one wave per SIMD (the loop is latency-bound: a dependent DDA chain with bit-field lookups that stay in
//    L1/L2 -- 256 KiB for lego).  Index-deciding arithmetic follows oracle/n2m_oracle.c operation for
//    operation and this file is compiled with -ffp-contract=off, so sample counts, offsets and sample
//    positions are bit-identical to the oracle.
//  * sample packing: counts are turned into offsets by an exclusive scan IN RAY ORDER (deterministic; the
//    reference's atomicAdd hands out offsets in arrival order).  A wave's 64 rays therefore own one contiguous
//    slab of the sample buffers.
//  * compositing (train): one WAVE per ray, lanes = 64 consecutive samples, transmittance by a wave prefix
//    product, colour/depth sums by wave reductions (forward) / prefix sums (backward).  All loads/stores are
//    coalesced; the serial per-ray loop of the reference becomes ceil(count/64) wave steps.
#include <float.h>
#include <stdlib.h>

#include "n2m_common.hpp"

namespace {

constexpr float kSqrt3 = 1.7320508075688772f;

// ------------------------------------------------------------------------------------------- small kernels

// slab test of one ray (raymarching.cu:91-145): (near, far), both FLT_MAX on a miss
__device__ __forceinline__ void near_far_of(const float (&o)[3], const float (&d)[3], const float* __restrict__ aabb, float min_near,
                                            float& near, float& far) {
    float tn = 0.f, tf = 0.f;
    bool hit = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (!hit) break;
        const float inv = 1.0f / d[a];
        const float org = o[a];
        float lo = (aabb[a] - org) * inv, hi = (aabb[a + 3] - org) * inv;
        if (lo > hi) { const float t = lo; lo = hi; hi = t; }
        if (a == 0) { tn = lo; tf = hi; continue; }
        if (tn > hi || lo > tf) { hit = false; break; }
        if (lo > tn) tn = lo;
        if (hi < tf) tf = hi;
    }
    if (!hit) { near = FLT_MAX; far = FLT_MAX; return; }
    if (tn < min_near) tn = min_near;
    near = tn;
    far = tf;
}

__global__ void near_far_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                const float* __restrict__ aabb, uint32_t N, float min_near,
                                float* __restrict__ nears, float* __restrict__ fars) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float o[3] = {rays_o[3 * n], rays_o[3 * n + 1], rays_o[3 * n + 2]}, d[3] = {rays_d[3 * n], rays_d[3 * n + 1], rays_d[3 * n + 2]};
    near_far_of(o, d, aabb, min_near, nears[n], fars[n]);
}

// A whole training batch from ONE tensor of uniforms u [N,6] in [0,1): view = floor(u0 V), pixel = floor(u1 H W) (random_image_batch,
// nerf/provider.py:302-303 + get_rays with N random pixels, nerf/utils.py:271), rays exactly as n2m_get_rays builds them
// (nerf/utils.py:242-290), ground-truth gather (nerf/provider.py:330), near/far exactly as n2m_near_far_from_aabb, march jitter u2,
// random background u3..u5 (nerf/utils.py:649-652); also clears the marcher's sample counter.  Seven small launches of the step's
// side stream become one (they ran 7-13 us EACH beside the optimizer update and delayed the march behind them).
__global__ void __launch_bounds__(256)
batch_rays_kernel(const float* __restrict__ poses /*[V,4,4]*/, const float* __restrict__ u /*[N,6]*/, uint32_t V, uint32_t N, uint32_t W,
                  uint32_t HW, float fx, float fy, float cx, float cy, const float* __restrict__ images /*[V,HW,4]*/,
                  const float* __restrict__ aabb, float min_near, float* __restrict__ rays_o, float* __restrict__ rays_d,
                  float* __restrict__ rgba, float* __restrict__ nears, float* __restrict__ fars, float* __restrict__ noises,
                  float* __restrict__ bg, int32_t* __restrict__ counter, const float* __restrict__ cam_near_far /*[V,2] or NULL*/) {
    const uint32_t n = blockIdx.x * 256 + threadIdx.x;
    if (n == 0 && counter) counter[0] = 0;
    if (n >= N) return;
    const float* __restrict__ un = u + (size_t)n * 6;
    const uint32_t v = min(V - 1u, (uint32_t)(un[0] * (float)V)), p = min(HW - 1u, (uint32_t)(un[1] * (float)HW));
    const float i = (float)(p % W) + 0.5f, j = (float)(p / W) + 0.5f;
    const float d0 = (i - cx) / fx, d1 = -(j - cy) / fy, d2 = -1.0f;
    const float* __restrict__ P = poses + (size_t)v * 16;
    float o[3], d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        d[k] = (d0 * P[4 * k] + d1 * P[4 * k + 1]) + d2 * P[4 * k + 2];
        o[k] = P[4 * k + 3];
        rays_d[(size_t)n * 3 + k] = d[k];
        rays_o[(size_t)n * 3 + k] = o[k];
    }
    *reinterpret_cast<float4*>(rgba + (size_t)n * 4) = *reinterpret_cast<const float4*>(images + ((size_t)v * HW + (size_t)p) * 4);
    float tn, tf;
    near_far_of(o, d, aabb, min_near, tn, tf);
    if (cam_near_far) {       // per-view clamp from the sparse points (nerf/renderer.py:689-691, colmap_provider.py:563-565): maximum / minimum
        tn = fmaxf(tn, cam_near_far[2 * v]);
        tf = fminf(tf, cam_near_far[2 * v + 1]);
    }
    nears[n] = tn; fars[n] = tf;
    noises[n] = un[2];
    if (bg) { bg[(size_t)n * 3] = un[3]; bg[(size_t)n * 3 + 1] = un[4]; bg[(size_t)n * 3 + 2] = un[5]; }
}

__global__ void sph_from_ray_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d, float radius,
                                    uint32_t N, float* __restrict__ coords) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float rpi = 0.3183098861837907f;
    const float ox = rays_o[3 * n], oy = rays_o[3 * n + 1], oz = rays_o[3 * n + 2];
    const float dx = rays_d[3 * n], dy = rays_d[3 * n + 1], dz = rays_d[3 * n + 2];
    const float A = dx * dx + dy * dy + dz * dz;
    const float Bh = ox * dx + oy * dy + oz * dz;
    const float Cc = ox * ox + oy * oy + oz * oz - radius * radius;
    const float t = (-Bh + sqrtf(Bh * Bh - A * Cc)) / A;
    const float x = ox + t * dx, y = oy + t * dy, z = oz + t * dz;
    coords[2 * n] = 2 * atan2f(sqrtf(x * x + z * z), y) * rpi - 1;
    coords[2 * n + 1] = atan2f(z, x) * rpi;
}

__global__ void morton_kernel(const int32_t* __restrict__ coords, uint32_t N, int32_t* __restrict__ indices) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    indices[n] = (int32_t)n2m_morton((uint32_t)coords[3 * n], (uint32_t)coords[3 * n + 1], (uint32_t)coords[3 * n + 2]);
}

__global__ void morton_invert_kernel(const int32_t* __restrict__ indices, uint32_t N, int32_t* __restrict__ coords) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int32_t v = indices[n];
    coords[3 * n] = (int32_t)n2m_gather3((uint32_t)(v >> 0));
    coords[3 * n + 1] = (int32_t)n2m_gather3((uint32_t)(v >> 1));
    coords[3 * n + 2] = (int32_t)n2m_gather3((uint32_t)(v >> 2));
}

// One lane packs 4 output bytes from 32 floats read as 8 x float4 (128 B per lane, 8 KiB per wave-iteration).
__global__ void packbits_kernel(const float* __restrict__ grid, uint32_t N, float thresh, uint8_t* __restrict__ bitfield,
                                const float* __restrict__ thresh_dev = nullptr) {
    if (thresh_dev) thresh = *thresh_dev;
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;   // index of a group of 4 output bytes
    const uint32_t nq = N >> 2;
    if (q < nq) {
        const float4* src = reinterpret_cast<const float4*>(grid) + (size_t)q * 8;
        uint32_t word = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 v = src[k];
            word |= (uint32_t)(v.x > thresh) << (4 * k);
            word |= (uint32_t)(v.y > thresh) << (4 * k + 1);
            word |= (uint32_t)(v.z > thresh) << (4 * k + 2);
            word |= (uint32_t)(v.w > thresh) << (4 * k + 3);
        }
        reinterpret_cast<uint32_t*>(bitfield)[q] = word;
    } else if (q == nq) {   // tail bytes when N is not a multiple of 4
        for (uint32_t n = nq * 4; n < N; ++n) {
            uint32_t bits = 0;
            for (int i = 0; i < 8; ++i) bits |= (uint32_t)(grid[(size_t)n * 8 + i] > thresh) << i;
            bitfield[n] = (uint8_t)bits;
        }
    }
}

__global__ void packbits_bytes_kernel(const float* __restrict__ grid, uint32_t N, float thresh, uint8_t* __restrict__ bitfield,
                                      const float* __restrict__ thresh_dev = nullptr) {
    if (thresh_dev) thresh = *thresh_dev;
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    uint32_t bits = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) bits |= (uint32_t)(grid[(size_t)n * 8 + i] > thresh) << i;
    bitfield[n] = (uint8_t)bits;
}

// one wave per ray: lanes fill the ray's run cooperatively
__global__ void flatten_rays_kernel(const int32_t* __restrict__ rays, uint32_t N, uint32_t M, int32_t* __restrict__ res) {
    const uint32_t n = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (n >= N) return;
    const uint32_t off = (uint32_t)rays[2 * n], cnt = (uint32_t)rays[2 * n + 1];
    for (uint32_t i = lane; i < cnt; i += 64)
        if (off + i < M) res[off + i] = (int32_t)n;
}

// ------------------------------------------------------------------------------------------------ marcher

struct MarchCtx {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
    float bound, dt_gamma, dt_min, dt_max, rH, H3f, Hf, Cf, top, halfH;
    bool contract;
    const uint8_t* __restrict__ bits;
};

struct MarchSample { float cx, cy, cz, t_after, dt; };

__device__ __forceinline__ int level_from_exponent(float mx, float cascades) {
    int e;
    (void)frexpf(mx, &e);
    return (int)fminf(cascades - 1.0f, fmaxf(0.0f, (float)e));
}

__device__ __forceinline__ void march_ctx_init(MarchCtx& c, const float* o, const float* d, float eps, const uint8_t* bits,
                                               float bound, bool contract, float dt_gamma, uint32_t max_steps, uint32_t C,
                                               uint32_t H) {
    c.ox = o[0]; c.oy = o[1]; c.oz = o[2];
    c.dx = d[0]; c.dy = d[1]; c.dz = d[2];
    c.rdx = 1.0f / (c.dx + eps); c.rdy = 1.0f / (c.dy + eps); c.rdz = 1.0f / (c.dz + eps);
    c.bound = bound; c.dt_gamma = dt_gamma; c.contract = contract; c.bits = bits;
    c.Hf = (float)H; c.halfH = 0.5f * (float)H; c.Cf = (float)C; c.top = (float)(H - 1);
    c.rH = 1.0f / (float)H;
    c.H3f = (float)(H * H * H);
    c.dt_min = 2 * kSqrt3 / (float)max_steps;
    c.dt_max = 2 * kSqrt3 * bound / (float)H;
}

// One iteration of the occupancy-grid DDA.  Returns true when a sample is kept (t advanced by one step),
// false when an empty voxel was skipped (t advanced past its exit face in whole steps).
__device__ __forceinline__ bool march_step(const MarchCtx& c, float& t, MarchSample& s) {
    const float x = n2m_clampf(c.ox + t * c.dx, -c.bound, c.bound);
    const float y = n2m_clampf(c.oy + t * c.dy, -c.bound, c.bound);
    const float z = n2m_clampf(c.oz + t * c.dz, -c.bound, c.bound);
    float dt = n2m_clampf(t * c.dt_gamma, c.dt_min, c.dt_max);

    const float mag = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    const int lp = level_from_exponent(mag, c.Cf);
    const int ld = level_from_exponent((dt * c.Hf) * 0.5f, c.Cf);   // *0.5 is exact in either precision
    const int level = lp > ld ? lp : ld;
    const float mip_bound = fminf(scalbnf(1.0f, level), c.bound);
    const float mip_rbound = 1.0f / mip_bound;

    float cx = x, cy = y, cz = z;
    const bool outside = c.contract && mag > 1.0f;
    if (outside) {
        const float k = (2.0f - 1.0f / mag) / mag;
        cx *= k; cy *= k; cz *= k;
    }
    // reference: clamp(0.5 * (c*r + 1) * H) evaluated in double and narrowed (raymarching.cu:422-424).  0.5*H is
    // exactly representable and the double product of a float by it is exact before the single narrowing, so one
    // fp32 multiply by 0.5f*H gives the identical, correctly rounded value for every H (oracle-checked).
    const int nx = (int)n2m_clampf((cx * mip_rbound + 1.0f) * c.halfH, 0.0f, c.top);
    const int ny = (int)n2m_clampf((cy * mip_rbound + 1.0f) * c.halfH, 0.0f, c.top);
    const int nz = (int)n2m_clampf((cz * mip_rbound + 1.0f) * c.halfH, 0.0f, c.top);
    const uint32_t index = (uint32_t)((float)level * c.H3f + (float)n2m_morton((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
    const bool occ = (c.bits[index >> 3] >> (index & 7u)) & 1u;

    if (occ || outside) {
        t += dt;
        s.cx = cx; s.cy = cy; s.cz = cz; s.t_after = t; s.dt = dt;
        return true;
    }
    const float tx = ((((float)nx + 0.5f + 0.5f * copysignf(1.0f, c.dx)) * c.rH * 2 - 1) * mip_bound - cx) * c.rdx;
    const float ty = ((((float)ny + 0.5f + 0.5f * copysignf(1.0f, c.dy)) * c.rH * 2 - 1) * mip_bound - cy) * c.rdy;
    const float tz = ((((float)nz + 0.5f + 0.5f * copysignf(1.0f, c.dz)) * c.rH * 2 - 1) * mip_bound - cz) * c.rdz;
    const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    do {
        dt = n2m_clampf(t * c.dt_gamma, c.dt_min, c.dt_max);
        t += dt;
    } while (t < tt);
    return false;
}


// ---------------------------------------------------------------------------------- wave-per-ray marcher (train)
// Both ways the reference advances a ray -- the sampling step (`t += dt`, raymarching.cu:433-434) and the empty-voxel
// skip (`do { dt = clamp(t*dt_gamma); t += dt; } while (t < tt)`, :460-463) -- apply the SAME update
//     t <- t + clamp(t * dt_gamma, dt_min, dt_max)
// so every t a ray ever visits is a member of one sequence T_0 = t0, T_{k+1} = T_k + clamp(T_k*g, ..) that does not
// depend on the occupancy grid.  One wavefront therefore marches one ray 64 candidates at a time:
//   1. lane j derives T_{base+j} with the sequential fp32 recurrence (bit-identical to the serial loop),
//   2. every lane evaluates its candidate in parallel: clamped position, cascade level, voxel, occupancy bit and, for
//      an empty voxel, the exit time tt of that voxel (the exact expressions of march_step),
//   3. the visited subsequence is resolved with ballots: a run of occupied lanes is kept wholesale, an empty lane
//      jumps to the first later lane with T >= tt (carried into the next chunk when it lies beyond this one),
//   4. WRITE pass: kept lanes store their sample at offset + popcount(kept lanes below) -- consecutive lanes,
//      consecutive rows, coalesced.
// Result: identical samples, counts and order as the serial kernel, ~64x shorter dependent chain per ray, and
// N waves (not N/64) to fill the machine.
struct LaneEval { float cx, cy, cz, dt, tt; bool keep; };

// Position (contracted when outside the unit box) and step of candidate t: the part of eval_candidate a kept sample's row is made of.
__device__ __forceinline__ void candidate_point(const MarchCtx& c, float t, float& cx, float& cy, float& cz, float& dt) {
    cx = n2m_clampf(c.ox + t * c.dx, -c.bound, c.bound);
    cy = n2m_clampf(c.oy + t * c.dy, -c.bound, c.bound);
    cz = n2m_clampf(c.oz + t * c.dz, -c.bound, c.bound);
    dt = n2m_clampf(t * c.dt_gamma, c.dt_min, c.dt_max);
    const float mag = fmaxf(fabsf(cx), fmaxf(fabsf(cy), fabsf(cz)));
    if (c.contract && mag > 1.0f) {
        const float k = (2.0f - 1.0f / mag) / mag;
        cx *= k; cy *= k; cz *= k;
    }
}

__device__ __forceinline__ LaneEval eval_candidate(const MarchCtx& c, float t) {
    LaneEval e;
    const float x = n2m_clampf(c.ox + t * c.dx, -c.bound, c.bound);
    const float y = n2m_clampf(c.oy + t * c.dy, -c.bound, c.bound);
    const float z = n2m_clampf(c.oz + t * c.dz, -c.bound, c.bound);
    const float dt = n2m_clampf(t * c.dt_gamma, c.dt_min, c.dt_max);
    const float mag = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int level = 0;
    float mip_bound = fminf(1.0f, c.bound);              // level 0: min(2^0, bound)
    if (c.Cf > 1.0f) {                                    // one cascade (bound <= 1): level is min(C-1, .) = 0 whatever the exponents say
        const int lp = level_from_exponent(mag, c.Cf);
        const int ld = level_from_exponent((dt * c.Hf) * 0.5f, c.Cf);
        level = lp > ld ? lp : ld;
        mip_bound = fminf(scalbnf(1.0f, level), c.bound);
    }
    const float mip_rbound = 1.0f / mip_bound;
    float cx = x, cy = y, cz = z;
    const bool outside = c.contract && mag > 1.0f;
    if (outside) {
        const float k = (2.0f - 1.0f / mag) / mag;
        cx *= k; cy *= k; cz *= k;
    }
    const int nx = (int)n2m_clampf((cx * mip_rbound + 1.0f) * c.halfH, 0.0f, c.top);
    const int ny = (int)n2m_clampf((cy * mip_rbound + 1.0f) * c.halfH, 0.0f, c.top);
    const int nz = (int)n2m_clampf((cz * mip_rbound + 1.0f) * c.halfH, 0.0f, c.top);
    const uint32_t index = (uint32_t)((float)level * c.H3f + (float)n2m_morton((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
    const bool occ = (c.bits[index >> 3] >> (index & 7u)) & 1u;
    const float tx = ((((float)nx + 0.5f + 0.5f * copysignf(1.0f, c.dx)) * c.rH * 2 - 1) * mip_bound - cx) * c.rdx;
    const float ty = ((((float)ny + 0.5f + 0.5f * copysignf(1.0f, c.dy)) * c.rH * 2 - 1) * mip_bound - cy) * c.rdy;
    const float tz = ((((float)nz + 0.5f + 0.5f * copysignf(1.0f, c.dz)) * c.rH * 2 - 1) * mip_bound - cz) * c.rdz;
    e.cx = cx; e.cy = cy; e.cz = cz; e.dt = dt;
    e.tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    e.keep = occ || outside;
    return e;
}

// rays whose parallel resolution failed its check and were redone serially (diagnostics: n2m_march_fallback_count)
__device__ unsigned int g_march_fallbacks;

// Inclusive prefix maximum across the wave on the DPP network (no LDS pipe): Hillis-Steele inside each row of 16 lanes
// (row_shr 1/2/4/8), then lane 15 of rows 0/2 into rows 1/3 (row_bcast:15) and lane 31 into rows 2/3 (row_bcast:31).
__device__ __forceinline__ float wave_scan_max_dpp(float v) {
    const int ninf = (int)0xff800000u;
#define N2M_DPP_MAX(ctrl, rows) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(ninf, __float_as_int(v), (ctrl), (rows), 0xf, false)))
    N2M_DPP_MAX(0x111, 0xf);   // row_shr:1
    N2M_DPP_MAX(0x112, 0xf);   // row_shr:2
    N2M_DPP_MAX(0x114, 0xf);   // row_shr:4
    N2M_DPP_MAX(0x118, 0xf);   // row_shr:8
    N2M_DPP_MAX(0x142, 0xa);   // row_bcast:15 -> rows 1, 3
    N2M_DPP_MAX(0x143, 0xc);   // row_bcast:31 -> rows 2, 3
#undef N2M_DPP_MAX
    return v;
}

// A chunk of 64 candidates that kept at least one sample, as the single-pass marcher remembers it between counting and writing:
// which lanes were kept, T of lane 0 and (constant-step closed form) the per-candidate increment of T's bit pattern, 0 = recurrence.
struct ChunkRec { unsigned long long mask; uint32_t tbits; uint32_t m; };
constexpr uint32_t kChunkRecCap = 24;

// One ray, one wave: the candidates of the ray from t_start to `far`, at most `budget` samples kept (WRITE: stored at rows out, out + 1, ...).
// Shared by the training marcher (march_train_one_ray: start = near + jitter, budget = max_steps or the counted samples) and the
// wave-per-ray inference marcher (march_infer_wave_*: start = the ray's current t, budget = n_step).
template <bool WRITE, bool RECORD = false>
__device__ __forceinline__ uint32_t march_one_ray_core(const MarchCtx& c, int lane, float far, float t_start, uint32_t budget, size_t out,
                        float* __restrict__ xyzs, float* __restrict__ dirs, float* __restrict__ ts, bool force_serial,
                        ChunkRec* rec = nullptr, uint32_t* n_rec_out = nullptr) {
    uint32_t n_rec = 0;
    // Resolution of the visited subsequence, two ways.  PARALLEL (first attempt): the serial chain keeps an occupied candidate j
    // unless a VISITED empty candidate i < j jumps over it, i.e. has exit time tt_i > T_j.  If no empty candidate at all -- visited
    // or not -- has tt_i > T_j for any later occupied j (a prefix maximum over the lanes, carried across chunks), the chain never
    // skips an occupied candidate and visits every one of them (induction over the chain: an empty candidate jumps to the first
    // T_k >= tt_i and everything in between is empty by the condition), so the kept set is exactly the occupied candidates, cut by
    // the sample budget and by `far`: no serial loop.  The condition only fails when rounding pushes a voxel's exit time past a
    // candidate that already lies in the next, occupied voxel (~1e-4 per empty->occupied crossing); the ray is then redone from
    // its start with the SERIAL resolution (the exact chain state is not tracked by the parallel form).
    bool serial = force_serial;
restart_ray:
    float t_base = t_start;
    uint32_t kept = 0;
    n_rec = 0;
    bool pending = false;      // an empty voxel's exit time lies beyond the previous chunk
    float pending_tt = 0.f;
    float carry_tt = -INFINITY; // parallel form: largest exit time of any empty candidate in earlier chunks
    // a ray can only end by t >= far or by the sample budget; the chunk cap turns a non-advancing t (far = inf with
    // t so large that t + dt == t: the serial reference would spin forever) into a bounded loop
    for (uint32_t chunk = 0; chunk < (1u << 20) && t_base < far && kept < budget; ++chunk) {
        // 1. lane j <- T_{base + j}: j sequential applications of the update, in serial fp32 order
        float t = t_base, t_next_base;
        // Constant step (dt_gamma == 0: t += dt_min).  Inside one binade [2^e, 2^(e+1)) every t is a multiple of
        // u = 2^(e-23); with dt = (m + r) u, |r| < 1/2, the serial update RN(t + dt) is EXACTLY t + m u -- i.e. the
        // integer m added to t's bit pattern -- so lane j reads T_{base+j} = bits(t_base) + j m in closed form.  Chunks
        // that leave the binade, and the tie |r| = 1/2 (round-to-even depends on t), take the serial loop below.
        bool closed_form = false;
        uint32_t cf_m = 0;
        if (c.dt_gamma == 0.0f) {
            const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(t_base));   // wave-uniform: scalar branch
            const uint32_t be = b >> 23;                                     // sign 0 and a normal exponent
            if (be >= 1u && be <= 254u && c.dt_min <= c.dt_max) {
                const int sh = 23 - ((int)be - 127) + 127;                   // biased exponent of 1/u
                if (sh >= 1 && sh <= 254) {
                    const float q = c.dt_min * __uint_as_float((uint32_t)sh << 23);      // dt / u, an exact scaling
                    const float mq = rintf(q);
                    if (q < 8388608.0f && mq >= 1.0f && fabsf(q - mq) != 0.5f) {
                        const uint32_t m = (uint32_t)mq;
                        const uint32_t last = b + 64u * m;
                        if ((last >> 23) == be) {
                            closed_form = true;
                            cf_m = m;
                            t = __uint_as_float(b + (uint32_t)lane * m);
                            t_next_base = __uint_as_float(last);
                        }
                    }
                }
            }
        }
        if (!closed_form) {
#pragma unroll 1
            for (int i = 0; i < 63; ++i) {
                const float nt = t + n2m_clampf(t * c.dt_gamma, c.dt_min, c.dt_max);
                t = i < lane ? nt : t;
            }
            t_next_base = __uint_as_float((uint32_t)__builtin_amdgcn_readlane(
                (int)__float_as_uint(t + n2m_clampf(t * c.dt_gamma, c.dt_min, c.dt_max)), 63));   // T_{base+64}
        }
        // 2. evaluate all candidates of the chunk
        const bool active = t < far;
        LaneEval e;
        e.keep = false; e.tt = t; e.cx = e.cy = e.cz = e.dt = 0.f;
        if (active) e = eval_candidate(c, t);
        const unsigned long long active_mask = __ballot(active);             // a prefix of the lanes (T increases)
        const unsigned long long keep_mask = __ballot(active && e.keep);
        unsigned long long kept_mask = 0ull;
        bool done = false;
        const uint32_t budget_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)budget);
        uint32_t kept_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)kept);
        if (!serial) {
            // 3a. parallel resolution (see above)
            const float incl = wave_scan_max_dpp((active && !e.keep) ? e.tt : -INFINITY);
            // exclusive form: lane j sees the maximum over lanes < j and over the earlier chunks (wave_shr:1, lane 0 keeps `old`)
            const float before = fmaxf(carry_tt, __int_as_float(__builtin_amdgcn_update_dpp((int)0xff800000u, __float_as_int(incl), 0x138, 0xf, 0xf, false)));
            if (__ballot(active && e.keep && t < before) != 0ull) {
                if (!WRITE && lane == 0) atomicAdd(&g_march_fallbacks, 1u);
                serial = true;
                goto restart_ray;
            }
            carry_tt = fmaxf(carry_tt, __uint_as_float((uint32_t)__builtin_amdgcn_readlane(__float_as_int(incl), 63)));
            const uint32_t room = budget_s - kept_s;
            kept_mask = keep_mask;
            if ((uint32_t)__popcll(keep_mask) > room)      // sample budget: the first `room` occupied candidates
                kept_mask = __ballot(active && e.keep && (uint32_t)__popcll(keep_mask & ((1ull << lane) - 1ull)) < room);
            kept_s += (uint32_t)__popcll(kept_mask);
            done = kept_s >= budget_s || active_mask != ~0ull;        // budget spent, or some T_j >= far lies in this chunk
        } else {
        // 3b. serial resolution.  Everything in this loop is wave-uniform (ballots, lane indices, counters); the
        // values are pinned to SGPRs with readfirstlane so that the loop runs on the scalar unit with scalar branches instead of
        // per-lane VALU compares under exec masks.
        int cur = 0;
        if (pending) {
            const unsigned long long reach = __ballot(t >= pending_tt);
            if (reach == 0ull) cur = 64;                                     // whole chunk lies inside the skipped voxel
            else { cur = (int)__ffsll((long long)reach) - 1; pending = false; }
        }
        cur = __builtin_amdgcn_readfirstlane(cur);
        while (cur < 64) {
            if (!((active_mask >> cur) & 1ull)) { done = true; break; }      // T_cur >= far
            if ((keep_mask >> cur) & 1ull) {
                const unsigned long long stop = ~keep_mask >> cur;           // first lane >= cur that is not kept
                const uint32_t run = stop ? (uint32_t)__ffsll((long long)stop) - 1u : (uint32_t)(64 - cur);
                const uint32_t take = min(run, budget_s - kept_s);
                kept_mask |= (take >= 64u ? ~0ull : ((1ull << take) - 1ull)) << cur;
                kept_s += take;
                if (kept_s >= budget_s) { done = true; break; }
                cur += (int)run;
            } else {
                // v_readlane with a scalar lane index instead of an LDS-pipe ds_bpermute round trip
                const float tt = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(e.tt), cur));
                const unsigned long long later = cur >= 63 ? 0ull : (~0ull << (cur + 1));
                const unsigned long long reach = __ballot(t >= tt) & later;
                if (reach) cur = (int)__ffsll((long long)reach) - 1;
                else { pending = true; pending_tt = tt; cur = 64; }
            }
            cur = __builtin_amdgcn_readfirstlane(cur);
        }
        }
        kept = kept_s;
        if (!WRITE && RECORD && kept_mask) {
            if (n_rec < kChunkRecCap && lane == 0) rec[n_rec] = ChunkRec{kept_mask, __float_as_uint(t_base), cf_m};
            ++n_rec;
        }
        // 4. emit
        if (WRITE && kept_mask) {
            if ((kept_mask >> lane) & 1ull) {
                const uint32_t rank = (uint32_t)__popcll(kept_mask & ((1ull << lane) - 1ull));
                const size_t row = out + (kept - (uint32_t)__popcll(kept_mask)) + rank;     // samples kept before this chunk + rank in it
                xyzs[3 * row] = e.cx; xyzs[3 * row + 1] = e.cy; xyzs[3 * row + 2] = e.cz;
                dirs[3 * row] = c.dx; dirs[3 * row + 1] = c.dy; dirs[3 * row + 2] = c.dz;
                *reinterpret_cast<float2*>(ts + 2 * row) = make_float2(t + e.dt, e.dt);
            }
        }
        if (done) break;
        t_base = t_next_base;
    }
    if (!WRITE && RECORD) *n_rec_out = n_rec;
    return kept;
}

// WRITE: emit samples (offset / budget from `rays`, or given by the caller when RECORD is set too: the single-pass marcher's fallback).
// !WRITE: count; with RECORD the non-empty chunks are logged into rec[0..kChunkRecCap) (wave-private LDS) and n_rec counts them all.
template <bool WRITE, bool RECORD = false>
__device__ __forceinline__ uint32_t march_train_one_ray(uint32_t n, int lane, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                        const uint8_t* __restrict__ grid,
                        float bound, bool contract, float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                        const float* __restrict__ nears, const float* __restrict__ fars, float* __restrict__ xyzs,
                        float* __restrict__ dirs, float* __restrict__ ts, int32_t* __restrict__ rays,
                        const float* __restrict__ noises, uint32_t max_points, bool force_serial,
                        ChunkRec* rec = nullptr, uint32_t* n_rec_out = nullptr, size_t out_given = 0, uint32_t budget_given = 0) {
    MarchCtx c;
    march_ctx_init(c, rays_o + 3 * (size_t)n, rays_d + 3 * (size_t)n, 0.0f, grid, bound, contract, dt_gamma, max_steps, C, H);
    uint32_t budget = max_steps;
    size_t out = 0;
    if (WRITE) {
        out = RECORD ? out_given : (size_t)(uint32_t)rays[2 * n];
        budget = RECORD ? budget_given : (uint32_t)rays[2 * n + 1];
        if (budget == 0 || out + budget > (size_t)max_points) return 0;     // does not fit the sample buffers (raymarching.cu:417)
    }
    const float far = fars[n];
    const float t_start = nears[n] + n2m_clampf(nears[n] * dt_gamma, c.dt_min, c.dt_max) * noises[n];
    const uint32_t kept = march_one_ray_core<WRITE, RECORD>(c, lane, far, t_start, budget, out, xyzs, dirs, ts, force_serial, rec, n_rec_out);
    if (!WRITE && lane == 0) rays[2 * n + 1] = (int32_t)kept;
    return kept;
}

// One wave per ray, four waves per workgroup; a workgroup walks rays blockIdx.x*4 + wave, += 4*gridDim.x.  With a grid that covers N
// this is one ray per wave; a SMALLER grid turns the kernel into a few resident waves per SIMD that work through the batch beside
// whatever else is running (the pass is queued on a second stream next to the training step's own kernels; a full-size grid
// slows an MFMA kernel it overlaps ~3.7x, measured -- but see march_grid: capping it cost more than it saved).
template <bool WRITE>
__global__ void __launch_bounds__(256)
march_train_wave_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d, const uint8_t* __restrict__ grid,
                        float bound, bool contract, float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                        const float* __restrict__ nears, const float* __restrict__ fars, float* __restrict__ xyzs,
                        float* __restrict__ dirs, float* __restrict__ ts, int32_t* __restrict__ rays,
                        const float* __restrict__ noises, uint32_t max_points, bool force_serial) {
    const int lane = threadIdx.x & 63;
    for (uint32_t n = blockIdx.x * 4 + (threadIdx.x >> 6); n < N; n += gridDim.x * 4)
        march_train_one_ray<WRITE>(n, lane, rays_o, rays_d, grid, bound, contract, dt_gamma, max_steps, N, C, H, nears, fars, xyzs, dirs, ts, rays,
                                   noises, max_points, force_serial);
}

// ---------------------------------------------------------------------------------- march once (count + record, then replay)
// The two-pass protocol (count, offset scan, march AGAIN to write: raymarching.py:229-241) with ONE march per ray:
//   march_train_record_kernel   the count pass; besides its count every ray leaves the chunks that kept samples (ChunkRec: kept-lane
//                               mask + T of lane 0 + closed-form increment, 16 B each) and adds its count to the total of its group
//                               of 256 rays (one no-return atomic per ray, 256 per address);
//   march_train_replay_kernel   a ray's offset = totals of the groups before its own + counts of the <= 255 rays before it inside the
//                               group (a handful of independent coalesced loads and one wave reduction: the deterministic ray-order
//                               prefix sum of the two-pass form, bit for bit, without a scan kernel in between); then the kept lanes of
//                               the recorded chunks are recomputed (T, position, step: no occupancy look-up, no resolution) and stored.
// (A single kernel with a decoupled look-back between the two halves was measured first: 99 us against 50 + 6 + 24 for count, scan and
// write -- the look-back chain over ~3 400 workgroups that all finish counting at about the same time costs more than it saves.)
constexpr uint32_t kMarchGroupLog2 = 8;

__global__ void __launch_bounds__(256)
march_train_record_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d, const uint8_t* __restrict__ grid,
                          float bound, bool contract, float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                          const float* __restrict__ nears, const float* __restrict__ fars, int32_t* __restrict__ rays,
                          const float* __restrict__ noises, bool force_serial, ChunkRec* __restrict__ recs, uint32_t* __restrict__ n_recs,
                          uint32_t* __restrict__ group_total) {
    const int lane = threadIdx.x & 63;
    const uint32_t n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    uint32_t n_rec = 0;
    const uint32_t kept = march_train_one_ray<false, true>(n, lane, rays_o, rays_d, grid, bound, contract, dt_gamma, max_steps, N, C, H, nears, fars,
                                                           nullptr, nullptr, nullptr, rays, noises, 0xFFFFFFFFu, force_serial,
                                                           recs + (size_t)n * kChunkRecCap, &n_rec);
    if (lane == 0) {
        n_recs[n] = n_rec;
        if (kept) __hip_atomic_fetch_add(group_total + (n >> kMarchGroupLog2), kept, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void __launch_bounds__(256)
march_train_replay_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d, const uint8_t* __restrict__ grid,
                          float bound, bool contract, float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                          const float* __restrict__ nears, const float* __restrict__ fars, float* __restrict__ xyzs,
                          float* __restrict__ dirs, float* __restrict__ ts, int32_t* __restrict__ rays, int32_t* __restrict__ counter,
                          const float* __restrict__ noises, uint32_t max_points, bool force_serial, const ChunkRec* __restrict__ recs,
                          const uint32_t* __restrict__ n_recs, const uint32_t* __restrict__ group_total) {
    const int lane = threadIdx.x & 63;
    const uint32_t n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    // offset: whole groups before this ray's + the rays before it inside its group
    const uint32_t g = n >> kMarchGroupLog2;
    uint32_t acc = 0;
    for (uint32_t i = (uint32_t)lane; i < g; i += 64u) acc += group_total[i];
    for (uint32_t i = (g << kMarchGroupLog2) + (uint32_t)lane; i < n; i += 64u) acc += (uint32_t)rays[2 * i + 1];
    const uint32_t kept = (uint32_t)rays[2 * n + 1], n_rec = n_recs[n];
    ChunkRec mine = ChunkRec{0ull, 0u, 0u};
    if ((uint32_t)lane < n_rec && n_rec <= kChunkRecCap) mine = recs[(size_t)n * kChunkRecCap + lane];      // lane r <- record r
    const uint32_t off = n2m_wave_sum_u32(acc);
    if (lane == 0) {
        rays[2 * n] = (int32_t)off;
        if (n == N - 1) counter[0] = (int32_t)(off + kept);                              // the batch's sample count
    }
    if (kept == 0 || (size_t)off + kept > (size_t)max_points) return;                   // does not fit the sample buffers (raymarching.cu:417)
    if (n_rec > kChunkRecCap) {                                                          // more sample-bearing chunks than recorded: march again
        march_train_one_ray<true, true>(n, lane, rays_o, rays_d, grid, bound, contract, dt_gamma, max_steps, N, C, H, nears, fars, xyzs, dirs, ts,
                                        rays, noises, max_points, force_serial, nullptr, nullptr, (size_t)off, kept);
        return;
    }
    MarchCtx c;
    march_ctx_init(c, rays_o + 3 * (size_t)n, rays_d + 3 * (size_t)n, 0.0f, grid, bound, contract, dt_gamma, max_steps, C, H);
    size_t row0 = off;
    for (uint32_t r = 0; r < n_rec; ++r) {
        const uint32_t mlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mine.mask, r);
        const uint32_t mhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mine.mask >> 32), r);
        const unsigned long long mask = ((unsigned long long)mhi << 32) | mlo;
        const uint32_t tbits = (uint32_t)__builtin_amdgcn_readlane((int)mine.tbits, r), m = (uint32_t)__builtin_amdgcn_readlane((int)mine.m, r);
        float t;
        if (m) t = __uint_as_float(tbits + (uint32_t)lane * m);
        else {
            t = __uint_as_float(tbits);
#pragma unroll 1
            for (int i = 0; i < 63; ++i) {
                const float nt = t + n2m_clampf(t * c.dt_gamma, c.dt_min, c.dt_max);
                t = i < lane ? nt : t;
            }
        }
        if ((mask >> lane) & 1ull) {
            float cx, cy, cz, dt;
            candidate_point(c, t, cx, cy, cz, dt);
            const size_t row = row0 + (size_t)__popcll(mask & ((1ull << lane) - 1ull));
            xyzs[3 * row] = cx; xyzs[3 * row + 1] = cy; xyzs[3 * row + 2] = cz;
            dirs[3 * row] = c.dx; dirs[3 * row + 1] = c.dy; dirs[3 * row + 2] = c.dz;
            *reinterpret_cast<float2*>(ts + 2 * row) = make_float2(t + dt, dt);
        }
        row0 += (size_t)__popcll(mask);
    }
}

template <bool WRITE>
__global__ void __launch_bounds__(64)
march_train_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d, const uint8_t* __restrict__ grid,
                   float bound, bool contract, float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                   const float* __restrict__ nears, const float* __restrict__ fars, float* __restrict__ xyzs,
                   float* __restrict__ dirs, float* __restrict__ ts, int32_t* __restrict__ rays,
                   const float* __restrict__ noises, uint32_t max_points) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    MarchCtx c;
    march_ctx_init(c, rays_o + 3 * (size_t)n, rays_d + 3 * (size_t)n, 0.0f, grid, bound, contract, dt_gamma, max_steps, C, H);
    uint32_t budget = max_steps;
    float *px = nullptr, *pd = nullptr, *pt = nullptr;
    if (WRITE) {
        const uint32_t off = (uint32_t)rays[2 * n];
        budget = (uint32_t)rays[2 * n + 1];
        if ((size_t)off + budget > (size_t)max_points) return;
        px = xyzs + 3 * (size_t)off; pd = dirs + 3 * (size_t)off; pt = ts + 2 * (size_t)off;
    }
    const float far = fars[n];
    float t = nears[n];
    t += n2m_clampf(t * dt_gamma, c.dt_min, c.dt_max) * noises[n];
    uint32_t kept = 0;
    MarchSample s;
    while (t < far && kept < budget) {
        if (!march_step(c, t, s)) continue;
        if (WRITE) {
            px[0] = s.cx; px[1] = s.cy; px[2] = s.cz;
            pd[0] = c.dx; pd[1] = c.dy; pd[2] = c.dz;
            *reinterpret_cast<float2*>(pt) = make_float2(s.t_after, s.dt);
            px += 3; pd += 3; pt += 2;
        }
        ++kept;
    }
    if (!WRITE) rays[2 * n + 1] = (int32_t)kept;
}

__global__ void __launch_bounds__(64)
march_infer_kernel(uint32_t n_alive, uint32_t n_step, const int32_t* __restrict__ rays_alive, const float* __restrict__ rays_t,
                   const float* __restrict__ rays_o, const float* __restrict__ rays_d, float bound, bool contract,
                   float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* __restrict__ grid,
                   const float* __restrict__ fars, float* __restrict__ xyzs, float* __restrict__ dirs,
                   float* __restrict__ ts, const float* __restrict__ noises) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
    const int32_t ray = rays_alive[n];
    MarchCtx c;
    march_ctx_init(c, rays_o + 3 * (size_t)ray, rays_d + 3 * (size_t)ray, 1e-10f, grid, bound, contract, dt_gamma, max_steps, C, H);
    float* px = xyzs + 3 * (size_t)n * n_step;
    float* pd = dirs + 3 * (size_t)n * n_step;
    float* pt = ts + 2 * (size_t)n * n_step;
    const float far = fars[ray];
    float t = rays_t[ray];
    t += n2m_clampf(t * dt_gamma, c.dt_min, c.dt_max) * noises[n];
    uint32_t kept = 0;
    MarchSample s;
    while (t < far && kept < n_step) {
        if (!march_step(c, t, s)) continue;
        px[0] = s.cx; px[1] = s.cy; px[2] = s.cz;
        pd[0] = c.dx; pd[1] = c.dy; pd[2] = c.dz;
        *reinterpret_cast<float2*>(pt) = make_float2(s.t_after, s.dt);
        px += 3; pd += 3; pt += 2;
        ++kept;
    }
}

// ---- inference loop with the ray count ON THE DEVICE (nerf/renderer.py:764-802 without its per-round host read of n_alive) ----
// state = {n_alive, step} (int32 x 2).  A round's kernels read the count from device memory and derive n_step exactly as the reference's
// host code does, n_step = max(min(N / n_alive, 8), 1) (:775); the host sizes its launches from an UPPER bound (n_alive never grows) that
// it learns a few rounds late through asynchronous copies, so nothing waits for a read-back.  A round whose state says "done"
// (n_alive == 0 or step >= max_steps) does nothing.
struct InferRound { uint32_t n_alive, n_step; bool active; };
__device__ __forceinline__ InferRound infer_round(const int32_t* __restrict__ state, uint32_t N, uint32_t max_steps) {
    InferRound r;
    r.n_alive = (uint32_t)state[0];
    const uint32_t step = (uint32_t)state[1];
    r.active = r.n_alive != 0u && step < max_steps;
    const uint32_t q = r.n_alive ? N / r.n_alive : 0u;
    r.n_step = max(min(q, 8u), 1u);
    return r;
}

__global__ void __launch_bounds__(64)
march_infer_dev_kernel(const int32_t* __restrict__ state, uint32_t N, const int32_t* __restrict__ rays_alive, const float* __restrict__ rays_t,
                       const float* __restrict__ rays_o, const float* __restrict__ rays_d, float bound, bool contract,
                       float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* __restrict__ grid,
                       const float* __restrict__ fars, float* __restrict__ xyzs, float* __restrict__ dirs,
                       float* __restrict__ ts, const float* __restrict__ noises) {
    const InferRound rd = infer_round(state, N, max_steps);
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (!rd.active || n >= rd.n_alive) return;
    const uint32_t n_step = rd.n_step;
    const int32_t ray = rays_alive[n];
    MarchCtx c;
    march_ctx_init(c, rays_o + 3 * (size_t)ray, rays_d + 3 * (size_t)ray, 1e-10f, grid, bound, contract, dt_gamma, max_steps, C, H);
    float* px = xyzs + 3 * (size_t)n * n_step;
    float* pd = dirs + 3 * (size_t)n * n_step;
    float* pt = ts + 2 * (size_t)n * n_step;
    const float far = fars[ray];
    float t = rays_t[ray];
    t += n2m_clampf(t * dt_gamma, c.dt_min, c.dt_max) * (noises ? noises[n] : 0.0f);
    uint32_t kept = 0;
    MarchSample s;
    while (t < far && kept < n_step) {
        if (!march_step(c, t, s)) continue;
        px[0] = s.cx; px[1] = s.cy; px[2] = s.cz;
        pd[0] = c.dx; pd[1] = c.dy; pd[2] = c.dz;
        *reinterpret_cast<float2*>(pt) = make_float2(s.t_after, s.dt);
        px += 3; pd += 3; pt += 2;
        ++kept;
    }
    // the reference's wrapper hands the kernel freshly zeroed buffers every round (raymarching.py:343-345): an unused slot has ts == 0,
    // which ends the ray in composite_rays (:877).  The buffers persist here, so the unused slots are cleared by their owner; positions and
    // directions of unused slots keep older (finite, in-range) values -- the field evaluates them, nobody reads the result.
    for (; kept < n_step; ++kept) { *reinterpret_cast<float2*>(pt) = make_float2(0.0f, 0.0f); pt += 2; }
}

// The same round with ONE WAVE PER ALIVE RAY (march_one_ray_core: 64 candidates at a time, the sample budget = n_step).  One lane per ray
// walks its ray alone, and a round lasts as long as its longest walk -- a ray that leaves the object and crosses empty space to `far` makes
// ~200 dependent occupancy loads -- whatever the number of rays still alive: 265 us per round, 28 rounds per 800 x 800 frame, measured.  A
// wave resolves 64 candidates per memory round trip, so a round of few rays is short; a round of MANY rays (the first ones: every pixel
// of the frame) is throughput-bound and stays with the lane-per-ray kernel (host-side switch in n2m_march_rays / n2m_march_rays_dev).
// Same samples, bit for bit: the wave marcher is the training marcher, which tests/ hold to the serial chain.
__device__ __forceinline__ void march_infer_wave_ray(uint32_t n, uint32_t n_step, int lane, const int32_t* __restrict__ rays_alive,
                                                     const float* __restrict__ rays_t, const float* __restrict__ rays_o,
                                                     const float* __restrict__ rays_d, float bound, bool contract, float dt_gamma,
                                                     uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* __restrict__ grid,
                                                     const float* __restrict__ fars, float* __restrict__ xyzs, float* __restrict__ dirs,
                                                     float* __restrict__ ts, const float* __restrict__ noises) {
    const int32_t ray = rays_alive[n];
    MarchCtx c;
    march_ctx_init(c, rays_o + 3 * (size_t)ray, rays_d + 3 * (size_t)ray, 1e-10f, grid, bound, contract, dt_gamma, max_steps, C, H);
    const float far = fars[ray];
    float t = rays_t[ray];
    t += n2m_clampf(t * dt_gamma, c.dt_min, c.dt_max) * (noises ? noises[n] : 0.0f);
    const size_t out = (size_t)n * n_step;
    const uint32_t kept = march_one_ray_core<true, false>(c, lane, far, t, n_step, out, xyzs, dirs, ts, false);
    // unused slots: ts == 0 ends the ray in composite_rays (see march_infer_dev_kernel)
    if ((uint32_t)lane < n_step - kept) *reinterpret_cast<float2*>(ts + 2 * (out + kept + (uint32_t)lane)) = make_float2(0.0f, 0.0f);
}

__global__ void __launch_bounds__(256)
march_infer_wave_kernel(uint32_t n_alive, uint32_t n_step, const int32_t* __restrict__ rays_alive, const float* __restrict__ rays_t,
                        const float* __restrict__ rays_o, const float* __restrict__ rays_d, float bound, bool contract,
                        float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* __restrict__ grid,
                        const float* __restrict__ fars, float* __restrict__ xyzs, float* __restrict__ dirs,
                        float* __restrict__ ts, const float* __restrict__ noises) {
    const uint32_t n = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (n >= n_alive) return;
    march_infer_wave_ray(n, n_step, (int)(threadIdx.x & 63u), rays_alive, rays_t, rays_o, rays_d, bound, contract, dt_gamma, max_steps, C, H, grid, fars,
                         xyzs, dirs, ts, noises);
}

__global__ void __launch_bounds__(256)
march_infer_wave_dev_kernel(const int32_t* __restrict__ state, uint32_t N, const int32_t* __restrict__ rays_alive, const float* __restrict__ rays_t,
                            const float* __restrict__ rays_o, const float* __restrict__ rays_d, float bound, bool contract,
                            float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* __restrict__ grid,
                            const float* __restrict__ fars, float* __restrict__ xyzs, float* __restrict__ dirs,
                            float* __restrict__ ts, const float* __restrict__ noises) {
    const InferRound rd = infer_round(state, N, max_steps);
    const uint32_t n = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (!rd.active || n >= rd.n_alive) return;
    march_infer_wave_ray(n, rd.n_step, (int)(threadIdx.x & 63u), rays_alive, rays_t, rays_o, rays_d, bound, contract, dt_gamma, max_steps, C, H, grid,
                         fars, xyzs, dirs, ts, noises);
}

// rays alive at or below which a round marches one wave per ray (N2M_INFER_WAVE: 0 = never, 1 = always, else the threshold itself)
static uint32_t infer_wave_threshold() {
    static const long v = getenv("N2M_INFER_WAVE") ? atol(getenv("N2M_INFER_WAVE")) : 131072;
    return v == 0 ? 0u : (v == 1 ? 0xFFFFFFFFu : (uint32_t)v);
}

// -------------------------------------------------------------------------------------- exclusive scans
// Generic exclusive scan over n items; Op supplies load(i), store(i, value, exclusive_prefix), base(), finish(total).

template <class Op>
__global__ void __launch_bounds__(1024) scan_single_block_kernel(Op op, uint32_t n) {
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t tile_tot;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    uint32_t carry = op.base();
    __syncthreads();   // every lane has read the base before lane 0 may overwrite it in finish()
    for (uint32_t start = 0; start < n; start += 1024) {
        const uint32_t i = start + tid;
        const uint32_t v = i < n ? op.load(i) : 0u;
        const uint32_t incl = n2m_wave_scan_add_u32(v, (int)lane);
        if (lane == 63) wave_tot[wid] = incl;
        __syncthreads();
        if (wid == 0) {
            const uint32_t tv = lane < 16 ? wave_tot[lane] : 0u;
            const uint32_t ti = n2m_wave_scan_add_u32(tv, (int)lane);
            if (lane < 16) wave_tot[lane] = ti - tv;
            if (lane == 15) tile_tot = ti;
        }
        __syncthreads();
        if (i < n) op.store(i, v, carry + wave_tot[wid] + incl - v);
        carry += tile_tot;
        __syncthreads();
    }
    if (tid == 0) op.finish(carry);
}

constexpr uint32_t kScanTile = 2048;   // items per 256-thread block in the 3-phase path (8 per lane)

template <class Op>
__global__ void __launch_bounds__(256) scan_block_sums_kernel(Op op, uint32_t n, uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t wave_tot[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t base = blockIdx.x * kScanTile;
    uint32_t acc = 0;
#pragma unroll
    for (uint32_t k = 0; k < kScanTile / 256; ++k) {
        const uint32_t i = base + k * 256 + tid;
        acc += i < n ? op.load(i) : 0u;
    }
    acc = n2m_wave_sum_u32(acc);
    if (lane == 0) wave_tot[wid] = acc;
    __syncthreads();
    if (tid == 0) block_sums[blockIdx.x] = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
}

// phase B of the 3-phase path: one block turns the per-block sums into exclusive prefixes that already contain
// op.base() (read here, by a single block, before anything is written back), total -> sums[n_blocks].
template <class Op>
struct BlockSumsOp {
    Op op;
    uint32_t* sums;
    uint32_t n_blocks;
    __device__ uint32_t base() const { return op.base(); }
    __device__ uint32_t load(uint32_t i) const { return sums[i]; }
    __device__ void store(uint32_t i, uint32_t, uint32_t excl) const { sums[i] = excl; }
    __device__ void finish(uint32_t total) const { sums[n_blocks] = total; }
};

template <class Op>
__global__ void __launch_bounds__(256) scan_apply_kernel(Op op, uint32_t n, const uint32_t* __restrict__ block_sums,
                                                         uint32_t n_blocks) {
    __shared__ uint32_t wave_tot[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t base = blockIdx.x * kScanTile;
    uint32_t carry = block_sums[blockIdx.x];
#pragma unroll 1
    for (uint32_t k = 0; k < kScanTile / 256; ++k) {
        const uint32_t i = base + k * 256 + tid;
        const uint32_t v = i < n ? op.load(i) : 0u;
        const uint32_t incl = n2m_wave_scan_add_u32(v, (int)lane);
        if (lane == 63) wave_tot[wid] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4; ++w) {
            const uint32_t x = wave_tot[w];
            if (w < wid) before += x;
            total += x;
        }
        if (i < n) op.store(i, v, carry + before + incl - v);
        carry += total;
        __syncthreads();
    }
    if (blockIdx.x == 0 && tid == 0) op.finish(block_sums[n_blocks]);
}

template <class Op>
int run_exclusive_scan(Op op, uint32_t n, hipStream_t stream) {
    if (n <= 131072u) {
        scan_single_block_kernel<Op><<<1, 1024, 0, stream>>>(op, n);
        return 0;
    }
    const uint32_t n_blocks = n2m_ceil_div(n, kScanTile);
    uint32_t* sums = nullptr;
    hipError_t e = hipMallocAsync(reinterpret_cast<void**>(&sums), sizeof(uint32_t) * (n_blocks + 1), stream);
    if (e != hipSuccess) return (int)e;
    scan_block_sums_kernel<Op><<<n_blocks, 256, 0, stream>>>(op, n, sums);
    scan_single_block_kernel<BlockSumsOp<Op>><<<1, 1024, 0, stream>>>(BlockSumsOp<Op>{op, sums, n_blocks}, n_blocks);
    scan_apply_kernel<Op><<<n_blocks, 256, 0, stream>>>(op, n, sums, n_blocks);
    e = hipFreeAsync(sums, stream);
    return (int)e;
}

struct RayOffsetsOp {   // rays[n] = (offset, count): offsets from counts, counter[0] += total
    int32_t* rays;
    int32_t* counter;
    __device__ uint32_t base() const { return (uint32_t)counter[0]; }
    __device__ uint32_t load(uint32_t i) const { return (uint32_t)rays[2 * i + 1]; }
    __device__ void store(uint32_t i, uint32_t, uint32_t excl) const { rays[2 * i] = (int32_t)excl; }
    __device__ void finish(uint32_t total) const { counter[0] = (int32_t)total; }
};

struct CompactOp {
    const int32_t* in;
    int32_t* out;
    int32_t* n_out;
    __device__ uint32_t base() const { return 0u; }
    __device__ uint32_t load(uint32_t i) const { return in[i] >= 0 ? 1u : 0u; }
    __device__ void store(uint32_t i, uint32_t v, uint32_t excl) const { if (v) out[excl] = in[i]; }
    __device__ void finish(uint32_t total) const { n_out[0] = (int32_t)total; }
};

struct SelectPositiveOp {  // out[k] = index of the k-th item whose value is > 0 (ascending), n_out = their number: torch.nonzero(v > 0) without its host read
    const float* values;
    uint32_t stride;
    int64_t* out;
    int32_t* n_out;
    __device__ uint32_t base() const { return 0u; }
    __device__ uint32_t load(uint32_t i) const { return values[(size_t)i * stride] > 0.0f ? 1u : 0u; }
    __device__ void store(uint32_t i, uint32_t v, uint32_t excl) const { if (v) out[excl] = (int64_t)i; }
    __device__ void finish(uint32_t total) const { n_out[0] = (int32_t)total; }
};

struct CompactDevOp {      // the same with the item count in device memory; writes the NEXT round's state {n_alive, step + n_step}
    const int32_t* in;
    int32_t* out;
    const int32_t* state;
    int32_t* state_out;
    uint32_t N, max_steps;
    __device__ uint32_t base() const { return 0u; }
    __device__ uint32_t load(uint32_t i) const {
        const InferRound rd = infer_round(state, N, max_steps);
        return (rd.active && i < rd.n_alive && in[i] >= 0) ? 1u : 0u;
    }
    __device__ void store(uint32_t i, uint32_t v, uint32_t excl) const { if (v) out[excl] = in[i]; }
    __device__ void finish(uint32_t total) const {
        const InferRound rd = infer_round(state, N, max_steps);
        state_out[0] = rd.active ? (int32_t)total : 0;
        state_out[1] = state[1] + (rd.active ? (int32_t)rd.n_step : 0);
    }
};

// ------------------------------------------------------------------------------------ compositing (train)

// 4 waves per workgroup, one ray per wave.
__global__ void __launch_bounds__(256)
composite_train_fwd_kernel(const float* __restrict__ sigmas, const float* __restrict__ rgbs, const float* __restrict__ ts,
                           const int32_t* __restrict__ rays, uint32_t M, uint32_t N, float T_thresh, bool alpha_mode,
                           float* __restrict__ weights, float* __restrict__ weights_sum, float* __restrict__ depth,
                           float* __restrict__ image) {
    const uint32_t n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (n >= N) return;
    const uint32_t off = (uint32_t)rays[2 * n], cnt = (uint32_t)rays[2 * n + 1];
    float r = 0, g = 0, b = 0, ws = 0, d = 0;
    // Every sample of the ray's range that lies inside [0, M) is WRITTEN (zero after the early stop; a ray cut off by M -- the
    // reference's kernel returns for it, raymarching.cu:513 -- zeroes its part), so a caller whose rays tile [0, M), as those of
    // march_rays_train do, need not clear `weights` first.
    if (cnt != 0 && off + cnt <= M) {
        float carry_T = 1.0f;
        bool stopped = false;
        for (uint32_t base = 0; base < cnt; base += 64) {
            const uint32_t k = base + lane;
            const bool valid = k < cnt;
            const size_t i = (size_t)off + k;
            if (stopped) {
                if (valid) weights[i] = 0.f;
                continue;
            }
            float alpha = 0.f, tmid = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
            if (valid) {
                const float2 tt = *reinterpret_cast<const float2*>(ts + 2 * i);
                const float sg = sigmas[i];
                alpha = alpha_mode ? sg : (1.0f - expf(-sg * tt.y));
                tmid = tt.x;
                cr = rgbs[3 * i]; cg = rgbs[3 * i + 1]; cb = rgbs[3 * i + 2];
            }
            const float incl = n2m_wave_scan_mul(1.0f - alpha, lane);
            const float excl = n2m_lane_below(incl, 1.0f);
            const float T_before = carry_T * excl, T_after = carry_T * incl;
            // the sample that drives T below the threshold is still composited; everything after it is not
            const unsigned long long stop = __ballot(valid && T_after < T_thresh);
            const int last = stop ? (int)__ffsll((long long)stop) - 1 : 63;
            const bool live = valid && lane <= last;
            const float w = live ? alpha * T_before : 0.f;
            if (valid) weights[i] = w;
            r += w * cr; g += w * cg; b += w * cb; ws += w; d += w * tmid;
            if (stop) stopped = true;
            else carry_T = n2m_lane63(T_after);
        }
        r = n2m_wave_sum(r); g = n2m_wave_sum(g); b = n2m_wave_sum(b); ws = n2m_wave_sum(ws); d = n2m_wave_sum(d);
    } else if (cnt != 0 && off < M) {
        for (uint32_t i = off + (uint32_t)lane; i < M; i += 64) weights[i] = 0.f;
    }
    if (lane == 0) {
        weights_sum[n] = ws;
        depth[n] = d;
        image[3 * n] = r; image[3 * n + 1] = g; image[3 * n + 2] = b;
    }
}

__global__ void __launch_bounds__(256)
composite_train_bwd_kernel(const float* __restrict__ grad_weights, const float* __restrict__ grad_weights_sum,
                           const float* __restrict__ grad_depth, const float* __restrict__ grad_image,
                           const float* __restrict__ sigmas, const float* __restrict__ rgbs, const float* __restrict__ ts,
                           const int32_t* __restrict__ rays, const float* __restrict__ weights_sum,
                           const float* __restrict__ depth, const float* __restrict__ image, uint32_t M, uint32_t N,
                           float T_thresh, bool alpha_mode, float* __restrict__ grad_sigmas, float* __restrict__ grad_rgbs) {
    const uint32_t n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (n >= N) return;
    const uint32_t off = (uint32_t)rays[2 * n], cnt = (uint32_t)rays[2 * n + 1];
    if (cnt == 0) return;
    if (off + cnt > M) {          // cut off by M: no gradient; like the forward, the part inside [0, M) is written (zeros)
        for (uint32_t i = off + (uint32_t)lane; i < M; i += 64) {
            grad_sigmas[i] = 0.f;
            grad_rgbs[3 * (size_t)i] = 0.f; grad_rgbs[3 * (size_t)i + 1] = 0.f; grad_rgbs[3 * (size_t)i + 2] = 0.f;
        }
        return;
    }
    const float gi0 = grad_image[3 * n], gi1 = grad_image[3 * n + 1], gi2 = grad_image[3 * n + 2];
    const float gws = grad_weights_sum[n], gd = grad_depth[n];
    const float rF = image[3 * n], gF = image[3 * n + 1], bF = image[3 * n + 2], wsF = weights_sum[n], dF = depth[n];
    float carry_T = 1.0f, r0 = 0, g0 = 0, b0 = 0, ws0 = 0, d0 = 0;   // running sums before this chunk
    bool stopped = false;
    for (uint32_t base = 0; base < cnt; base += 64) {
        const uint32_t k = base + lane;
        const bool valid = k < cnt;
        const size_t i = (size_t)off + k;
        if (stopped) {
            if (valid) { grad_sigmas[i] = 0.f; grad_rgbs[3 * i] = 0.f; grad_rgbs[3 * i + 1] = 0.f; grad_rgbs[3 * i + 2] = 0.f; }
            continue;
        }
        float alpha = 0.f, tmid = 0.f, dt = 0.f, cr = 0.f, cg = 0.f, cb = 0.f, gw = 0.f;
        if (valid) {
            const float2 tt = *reinterpret_cast<const float2*>(ts + 2 * i);
            const float sg = sigmas[i];
            alpha = alpha_mode ? sg : (1.0f - expf(-sg * tt.y));
            tmid = tt.x; dt = tt.y;
            cr = rgbs[3 * i]; cg = rgbs[3 * i + 1]; cb = rgbs[3 * i + 2];
            gw = grad_weights[i];
        }
        const float incl = n2m_wave_scan_mul(1.0f - alpha, lane);
        const float excl = n2m_lane_below(incl, 1.0f);
        const float T_before = carry_T * excl, T_after = carry_T * incl;
        const unsigned long long stop = __ballot(valid && T_after < T_thresh);
        const int last = stop ? (int)__ffsll((long long)stop) - 1 : 63;
        const bool live = valid && lane <= last;
        const float w = live ? alpha * T_before : 0.f;
        // inclusive running sums up to and including this sample
        const float r = r0 + n2m_wave_scan_add(w * cr, lane);
        const float g = g0 + n2m_wave_scan_add(w * cg, lane);
        const float b = b0 + n2m_wave_scan_add(w * cb, lane);
        const float ws = ws0 + n2m_wave_scan_add(w, lane);
        const float d = d0 + n2m_wave_scan_add(w * tmid, lane);
        if (live) {
            grad_rgbs[3 * i] = gi0 * w; grad_rgbs[3 * i + 1] = gi1 * w; grad_rgbs[3 * i + 2] = gi2 * w;
            const float scale = alpha_mode ? (1.0f / (1.0f - alpha)) : dt;
            grad_sigmas[i] = scale * (gi0 * (T_after * cr - (rF - r)) + gi1 * (T_after * cg - (gF - g)) +
                                      gi2 * (T_after * cb - (bF - b)) + (gws + gw) * (T_after - (wsF - ws)) +
                                      gd * (T_after * tmid - (dF - d)));
        } else if (valid) {
            grad_sigmas[i] = 0.f; grad_rgbs[3 * i] = 0.f; grad_rgbs[3 * i + 1] = 0.f; grad_rgbs[3 * i + 2] = 0.f;
        }
        if (stop) { stopped = true; continue; }
        carry_T = n2m_lane63(T_after);
        r0 = n2m_lane63(r); g0 = n2m_lane63(g); b0 = n2m_lane63(b);
        ws0 = n2m_lane63(ws); d0 = n2m_lane63(d);
    }
}

// Compositing + photometric loss head + their backward in ONE launch (training fast path, nerf2mesh_amd/engine.py).  The loss of a ray
// needs only that ray's composited colour and opacity, and its seed gradient (the loss scale / N) does not depend on the loss VALUE,
// so a wave can composite its ray forward (composite_train_fwd_kernel's arithmetic, sample weights not stored), form the ray's loss
// term and gradients (photo_loss_forward/backward_kernel's arithmetic: background blend nerf/renderer.py:747, target nerf/utils.py:
// 663-664, MSE + mask MSE :679-683) and run the backward scan (composite_train_bwd_kernel's arithmetic) while the ray's samples are
// still in cache.  Replaces four launches and the [N]-sized round trips between them; gradients are bit-identical to the four-kernel
// chain, the loss value differs by its summation order only (per-workgroup partials, summed in index order by the last workgroup).
// 16 waves per workgroup, one ray per wave: same-address atomics retire one per ~11 ns (tools/atomic_bench.hip), so the arrival ticket
// must be taken by ~N/16 workgroups, not N/4 (measured: 80 us with 4-ray workgroups at N = 15 k, the ticket chain alone)
// ENT: + the entropy regulariser of nerf/utils.py:728-733, lambda_entropy * (mean_m H(clamp(w_m)) + mean_n H(clamp(ws_n))) with
// H(p) = -p log2 p - (1-p) log2(1-p) and clamp to [1e-5, 1 - 1e-5] -- the one loss term that hands composite_rays_train's backward a
// non-zero grad_weights (config 4, scripts/runall_360_outdoor.sh:2).  A sample's H'(w) enters the backward exactly where the reference
// kernel reads grad_weights[i] (raymarching.cu:676, its per-sample factor on the ray-suffix term, kept as it is).
__device__ __forceinline__ float n2m_entropy(float p) { return -p * log2f(p) - (1.0f - p) * log2f(1.0f - p); }
__device__ __forceinline__ float n2m_entropy_grad(float x) {       // d H(clamp(x)) / dx: clamp passes the gradient on [1e-5, 1 - 1e-5] inclusive
    const float lo = 1e-5f, hi = 1.0f - 1e-5f;
    return (x >= lo && x <= hi) ? (-log2f(x) - 1.4426950408889634f) + (log2f(1.0f - x) + 1.4426950408889634f) : 0.0f;
}
// ALPHA: the SDF recipe's alpha mode (raymarching.cu:534,671: alpha = the `sigmas` input itself, backward scale 1 / (1 - alpha))
template <bool ENT, bool ALPHA = false>
__global__ void __launch_bounds__(1024)
composite_loss_train_kernel(const float* __restrict__ sigmas, const float* __restrict__ rgbs, const float* __restrict__ ts,
                            const int32_t* __restrict__ rays, uint32_t M, uint32_t N, float T_thresh, const float* __restrict__ gt,
                            const float* __restrict__ bg, float bg_scalar, float lambda_rgb, float lambda_mask,
                            const float* __restrict__ grad_loss, float* __restrict__ weights_sum, float* __restrict__ image,
                            float* __restrict__ grad_sigmas, float* __restrict__ grad_rgbs, float* __restrict__ partial,
                            uint32_t* __restrict__ ticket, float* __restrict__ loss, float* __restrict__ loss_sum, float lambda_entropy,
                            int32_t* __restrict__ live_out, uint32_t* __restrict__ block_live_out) {
    __shared__ float wave_loss[16];
    __shared__ uint32_t wave_live[16];
    __shared__ bool last_block;
    const uint32_t wid = threadIdx.x >> 6, n = blockIdx.x * 16 + wid;
    const int lane = threadIdx.x & 63;
    float l_ray = 0.0f;
    uint32_t n_live = 0u;                  // samples of this ray up to and including the one the early stop fell on: the others get zero gradients
    if (n < N) {
        const uint32_t off = (uint32_t)rays[2 * n], cnt = (uint32_t)rays[2 * n + 1];
        const bool whole = cnt != 0 && off + cnt <= M;
        // everything per-ray the loss head needs is requested now, not after the forward loop (a wave owns its ray alone: every
        // dependent global round trip is exposed -- the kernel cost 15 us for 110 rays and 18 us for 17 000)
        const float4 gtv = *reinterpret_cast<const float4*>(gt + (size_t)n * 4);
        const float gl = *grad_loss;
        float bgv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) bgv[c] = bg ? bg[(size_t)n * 3 + c] : bg_scalar;
        // ---- forward
        float rF = 0, gF = 0, bF = 0, wsF = 0;
        float a0 = 0.f, dt0 = 0.f, cr0 = 0.f, cg0 = 0.f, cb0 = 0.f;     // the first 64 samples stay in registers for the backward pass
        float entF = 0.f;                  // ENT: sum of H(clamp(w)) over this ray's samples (weights after the early stop are zeros)
        uint32_t visited = 0;
        if (whole) {
            float carry_T = 1.0f;
            n_live = cnt;
            for (uint32_t base = 0; base < cnt; base += 64) {
                const uint32_t k = base + lane;
                const bool valid = k < cnt;
                const size_t i = (size_t)off + k;
                float alpha = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
                if (valid) {
                    const float2 tt = *reinterpret_cast<const float2*>(ts + 2 * i);
                    alpha = ALPHA ? sigmas[i] : 1.0f - expf(-sigmas[i] * tt.y);
                    cr = rgbs[3 * i]; cg = rgbs[3 * i + 1]; cb = rgbs[3 * i + 2];
                    if (base == 0) { a0 = alpha; dt0 = tt.y; cr0 = cr; cg0 = cg; cb0 = cb; }
                }
                const float incl = n2m_wave_scan_mul(1.0f - alpha, lane);
                const float excl = n2m_lane_below(incl, 1.0f);
                const float T_before = carry_T * excl, T_after = carry_T * incl;
                const unsigned long long stop = __ballot(valid && T_after < T_thresh);
                const int last = stop ? (int)__ffsll((long long)stop) - 1 : 63;
                const float w = (valid && lane <= last) ? alpha * T_before : 0.f;
                rF += w * cr; gF += w * cg; bF += w * cb; wsF += w;
                if (ENT) {
                    if (valid) entF += n2m_entropy(fminf(fmaxf(w, 1e-5f), 1.0f - 1e-5f));
                    visited = min(cnt, base + 64u);
                }
                if (stop) { n_live = base + (uint32_t)last + 1u; break; }
                carry_T = n2m_lane63(T_after);
            }
            rF = n2m_wave_sum(rF); gF = n2m_wave_sum(gF); bF = n2m_wave_sum(bF); wsF = n2m_wave_sum(wsF);
            if (ENT) entF = n2m_wave_sum(entF) + (float)(cnt - visited) * n2m_entropy(1e-5f);
        } else if (ENT && cnt != 0 && off < M) entF = (float)(M - off) * n2m_entropy(1e-5f);      // cut off by M: its weights stay zero
        // ---- loss term of the ray and its gradients (wave-uniform)
        const float a = gtv.w;
        const float gc[3] = {gtv.x, gtv.y, gtv.z}, pc[3] = {rF, gF, bF};
        const float gscale = gl / (float)N;
        float gi[3];
        const float m = wsF - a;
        float gws = gscale * lambda_mask * 2.0f * m;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float bgc = bgv[c];
            const float target = gc[c] * a + bgc * (1.0f - a);
            const float pred = pc[c] + (1.0f - wsF) * bgc;
            const float e = pred - target;
            gi[c] = gscale * lambda_rgb * (2.0f * e / 3.0f);
            gws -= gi[c] * bgc;
            if (c == 0) l_ray = e * e; else l_ray += e * e;
        }
        l_ray = lambda_rgb * (l_ray / 3.0f) + lambda_mask * (m * m);
        float gE = 0.f;                     // ENT: seed gradient of one sample's entropy term, lambda / M
        if (ENT) {
            const float inv_m = M ? 1.0f / (float)M : 0.0f;
            l_ray += lambda_entropy * (n2m_entropy(fminf(fmaxf(wsF, 1e-5f), 1.0f - 1e-5f)) + entF * ((float)N * inv_m));
            gws += gscale * lambda_entropy * n2m_entropy_grad(wsF);
            gE = gl * lambda_entropy * inv_m;
        }
        if (lane == 0) {
            if (weights_sum) weights_sum[n] = wsF;
            if (image) { image[3 * n] = rF; image[3 * n + 1] = gF; image[3 * n + 2] = bF; }
            if (live_out) live_out[n] = (int32_t)n_live;
        }
        // ---- backward
        if (cnt != 0 && !whole) {                       // cut off by M: no gradient, the part inside [0, M) is zeroed
            for (uint32_t i = off + (uint32_t)lane; i < M; i += 64) {
                grad_sigmas[i] = 0.f;
                grad_rgbs[3 * (size_t)i] = 0.f; grad_rgbs[3 * (size_t)i + 1] = 0.f; grad_rgbs[3 * (size_t)i + 2] = 0.f;
            }
        } else if (whole) {
            float carry_T = 1.0f, r0 = 0, g0 = 0, b0 = 0, ws0 = 0;
            bool stopped = false;
            for (uint32_t base = 0; base < cnt; base += 64) {
                const uint32_t k = base + lane;
                const bool valid = k < cnt;
                const size_t i = (size_t)off + k;
                if (stopped) {
                    if (valid) { grad_sigmas[i] = 0.f; grad_rgbs[3 * i] = 0.f; grad_rgbs[3 * i + 1] = 0.f; grad_rgbs[3 * i + 2] = 0.f; }
                    continue;
                }
                float alpha = 0.f, dt = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
                if (base == 0) { alpha = a0; dt = dt0; cr = cr0; cg = cg0; cb = cb0; }      // the forward pass's own values (zeros where !valid)
                else if (valid) {
                    const float2 tt = *reinterpret_cast<const float2*>(ts + 2 * i);
                    alpha = ALPHA ? sigmas[i] : 1.0f - expf(-sigmas[i] * tt.y);
                    dt = tt.y;
                    cr = rgbs[3 * i]; cg = rgbs[3 * i + 1]; cb = rgbs[3 * i + 2];
                }
                const float incl = n2m_wave_scan_mul(1.0f - alpha, lane);
                const float excl = n2m_lane_below(incl, 1.0f);
                const float T_before = carry_T * excl, T_after = carry_T * incl;
                const unsigned long long stop = __ballot(valid && T_after < T_thresh);
                const int last = stop ? (int)__ffsll((long long)stop) - 1 : 63;
                const bool live = valid && lane <= last;
                const float w = live ? alpha * T_before : 0.f;
                const float r = r0 + n2m_wave_scan_add(w * cr, lane);
                const float g = g0 + n2m_wave_scan_add(w * cg, lane);
                const float b = b0 + n2m_wave_scan_add(w * cb, lane);
                const float ws = ws0 + n2m_wave_scan_add(w, lane);
                if (live) {
                    grad_rgbs[3 * i] = gi[0] * w; grad_rgbs[3 * i + 1] = gi[1] * w; grad_rgbs[3 * i + 2] = gi[2] * w;
                    // composite_train_bwd_kernel's expression with grad_weights = grad_depth = 0 (their terms are exact zeros there)
                    const float gw = ENT ? gE * n2m_entropy_grad(w) : 0.f;      // grad_weights[i]
                    const float gscl = ALPHA ? 1.0f / (1.0f - alpha) : dt;      // (alpha = 1: inf, un-guarded like the reference; the caller clips alpha)
                    grad_sigmas[i] = gscl * (gi[0] * (T_after * cr - (rF - r)) + gi[1] * (T_after * cg - (gF - g)) +
                                           gi[2] * (T_after * cb - (bF - b)) + (gws + gw) * (T_after - (wsF - ws)) + 0.f * 0.f);
                } else if (valid) {
                    grad_sigmas[i] = 0.f; grad_rgbs[3 * i] = 0.f; grad_rgbs[3 * i + 1] = 0.f; grad_rgbs[3 * i + 2] = 0.f;
                }
                if (stop) { stopped = true; continue; }
                carry_T = n2m_lane63(T_after);
                r0 = n2m_lane63(r); g0 = n2m_lane63(g); b0 = n2m_lane63(b); ws0 = n2m_lane63(ws);
            }
        }
    }
    // ---- loss value: per-workgroup partial, the last workgroup to arrive sums them in index order (reproducible)
    if (lane == 0) { wave_loss[wid] = l_ray; wave_live[wid] = n_live; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float p = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) p += wave_loss[q];
        partial[blockIdx.x] = p;
        if (block_live_out) {
            uint32_t c = 0u;
#pragma unroll
            for (int q = 0; q < 16; ++q) c += wave_live[q];
            block_live_out[blockIdx.x] = c;
        }
        last_block = false;
        if (ticket) {           // ticket == NULL: the caller reduces the partials itself (n2m_scaler_update_slots_loss)
            __threadfence();
            last_block = atomicAdd(ticket, 1u) == gridDim.x - 1;
        }
    }
    if (ticket == nullptr) return;
    __syncthreads();
    if (last_block) {
        __threadfence();
        float sum = 0.0f;
        for (uint32_t i = threadIdx.x; i < gridDim.x; i += 1024) sum += __hip_atomic_load(&partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sum = n2m_wave_sum(sum);
        __syncthreads();
        if (lane == 0) wave_loss[wid] = sum;
        __syncthreads();
        if (threadIdx.x == 0) {
            float tot = 0.0f;
#pragma unroll
            for (int q = 0; q < 16; ++q) tot += wave_loss[q];
            const float v = tot / (float)N;
            *loss = v;
            if (loss_sum) *loss_sum += v;
            *ticket = 0;
        }
    }
}

// ----------------------------------------------------------------------------------- compositing (inference)

__global__ void composite_infer_kernel(uint32_t n_alive, uint32_t n_step, float T_thresh, bool alpha_mode,
                                       int32_t* __restrict__ rays_alive, float* __restrict__ rays_t,
                                       const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                       const float* __restrict__ ts, float* __restrict__ weights_sum,
                                       float* __restrict__ depth, float* __restrict__ image) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
    const int32_t ray = rays_alive[n];
    const size_t base = (size_t)n * n_step;
    float t = 0.0f;
    float d = depth[ray], r = image[3 * ray], g = image[3 * ray + 1], b = image[3 * ray + 2], ws = weights_sum[ray];
    uint32_t step = 0;
    while (step < n_step) {
        const size_t i = base + step;
        const float2 tt = *reinterpret_cast<const float2*>(ts + 2 * i);
        if (tt.x == 0) break;
        const float alpha = alpha_mode ? sigmas[i] : (1.0f - expf(-sigmas[i] * tt.y));
        const float T = 1 - ws;
        const float w = alpha * T;
        ws += w;
        t = tt.x;
        d += w * t;
        r += w * rgbs[3 * i]; g += w * rgbs[3 * i + 1]; b += w * rgbs[3 * i + 2];
        if (T < T_thresh) break;
        ++step;
    }
    if (step < n_step) rays_alive[n] = -1; else rays_t[ray] = t;
    weights_sum[ray] = ws;
    depth[ray] = d;
    image[3 * ray] = r; image[3 * ray + 1] = g; image[3 * ray + 2] = b;
}

__global__ void composite_infer_dev_kernel(const int32_t* __restrict__ state, uint32_t N, uint32_t max_steps, float T_thresh, bool alpha_mode,
                                           int32_t* __restrict__ rays_alive, float* __restrict__ rays_t,
                                           const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                           const float* __restrict__ ts, float* __restrict__ weights_sum,
                                           float* __restrict__ depth, float* __restrict__ image) {
    const InferRound rd = infer_round(state, N, max_steps);
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (!rd.active || n >= rd.n_alive) return;
    const uint32_t n_step = rd.n_step;
    const int32_t ray = rays_alive[n];
    const size_t base = (size_t)n * n_step;
    float t = 0.0f;
    float d = depth[ray], r = image[3 * ray], g = image[3 * ray + 1], b = image[3 * ray + 2], ws = weights_sum[ray];
    uint32_t step = 0;
    while (step < n_step) {
        const size_t i = base + step;
        const float2 tt = *reinterpret_cast<const float2*>(ts + 2 * i);
        if (tt.x == 0) break;
        const float alpha = alpha_mode ? sigmas[i] : (1.0f - expf(-sigmas[i] * tt.y));
        const float T = 1 - ws;
        const float w = alpha * T;
        ws += w;
        t = tt.x;
        d += w * t;
        r += w * rgbs[3 * i]; g += w * rgbs[3 * i + 1]; b += w * rgbs[3 * i + 2];
        if (T < T_thresh) break;
        ++step;
    }
    if (step < n_step) rays_alive[n] = -1; else rays_t[ray] = t;
    weights_sum[ray] = ws;
    depth[ray] = d;
    image[3 * ray] = r; image[3 * ray + 1] = g; image[3 * ray + 2] = b;
}

bool serial_march() {   // A/B switch: N2M_MARCH_SERIAL=1 selects the one-ray-per-lane kernels
    static const bool v = getenv("N2M_MARCH_SERIAL") != nullptr;
    return v;
}

}  // namespace

// ================================================================================================ C ABI

extern "C" int n2m_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N,
                                      float min_near, float* nears, float* fars, void* stream) {
    N2M_NOTNULL(rays_o); N2M_NOTNULL(rays_d); N2M_NOTNULL(aabb); N2M_NOTNULL(nears); N2M_NOTNULL(fars);
    if (N == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    N2M_PROF(N2M_K_NEAR_FAR, s, 32.0 * N);
    near_far_kernel<<<n2m_ceil_div(N, 256), 256, 0, s>>>(rays_o, rays_d, aabb, N, min_near, nears, fars);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_batch_rays_cnf(const float* poses, const float* uniforms, uint32_t V, uint32_t N, uint32_t H, uint32_t W, float fx, float fy,
                              float cx, float cy, const float* images, const float* aabb, float min_near, float* rays_o, float* rays_d,
                              float* rgba, float* nears, float* fars, float* noises, float* bg, int32_t* counter, const float* cam_near_far,
                                  void* stream) {
    N2M_NOTNULL(poses); N2M_NOTNULL(uniforms); N2M_NOTNULL(images); N2M_NOTNULL(aabb); N2M_NOTNULL(rays_o); N2M_NOTNULL(rays_d);
    N2M_NOTNULL(rgba); N2M_NOTNULL(nears); N2M_NOTNULL(fars); N2M_NOTNULL(noises);
    N2M_REQUIRE(V >= 1 && (uint64_t)H * W < (1ull << 24), N2M_EINVAL, "batch_rays: need V >= 1 and H*W < 2^24 (pixel index from an fp32 uniform)");
    if (N == 0) return 0;
    batch_rays_kernel<<<n2m_ceil_div(N, 256), 256, 0, (hipStream_t)stream>>>(poses, uniforms, V, N, W, H * W, fx, fy, cx, cy, images, aabb, min_near,
                                                                             rays_o, rays_d, rgba, nears, fars, noises, bg, counter, cam_near_far);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_batch_rays(const float* poses, const float* uniforms, uint32_t V, uint32_t N, uint32_t H, uint32_t W, float fx, float fy,
                              float cx, float cy, const float* images, const float* aabb, float min_near, float* rays_o, float* rays_d,
                              float* rgba, float* nears, float* fars, float* noises, float* bg, int32_t* counter, void* stream) {
    return n2m_batch_rays_cnf(poses, uniforms, V, N, H, W, fx, fy, cx, cy, images, aabb, min_near, rays_o, rays_d, rgba, nears, fars, noises, bg,
                              counter, nullptr, stream);
}

extern "C" int n2m_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords,
                                void* stream) {
    N2M_NOTNULL(rays_o); N2M_NOTNULL(rays_d); N2M_NOTNULL(coords);
    if (N == 0) return 0;
    sph_from_ray_kernel<<<n2m_ceil_div(N, 256), 256, 0, (hipStream_t)stream>>>(rays_o, rays_d, radius, N, coords);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, void* stream) {
    N2M_NOTNULL(coords); N2M_NOTNULL(indices);
    if (N == 0) return 0;
    morton_kernel<<<n2m_ceil_div(N, 256), 256, 0, (hipStream_t)stream>>>(coords, N, indices);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords, void* stream) {
    N2M_NOTNULL(coords); N2M_NOTNULL(indices);
    if (N == 0) return 0;
    morton_invert_kernel<<<n2m_ceil_div(N, 256), 256, 0, (hipStream_t)stream>>>(indices, N, coords);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield, void* stream) {
    N2M_NOTNULL(grid); N2M_NOTNULL(bitfield);
    if (N == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    N2M_PROF(N2M_K_PACKBITS, s, 33.0 * N);
    const bool aligned = (((uintptr_t)grid & 15u) == 0) && (((uintptr_t)bitfield & 3u) == 0);
    if (aligned) packbits_kernel<<<n2m_ceil_div((uint64_t)(N >> 2) + 1, 256), 256, 0, s>>>(grid, N, density_thresh, bitfield);
    else packbits_bytes_kernel<<<n2m_ceil_div(N, 256), 256, 0, s>>>(grid, N, density_thresh, bitfield);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_packbits_dev(const float* grid, uint32_t N, const float* density_thresh, uint8_t* bitfield, void* stream) {
    N2M_NOTNULL(grid); N2M_NOTNULL(bitfield); N2M_NOTNULL(density_thresh);
    if (N == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    N2M_PROF(N2M_K_PACKBITS, s, 33.0 * N);
    const bool aligned = (((uintptr_t)grid & 15u) == 0) && (((uintptr_t)bitfield & 3u) == 0);
    if (aligned) packbits_kernel<<<n2m_ceil_div((uint64_t)(N >> 2) + 1, 256), 256, 0, s>>>(grid, N, 0.f, bitfield, density_thresh);
    else packbits_bytes_kernel<<<n2m_ceil_div(N, 256), 256, 0, s>>>(grid, N, 0.f, bitfield, density_thresh);
    N2M_CHECK_LAUNCH();
    return 0;
}

// ---- occupancy refresh (nerf/renderer.py:1074-1149), the elementwise work around the density query as two launches
// points: xyz = cell * (bound - hgs) + (u * 2 - 1) * hgs, the reference's expression (:1096-1100) with its rounding points (no contraction);
// idx != NULL: only the listed cells (point j <- cell idx[j]: the cells whose grid value is >= 0 -- the others are never updated, :1131-1134)
__global__ void __launch_bounds__(256)
occupancy_points_kernel(const float* __restrict__ cells, const float* __restrict__ u, const int32_t* __restrict__ idx, float inner, float hgs,
                        float* __restrict__ xyz, uint32_t n) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= n) return;
    const size_t c = idx ? (size_t)idx[j] : (size_t)j;
#pragma unroll
    for (uint32_t a = 0; a < 3; ++a) {
        const float jitter = (u[c * 3u + a] * 2.0f - 1.0f) * hgs;
        xyz[(size_t)j * 3u + a] = cells[c * 3u + a] * inner + jitter;
    }
}

// update: grid = max(grid * decay, tmp) where both are >= 0 (:1133-1134); sum of max(grid, 0) -> mean (:1136) -> threshold
// min(mean, density_thresh) (:1140), all on the device: the last workgroup to finish adds the per-workgroup sums in a fixed order
__global__ void __launch_bounds__(256)
occupancy_update_kernel(float* __restrict__ grid, const float* __restrict__ tmp, float decay, uint32_t n, float density_thresh,
                        float* __restrict__ partials, uint32_t* __restrict__ ticket, float* __restrict__ mean_out, float* __restrict__ thresh_out) {
    __shared__ float red[4];
    __shared__ bool last;
    float acc = 0.0f;
    // (a few hundred workgroups, grid-stride: each ends with one same-address atomic, and those serialise at ~30 ns apiece across the XCDs --
    // one workgroup per 1024 cells spent 60 of its 68 us there)
    for (uint32_t i0 = (blockIdx.x * 256u + threadIdx.x) * 4u; i0 < n; i0 += gridDim.x * 1024u) {
        if (i0 + 4u <= n) {
            float4 g = *reinterpret_cast<const float4*>(grid + i0);
            const float4 t = *reinterpret_cast<const float4*>(tmp + i0);
            if (g.x >= 0.0f && t.x >= 0.0f) g.x = fmaxf(g.x * decay, t.x);
            if (g.y >= 0.0f && t.y >= 0.0f) g.y = fmaxf(g.y * decay, t.y);
            if (g.z >= 0.0f && t.z >= 0.0f) g.z = fmaxf(g.z * decay, t.z);
            if (g.w >= 0.0f && t.w >= 0.0f) g.w = fmaxf(g.w * decay, t.w);
            *reinterpret_cast<float4*>(grid + i0) = g;
            acc += (fmaxf(g.x, 0.0f) + fmaxf(g.y, 0.0f)) + (fmaxf(g.z, 0.0f) + fmaxf(g.w, 0.0f));
        } else {
            for (uint32_t i = i0; i < n; ++i) {
                float g = grid[i];
                const float t = tmp[i];
                if (g >= 0.0f && t >= 0.0f) g = fmaxf(g * decay, t);
                grid[i] = g;
                acc += fmaxf(g, 0.0f);
            }
        }
    }
    acc = n2m_wave_sum(acc);
    if ((threadIdx.x & 63u) == 0u) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0u) {
        partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1u;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    // fixed-order sum of the per-workgroup sums in double: the mean does not depend on which workgroup came last
    double s = 0.0;
    for (uint32_t j = threadIdx.x; j < gridDim.x; j += 256u) s += (double)__hip_atomic_load(partials + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __shared__ double dred[256];
    dred[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t w = 128u; w > 0u; w >>= 1) {
        if (threadIdx.x < w) dred[threadIdx.x] += dred[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0u) {
        const float mean = (float)(dred[0] / (double)n);
        *mean_out = mean;
        *thresh_out = fminf(mean, density_thresh);
        *ticket = 0u;
    }
}

extern "C" int n2m_occupancy_points(const float* cells, const float* u, const int32_t* idx, float inner, float half_grid_size, float* xyz,
                                    uint32_t n_points, void* stream) {
    N2M_NOTNULL(cells); N2M_NOTNULL(u); N2M_NOTNULL(xyz);
    if (n_points == 0) return 0;
    occupancy_points_kernel<<<n2m_ceil_div(n_points, 256), 256, 0, (hipStream_t)stream>>>(cells, u, idx, inner, half_grid_size, xyz, n_points);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" uint32_t n2m_occupancy_update_partials(uint32_t n) { const uint32_t b = n2m_ceil_div(n, 1024u); return b < 256u ? b : 256u; }

extern "C" int n2m_occupancy_update(float* grid, const float* tmp, float decay, uint32_t n, float density_thresh, float* partials,
                                    uint32_t* ticket, float* mean, float* thresh, void* stream) {
    N2M_NOTNULL(grid); N2M_NOTNULL(tmp); N2M_NOTNULL(partials); N2M_NOTNULL(ticket); N2M_NOTNULL(mean); N2M_NOTNULL(thresh);
    N2M_REQUIRE(n > 0 && (((uintptr_t)grid | (uintptr_t)tmp) & 15u) == 0, N2M_EINVAL, "occupancy_update: empty grid or pointers not 16-byte aligned");
    occupancy_update_kernel<<<n2m_occupancy_update_partials(n), 256, 0, (hipStream_t)stream>>>(grid, tmp, decay, n, density_thresh, partials, ticket, mean, thresh);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_flatten_rays(const int32_t* rays, uint32_t N, uint32_t M, int32_t* res, void* stream) {
    N2M_NOTNULL(rays); N2M_NOTNULL(res);
    if (N == 0) return 0;
    flatten_rays_kernel<<<n2m_ceil_div((uint64_t)N * 64, 256), 256, 0, (hipStream_t)stream>>>(rays, N, M, res);
    N2M_CHECK_LAUNCH();
    return 0;
}

static bool serial_resolve() {   // A/B switch: N2M_MARCH_RESOLVE=serial keeps the scalar-loop resolution of the visited subsequence for every ray
    static const bool v = getenv("N2M_MARCH_RESOLVE") != nullptr && getenv("N2M_MARCH_RESOLVE")[0] == 's';
    return v;
}

// workgroups of the wave marcher: one ray per wave, or N2M_MARCH_GRID_CAP workgroups walking the batch (see the kernel)
static uint32_t march_grid(uint32_t N) {
    // measurement switch: capping the grid did not pay on MI355X (256 / 512 / 1024 / 2048 workgroups: 1.21 / 1.05 / 0.99 / 0.97 ms per
    // training step against 0.98 uncapped) -- the overlapped pass then ends up on the critical path
    static const uint32_t cap = getenv("N2M_MARCH_GRID_CAP") ? (uint32_t)atoi(getenv("N2M_MARCH_GRID_CAP")) : 0u;
    const uint32_t full = n2m_ceil_div(N, 4);
    return cap && cap < full ? cap : full;
}

static int march_rays_train_impl(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound,
                                 int contract, float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                 const float* nears, const float* fars, float* xyzs, float* dirs, float* ts,
                                 int32_t* rays, int32_t* counter, const float* noises, uint32_t max_points, void* stream) {
    N2M_NOTNULL(rays_o); N2M_NOTNULL(rays_d); N2M_NOTNULL(grid); N2M_NOTNULL(nears); N2M_NOTNULL(fars);
    N2M_NOTNULL(rays); N2M_NOTNULL(noises);
    N2M_REQUIRE(C >= 1 && H >= 1 && H <= 1024 && max_steps >= 1, N2M_EINVAL,
                "march_rays_train: need C>=1, 1<=H<=1024, max_steps>=1 (got C=%u H=%u max_steps=%u)", C, H, max_steps);
    N2M_REQUIRE((double)C * H * H * H < 16777216.0 * 8, N2M_EINVAL, "march_rays_train: C*H^3 too large (%u x %u^3)", C, H);
    const bool first_pass = (xyzs == nullptr);
    if (!first_pass) { N2M_NOTNULL(dirs); N2M_NOTNULL(ts); }
    else N2M_NOTNULL(counter);
    hipStream_t s = (hipStream_t)stream;
    if (first_pass) {
        if (N > 0) {
            N2M_PROF(N2M_K_MARCH_COUNT, s, 52.0 * N);
            if (serial_march())
                march_train_kernel<false><<<n2m_ceil_div(N, 64), 64, 0, s>>>(rays_o, rays_d, grid, bound, contract != 0, dt_gamma,
                                                                             max_steps, N, C, H, nears, fars, nullptr, nullptr,
                                                                             nullptr, rays, noises, max_points);
            else
                march_train_wave_kernel<false><<<march_grid(N), 256, 0, s>>>(rays_o, rays_d, grid, bound, contract != 0,
                                                                                  dt_gamma, max_steps, N, C, H, nears, fars, nullptr,
                                                                                  nullptr, nullptr, rays, noises, max_points, serial_resolve());
            N2M_CHECK_LAUNCH();
        }
        const int rc = run_exclusive_scan(RayOffsetsOp{rays, counter}, N, s);
        if (rc) { n2m_set_error("march_rays_train: offset scan failed (%d)", rc); return rc; }
        N2M_CHECK_LAUNCH();
    } else if (N > 0) {
        N2M_PROF(N2M_K_MARCH_WRITE, s, 44.0 * N);   // + 32 B per sample, added by the caller who knows M
        if (serial_march())
            march_train_kernel<true><<<n2m_ceil_div(N, 64), 64, 0, s>>>(rays_o, rays_d, grid, bound, contract != 0, dt_gamma,
                                                                        max_steps, N, C, H, nears, fars, xyzs, dirs, ts, rays,
                                                                        noises, max_points);
        else
            march_train_wave_kernel<true><<<march_grid(N), 256, 0, s>>>(rays_o, rays_d, grid, bound, contract != 0, dt_gamma,
                                                                             max_steps, N, C, H, nears, fars, xyzs, dirs, ts, rays,
                                                                             noises, max_points, serial_resolve());
        N2M_CHECK_LAUNCH();
    }
    return 0;
}

// n2m_march_rays_train (pass 1 + pass 2) with ONE march per ray (see march_train_record_kernel): rays[n] = (offset, count) with the
// offsets in ray order from 0, counter[0] = the sample count, and the samples of every ray that fits max_points rows
// (raymarching.cu:417) -- the same bits the two-pass call produces with a zeroed counter.  `workspace`:
// n2m_march_fused_workspace_bytes(N) bytes, contents irrelevant.
static inline uint32_t march_groups(uint32_t N) { return (N >> kMarchGroupLog2) + 1u; }
// the group totals lead the workspace, padded to 256 bytes: their clear is then ONE aligned fill (an unaligned hipMemsetAsync splits into three)
static inline uint64_t march_group_bytes(uint32_t N) { return ((uint64_t)march_groups(N) * 4u + 255u) & ~(uint64_t)255u; }
extern "C" uint64_t n2m_march_fused_workspace_bytes(uint32_t N) {
    return march_group_bytes(N) + (uint64_t)N * kChunkRecCap * sizeof(ChunkRec) + (uint64_t)N * 4u + 64u;
}

extern "C" int n2m_march_rays_train_fused(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, int contract,
                                          float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, const float* nears,
                                          const float* fars, float* xyzs, float* dirs, float* ts, int32_t* rays, int32_t* counter,
                                          const float* noises, uint32_t max_points, void* workspace, uint64_t workspace_bytes, void* stream) {
    N2M_NOTNULL(rays_o); N2M_NOTNULL(rays_d); N2M_NOTNULL(grid); N2M_NOTNULL(nears); N2M_NOTNULL(fars);
    N2M_NOTNULL(rays); N2M_NOTNULL(noises); N2M_NOTNULL(counter); N2M_NOTNULL(workspace);
    if (max_points > 0) { N2M_NOTNULL(xyzs); N2M_NOTNULL(dirs); N2M_NOTNULL(ts); }
    N2M_REQUIRE(C >= 1 && H >= 1 && H <= 1024 && max_steps >= 1, N2M_EINVAL,
                "march_rays_train_fused: need C>=1, 1<=H<=1024, max_steps>=1 (got C=%u H=%u max_steps=%u)", C, H, max_steps);
    N2M_REQUIRE((double)C * H * H * H < 16777216.0 * 8, N2M_EINVAL, "march_rays_train_fused: C*H^3 too large (%u x %u^3)", C, H);
    N2M_REQUIRE(workspace_bytes >= n2m_march_fused_workspace_bytes(N), N2M_EINVAL, "march_rays_train_fused: workspace of %llu bytes, need %llu",
                (unsigned long long)workspace_bytes, (unsigned long long)n2m_march_fused_workspace_bytes(N));
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) { N2M_HIP(hipMemsetAsync(counter, 0, sizeof(int32_t), s)); return 0; }
    uint32_t* group_total = (uint32_t*)workspace;                            // [N / 256 + 1], padded
    ChunkRec* recs = (ChunkRec*)((char*)workspace + march_group_bytes(N));  // [N][kChunkRecCap]
    uint32_t* n_recs = (uint32_t*)(recs + (size_t)N * kChunkRecCap);        // [N]
    const uint32_t blocks = n2m_ceil_div(N, 4);
    N2M_HIP(hipMemsetAsync(group_total, 0, march_group_bytes(N), s));
    {
        N2M_PROF_K(N2M_K_MARCH_COUNT, s, 52.0 * N);
        N2M_LAUNCH(march_train_record_kernel, blocks, 256, 0, s, rays_o, rays_d, grid, bound, contract != 0, dt_gamma, max_steps, N, C, H, nears, fars, rays,
                                                         noises, serial_resolve(), recs, n_recs, group_total);
        N2M_CHECK_LAUNCH();
    }
    {
        N2M_PROF_K(N2M_K_MARCH_WRITE, s, 44.0 * N);   // + 32 B per sample, added by the caller who knows M
        N2M_LAUNCH(march_train_replay_kernel, blocks, 256, 0, s, rays_o, rays_d, grid, bound, contract != 0, dt_gamma, max_steps, N, C, H, nears, fars, xyzs,
                                                         dirs, ts, rays, counter, noises, max_points, serial_resolve(), recs, n_recs, group_total);
        N2M_CHECK_LAUNCH();
    }
    return 0;
}

// Diagnostics: number of rays (count passes only) whose prefix-maximum resolution failed its exactness check and that were re-marched
// with the serial resolution since the last call; resets the counter.  Synchronises the device.
extern "C" int n2m_march_fallback_count(uint32_t* out) {
    N2M_NOTNULL(out);
    unsigned int v = 0, zero = 0;
    N2M_HIP(hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_march_fallbacks), sizeof(v)));
    N2M_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_march_fallbacks), &zero, sizeof(zero)));
    *out = v;
    return 0;
}

extern "C" int n2m_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound,
                                    int contract, float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                    const float* nears, const float* fars, float* xyzs, float* dirs, float* ts,
                                    int32_t* rays, int32_t* counter, const float* noises, void* stream) {
    return march_rays_train_impl(rays_o, rays_d, grid, bound, contract, dt_gamma, max_steps, N, C, H, nears, fars, xyzs, dirs, ts, rays,
                                 counter, noises, 0xFFFFFFFFu, stream);
}

// Pass 2 into sample buffers of max_points rows: a ray whose range does not fit is not written (the reference's guard, :417).
// Lets a caller issue the write pass BEFORE it has read the sample count back (buffers sized from the previous batch).
extern "C" int n2m_march_rays_train_write(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound,
                                          int contract, float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                          const float* nears, const float* fars, float* xyzs, float* dirs, float* ts,
                                          const int32_t* rays, const float* noises, uint32_t max_points, void* stream) {
    N2M_NOTNULL(xyzs);
    return march_rays_train_impl(rays_o, rays_d, grid, bound, contract, dt_gamma, max_steps, N, C, H, nears, fars, xyzs, dirs, ts,
                                 const_cast<int32_t*>(rays), nullptr, noises, max_points, stream);
}

extern "C" int n2m_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* ts,
                                                const int32_t* rays, uint32_t M, uint32_t N, float T_thresh,
                                                int alpha_mode, float* weights, float* weights_sum, float* depth,
                                                float* image, void* stream) {
    N2M_NOTNULL(rays); N2M_NOTNULL(weights_sum); N2M_NOTNULL(depth); N2M_NOTNULL(image);
    if (M > 0) { N2M_NOTNULL(sigmas); N2M_NOTNULL(rgbs); N2M_NOTNULL(ts); N2M_NOTNULL(weights); }
    if (N == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    N2M_PROF(N2M_K_COMPOSITE_FWD, s, 28.0 * M + 28.0 * N);
    composite_train_fwd_kernel<<<n2m_ceil_div(N, 4), 256, 0, s>>>(sigmas, rgbs, ts, rays, M, N, T_thresh, alpha_mode != 0,
                                                                   weights, weights_sum, depth, image);
    N2M_CHECK_LAUNCH();
    return 0;
}

// Training fast path: compositing, loss head and both backward passes of n2m_composite_rays_train_forward/backward +
// n2m_photo_loss_forward/backward in one launch (density mode; no grad_weights / grad_depth: the plain rgb + mask loss).
// Optional outputs of the NEXT n2m_composite_loss_train* calls of this thread (sticky until cleared with NULLs): per ray the number of
// samples up to and including the one the early stop fell on (every later sample of the ray receives exactly zero gradients: raymarching.cu:
// 553,640 `break`), and their sums per workgroup of 16 rays.  n2m_sample_order_live_first turns them into the order the table backward wants.
static thread_local int32_t* g_live_out = nullptr;
static thread_local uint32_t* g_block_live_out = nullptr;
extern "C" int n2m_composite_live_counts(int32_t* live /*[N]*/, uint32_t* block_live /*[ceil(N / 16)]*/) {
    N2M_REQUIRE((live == nullptr) == (block_live == nullptr), N2M_ENULL, "n2m_composite_live_counts: both outputs or neither");
    g_live_out = live;
    g_block_live_out = block_live;
    return 0;
}

// perm[0 .. M) <- the samples of a marched batch with every ray's live prefix first (ray order, sample order inside a ray), then every
// ray's dead tail (same order): position of sample off_r + k = live_before(r) + k for k < live_r, else M_live + (off_r - live_before(r)) +
// (k - live_r).  One launch: a workgroup owns 16 rays (one workgroup of the compositing kernel) and derives its two offsets from the
// per-workgroup sums itself (N / 16 values, L2-resident).  Rays cut off by M (off + cnt > M) count as dead over [off, M).  If the ranges
// do not tile [0, M) -- never the case for n2m_march_rays_train's (offset, count) pairs -- the order degrades to the identity.
__global__ void __launch_bounds__(256) sample_order_kernel(const int32_t* __restrict__ rays, const int32_t* __restrict__ live,
                                                           const uint32_t* __restrict__ block_live, uint32_t N, uint32_t M,
                                                           uint32_t* __restrict__ perm) {
    __shared__ uint32_t red[2][4];
    __shared__ uint32_t ray_live_off[16], ray_off[16], ray_cnt[16], ray_live[16];
    const uint32_t nblk = (N + 15u) / 16u, b = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    uint32_t before = 0u, total = 0u;
    for (uint32_t i = tid; i < nblk; i += 256u) {
        const uint32_t v = block_live[i];
        total += v;
        if (i < b) before += v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { before += __shfl_xor(before, o, 64); total += __shfl_xor(total, o, 64); }
    if (lane == 0u) { red[0][wid] = before; red[1][wid] = total; }
    __syncthreads();
    before = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    total = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    if (tid < 16u) {
        const uint32_t n = b * 16u + tid;
        uint32_t off = 0u, c = 0u, lv = 0u;
        if (n < N) {
            off = (uint32_t)rays[2 * n];
            const uint32_t cnt = (uint32_t)rays[2 * n + 1];
            c = off < M ? min(cnt, M - off) : 0u;
            lv = min((uint32_t)live[n], c);
        }
        // exclusive prefix of the 16 rays' live counts (lanes 0..15 of wave 0)
        uint32_t incl = lv;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) { const uint32_t t = __shfl_up(incl, d, 64); if ((int)tid >= d) incl += t; }
        ray_live_off[tid] = before + incl - lv;
        ray_off[tid] = off; ray_cnt[tid] = c; ray_live[tid] = lv;
    }
    __syncthreads();
    for (uint32_t r = wid; r < 16u; r += 4u) {
        const uint32_t off = ray_off[r], c = ray_cnt[r], lv = ray_live[r], lo = ray_live_off[r];
        const uint32_t dead0 = total + (off - lo);
        for (uint32_t k = lane; k < c; k += 64u) {
            const uint32_t pos = k < lv ? lo + k : dead0 + (k - lv);
            if (pos < M) perm[pos] = off + k;
        }
    }
}

extern "C" int n2m_sample_order_live_first(const int32_t* rays, const int32_t* live, const uint32_t* block_live, uint32_t N, uint32_t M,
                                           uint32_t* perm, void* stream) {
    if (N == 0 || M == 0) return 0;
    N2M_NOTNULL(rays); N2M_NOTNULL(live); N2M_NOTNULL(block_live); N2M_NOTNULL(perm);
    hipLaunchKernelGGL(sample_order_kernel, dim3(n2m_ceil_div(N, 16)), dim3(256), 0, (hipStream_t)stream, rays, live, block_live, N, M, perm);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_composite_loss_train_ex(const float* sigmas, const float* rgbs, const float* ts, const int32_t* rays, uint32_t M, uint32_t N,
                                        float T_thresh, const float* gt_rgba, const float* bg, float bg_scalar, float lambda_rgb,
                                        float lambda_mask, const float* grad_loss, float* weights_sum, float* image, float* grad_sigmas,
                                        float* grad_rgbs, float* partial, uint32_t* ticket, float* loss, float* loss_sum, float lambda_entropy,
                                            int alpha_mode, void* stream) {
    N2M_NOTNULL(rays); N2M_NOTNULL(gt_rgba); N2M_NOTNULL(grad_loss); N2M_NOTNULL(partial);
    N2M_REQUIRE(ticket == nullptr || loss != nullptr, N2M_ENULL, "composite_loss_train: a ticket needs the loss output");
    if (M > 0) { N2M_NOTNULL(sigmas); N2M_NOTNULL(rgbs); N2M_NOTNULL(ts); N2M_NOTNULL(grad_sigmas); N2M_NOTNULL(grad_rgbs); }
    if (N == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    N2M_PROF_K(N2M_K_COMPOSITE_FWD, s, 28.0 * M + 28.0 * N + 44.0 * M + 48.0 * N);     // forward + backward of SURVEY 8d, one launch
    if (alpha_mode) {
        N2M_REQUIRE(lambda_entropy <= 0.0f, N2M_EUNSUPPORTED, "composite_loss_train: alpha mode with the entropy term is not built");
        N2M_LAUNCH((composite_loss_train_kernel<false, true>), n2m_ceil_div(N, 16), 1024, 0, s, sigmas, rgbs, ts, rays, M, N, T_thresh, gt_rgba, bg, bg_scalar,
                                                                                     lambda_rgb, lambda_mask, grad_loss, weights_sum, image, grad_sigmas,
                                                                                     grad_rgbs, partial, ticket, loss, loss_sum, 0.0f, g_live_out, g_block_live_out);
    } else if (lambda_entropy > 0.0f)
        N2M_LAUNCH((composite_loss_train_kernel<true>), n2m_ceil_div(N, 16), 1024, 0, s, sigmas, rgbs, ts, rays, M, N, T_thresh, gt_rgba, bg, bg_scalar,
                                                                              lambda_rgb, lambda_mask, grad_loss, weights_sum, image, grad_sigmas,
                                                                              grad_rgbs, partial, ticket, loss, loss_sum, lambda_entropy, g_live_out, g_block_live_out);
    else
        N2M_LAUNCH((composite_loss_train_kernel<false>), n2m_ceil_div(N, 16), 1024, 0, s, sigmas, rgbs, ts, rays, M, N, T_thresh, gt_rgba, bg, bg_scalar,
                                                                               lambda_rgb, lambda_mask, grad_loss, weights_sum, image, grad_sigmas,
                                                                               grad_rgbs, partial, ticket, loss, loss_sum, 0.0f, g_live_out, g_block_live_out);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_composite_loss_train_ent(const float* sigmas, const float* rgbs, const float* ts, const int32_t* rays, uint32_t M, uint32_t N,
                                            float T_thresh, const float* gt_rgba, const float* bg, float bg_scalar, float lambda_rgb,
                                            float lambda_mask, const float* grad_loss, float* weights_sum, float* image, float* grad_sigmas,
                                            float* grad_rgbs, float* partial, uint32_t* ticket, float* loss, float* loss_sum, float lambda_entropy,
                                            void* stream) {
    return n2m_composite_loss_train_ex(sigmas, rgbs, ts, rays, M, N, T_thresh, gt_rgba, bg, bg_scalar, lambda_rgb, lambda_mask, grad_loss, weights_sum,
                                       image, grad_sigmas, grad_rgbs, partial, ticket, loss, loss_sum, lambda_entropy, 0, stream);
}

extern "C" int n2m_composite_loss_train(const float* sigmas, const float* rgbs, const float* ts, const int32_t* rays, uint32_t M, uint32_t N,
                                        float T_thresh, const float* gt_rgba, const float* bg, float bg_scalar, float lambda_rgb,
                                        float lambda_mask, const float* grad_loss, float* weights_sum, float* image, float* grad_sigmas,
                                        float* grad_rgbs, float* partial, uint32_t* ticket, float* loss, float* loss_sum, void* stream) {
    return n2m_composite_loss_train_ent(sigmas, rgbs, ts, rays, M, N, T_thresh, gt_rgba, bg, bg_scalar, lambda_rgb, lambda_mask, grad_loss,
                                        weights_sum, image, grad_sigmas, grad_rgbs, partial, ticket, loss, loss_sum, 0.0f, stream);
}

extern "C" int n2m_composite_rays_train_backward(const float* grad_weights, const float* grad_weights_sum,
                                                 const float* grad_depth, const float* grad_image, const float* sigmas,
                                                 const float* rgbs, const float* ts, const int32_t* rays,
                                                 const float* weights_sum, const float* depth, const float* image,
                                                 uint32_t M, uint32_t N, float T_thresh, int alpha_mode,
                                                 float* grad_sigmas, float* grad_rgbs, void* stream) {
    N2M_NOTNULL(rays); N2M_NOTNULL(grad_weights_sum); N2M_NOTNULL(grad_depth); N2M_NOTNULL(grad_image);
    N2M_NOTNULL(weights_sum); N2M_NOTNULL(depth); N2M_NOTNULL(image);
    if (M == 0 || N == 0) return 0;
    N2M_NOTNULL(grad_weights); N2M_NOTNULL(sigmas); N2M_NOTNULL(rgbs); N2M_NOTNULL(ts);
    N2M_NOTNULL(grad_sigmas); N2M_NOTNULL(grad_rgbs);
    hipStream_t s = (hipStream_t)stream;
    N2M_PROF(N2M_K_COMPOSITE_BWD, s, 44.0 * M + 48.0 * N);
    composite_train_bwd_kernel<<<n2m_ceil_div(N, 4), 256, 0, s>>>(grad_weights, grad_weights_sum, grad_depth, grad_image, sigmas,
                                                                   rgbs, ts, rays, weights_sum, depth, image, M, N, T_thresh,
                                                                   alpha_mode != 0, grad_sigmas, grad_rgbs);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                              const float* rays_o, const float* rays_d, float bound, int contract, float dt_gamma,
                              uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid, const float* nears,
                              const float* fars, float* xyzs, float* dirs, float* ts, const float* noises, void* stream) {
    (void)nears;
    if (n_alive == 0 || n_step == 0) return 0;
    N2M_NOTNULL(rays_alive); N2M_NOTNULL(rays_t); N2M_NOTNULL(rays_o); N2M_NOTNULL(rays_d); N2M_NOTNULL(grid);
    N2M_NOTNULL(fars); N2M_NOTNULL(xyzs); N2M_NOTNULL(dirs); N2M_NOTNULL(ts); N2M_NOTNULL(noises);
    N2M_REQUIRE(C >= 1 && H >= 1 && H <= 1024 && max_steps >= 1, N2M_EINVAL, "march_rays: bad C/H/max_steps");
    if (n_alive <= infer_wave_threshold() && n_step <= 64u)
        march_infer_wave_kernel<<<n2m_ceil_div(n_alive, 4), 256, 0, (hipStream_t)stream>>>(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound,
                                                                                          contract != 0, dt_gamma, max_steps, C, H, grid, fars, xyzs,
                                                                                          dirs, ts, noises);
    else
        march_infer_kernel<<<n2m_ceil_div(n_alive, 64), 64, 0, (hipStream_t)stream>>>(n_alive, n_step, rays_alive, rays_t, rays_o,
                                                                                     rays_d, bound, contract != 0, dt_gamma,
                                                                                     max_steps, C, H, grid, fars, xyzs, dirs, ts,
                                                                                     noises);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int alpha_mode, int32_t* rays_alive,
                                  float* rays_t, const float* sigmas, const float* rgbs, const float* ts,
                                  float* weights_sum, float* depth, float* image, void* stream) {
    if (n_alive == 0) return 0;
    N2M_NOTNULL(rays_alive); N2M_NOTNULL(rays_t); N2M_NOTNULL(weights_sum); N2M_NOTNULL(depth); N2M_NOTNULL(image);
    if (n_step > 0) { N2M_NOTNULL(sigmas); N2M_NOTNULL(rgbs); N2M_NOTNULL(ts); }
    composite_infer_kernel<<<n2m_ceil_div(n_alive, 128), 128, 0, (hipStream_t)stream>>>(n_alive, n_step, T_thresh,
                                                                                       alpha_mode != 0, rays_alive, rays_t,
                                                                                       sigmas, rgbs, ts, weights_sum, depth,
                                                                                       image);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_compact_alive(const int32_t* rays_alive, uint32_t n_alive, int32_t* out, int32_t* n_out_dev,
                                 void* stream) {
    N2M_NOTNULL(n_out_dev);
    if (n_alive > 0) { N2M_NOTNULL(rays_alive); N2M_NOTNULL(out); }
    const int rc = run_exclusive_scan(CompactOp{rays_alive, out, n_out_dev}, n_alive, (hipStream_t)stream);
    if (rc) { n2m_set_error("compact_alive: scan failed (%d)", rc); return rc; }
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_select_positive(const float* values, uint32_t n, uint32_t stride, int64_t* out_idx, int32_t* n_out_dev, void* stream) {
    N2M_NOTNULL(n_out_dev);
    N2M_REQUIRE(stride >= 1, N2M_EINVAL, "select_positive: stride must be >= 1 float");
    if (n > 0) { N2M_NOTNULL(values); N2M_NOTNULL(out_idx); }
    const int rc = run_exclusive_scan(SelectPositiveOp{values, stride, out_idx, n_out_dev}, n, (hipStream_t)stream);
    if (rc) { n2m_set_error("select_positive: scan failed (%d)", rc); return rc; }
    N2M_CHECK_LAUNCH();
    return 0;
}

// ---- the device-count forms (include/n2m_hip.h): `state` = {n_alive, step} in device memory, n_alive_ub = the host's upper bound of n_alive
extern "C" int n2m_march_rays_dev(const int32_t* state, uint32_t n_alive_ub, uint32_t N, const int32_t* rays_alive, const float* rays_t,
                                  const float* rays_o, const float* rays_d, float bound, int contract, float dt_gamma, uint32_t max_steps,
                                  uint32_t C, uint32_t H, const uint8_t* grid, const float* fars, float* xyzs, float* dirs, float* ts,
                                  const float* noises, void* stream) {
    if (n_alive_ub == 0) return 0;
    N2M_NOTNULL(state); N2M_NOTNULL(rays_alive); N2M_NOTNULL(rays_t); N2M_NOTNULL(rays_o); N2M_NOTNULL(rays_d); N2M_NOTNULL(grid);
    N2M_NOTNULL(fars); N2M_NOTNULL(xyzs); N2M_NOTNULL(dirs); N2M_NOTNULL(ts);
    N2M_REQUIRE(C >= 1 && H >= 1 && H <= 1024 && max_steps >= 1 && N >= n_alive_ub, N2M_EINVAL, "march_rays_dev: bad C/H/max_steps/N");
    // (the device-side n_step is at most 8: the clear of unused slots by lanes 0 .. n_step - 1 always fits a wave)
    if (n_alive_ub <= infer_wave_threshold())
        march_infer_wave_dev_kernel<<<n2m_ceil_div(n_alive_ub, 4), 256, 0, (hipStream_t)stream>>>(state, N, rays_alive, rays_t, rays_o, rays_d, bound,
                                                                                                 contract != 0, dt_gamma, max_steps, C, H, grid, fars,
                                                                                                 xyzs, dirs, ts, noises);
    else
        march_infer_dev_kernel<<<n2m_ceil_div(n_alive_ub, 64), 64, 0, (hipStream_t)stream>>>(state, N, rays_alive, rays_t, rays_o, rays_d, bound,
                                                                                            contract != 0, dt_gamma, max_steps, C, H, grid, fars,
                                                                                            xyzs, dirs, ts, noises);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_composite_rays_dev(const int32_t* state, uint32_t n_alive_ub, uint32_t N, uint32_t max_steps, float T_thresh, int alpha_mode,
                                      int32_t* rays_alive, float* rays_t, const float* sigmas, const float* rgbs, const float* ts,
                                      float* weights_sum, float* depth, float* image, void* stream) {
    if (n_alive_ub == 0) return 0;
    N2M_NOTNULL(state); N2M_NOTNULL(rays_alive); N2M_NOTNULL(rays_t); N2M_NOTNULL(weights_sum); N2M_NOTNULL(depth); N2M_NOTNULL(image);
    N2M_NOTNULL(sigmas); N2M_NOTNULL(rgbs); N2M_NOTNULL(ts);
    composite_infer_dev_kernel<<<n2m_ceil_div(n_alive_ub, 128), 128, 0, (hipStream_t)stream>>>(state, N, max_steps, T_thresh, alpha_mode != 0,
                                                                                              rays_alive, rays_t, sigmas, rgbs, ts, weights_sum,
                                                                                              depth, image);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_compact_alive_dev(const int32_t* rays_alive, const int32_t* state, uint32_t n_alive_ub, uint32_t N, uint32_t max_steps,
                                     int32_t* out, int32_t* state_out, void* stream) {
    N2M_NOTNULL(state); N2M_NOTNULL(state_out); N2M_NOTNULL(rays_alive); N2M_NOTNULL(out);
    N2M_REQUIRE(state != state_out && rays_alive != out, N2M_EINVAL, "compact_alive_dev: state / ray lists must ping-pong (in != out)");
    // n_alive_ub == 0 still runs the one-block scan: its finish() writes the next state
    const int rc = run_exclusive_scan(CompactDevOp{rays_alive, out, state, state_out, N, max_steps}, n_alive_ub, (hipStream_t)stream);
    if (rc) { n2m_set_error("compact_alive_dev: scan failed (%d)", rc); return rc; }
    N2M_CHECK_LAUNCH();
    return 0;
}
