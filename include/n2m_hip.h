/*
 * n2m_hip.h -- C ABI of libn2m_hip.so: the MI355X (gfx950) implementation of nerf2mesh's hot path.
 *
 * This is the drop-in boundary.  Every entry point replaces one function of the
 * reference's `_backend` extension modules; the reference interface it stands in for is cited next to it
 * (paths relative to the reference checkout).  The host-side Python above this ABI
 * (nerf2mesh_amd/backends/_raymarching_mob.py, _gridencoder.py, _shencoder.py, and the operator modules
 * nerf2mesh_amd/{raymarching,gridencoder,shencoder}.py) binds these symbols with ctypes.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer (HBM) unless the name ends in _host.
 *  - all outputs are pre-allocated by the caller and written in place; the library never allocates or frees
 *    caller-visible memory and never synchronises the stream (exception: n2m_prof_* readers, documented there).
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Work is enqueued, not awaited.
 *  - return value: 0 on success, otherwise a negative N2M_E* code for a rejected argument or the positive
 *    hipError_t of a failed launch.  n2m_last_error() returns a static, thread-local message for the last
 *    non-zero return on this thread.
 *  - `dtype` selects the embedding/feature element type of the grid encoder: N2M_F32 or N2M_F16 (IEEE half).
 *    Ray marching, compositing and SH always compute in fp32 (the reference wrappers cast to fp32 with
 *    custom_fwd(cast_inputs=torch.float32): raymarching/raymarching.py:21,54,130,186,250,313,364).
 */
#ifndef N2M_HIP_H
#define N2M_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define N2M_ABI_VERSION 1

enum { N2M_F32 = 0, N2M_F16 = 1 };

enum {
    N2M_OK = 0,
    N2M_EINVAL = -1,      /* bad scalar argument (unsupported D / C / degree ...) */
    N2M_ENULL = -2,       /* required pointer is NULL */
    N2M_EUNSUPPORTED = -3 /* valid in the reference, not implemented here (message says what) */
};

int n2m_abi_version(void);
const char* n2m_last_error(void);

/* ------------------------------------------------------------------------------------------------------
 * raymarching   (reference: raymarching/src/raymarching.h:7-19, raymarching/src/bindings.cpp:5-20)
 * ---------------------------------------------------------------------------------------------------- */

/* raymarching.h:7  near_far_from_aabb   kernel raymarching.cu:91-145
 * rays_o, rays_d [N,3]; aabb [6] = (xmin,ymin,zmin,xmax,ymax,zmax); nears, fars [N].
 * Miss => nears[n] = fars[n] = FLT_MAX.  near is clamped from below to min_near. */
int n2m_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N,
                           float min_near, float* nears, float* fars, void* stream);

/* raymarching.h:8  sph_from_ray   kernel raymarching.cu:162-198.  coords [N,2] in [-1,1]. */
int n2m_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords,
                     void* stream);

/* raymarching.h:9-10  morton3D / morton3D_invert   kernels raymarching.cu:214-254.
 * coords [N,3] int32 (10 bits per axis are used), indices [N] int32. */
int n2m_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, void* stream);
int n2m_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords, void* stream);

/* raymarching.h:11  packbits   kernel raymarching.cu:267-289.
 * grid [N*8] fp32, bitfield [N] u8: bit i of byte n  <=>  grid[8n+i] > density_thresh  (strict, LSB first). */
int n2m_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield, void* stream);
/* New: packbits with the threshold read from device memory, so that the occupancy refresh (threshold = min(mean density,
 * density_thresh), nerf/renderer.py:1142-1145) needs no host read-back of the mean. */
int n2m_packbits_dev(const float* grid, uint32_t N, const float* density_thresh, uint8_t* bitfield, void* stream);

/* Occupancy refresh (nerf/renderer.py:1074-1149), the elementwise work around the density query.
 * n2m_occupancy_points: xyz[j] = cells[c] * inner + (u[c] * 2 - 1) * half_grid_size per component, c = idx ? idx[j] : j, for n_points points --
 * the reference's expression (:1096-1100: cells [H^3, 3] = 2 c / (H - 1) - 1 in Morton order, inner = bound - half_grid_size, u = torch.rand_like
 * draws for ALL cells) with its rounding points.  idx lists the cells whose grid value is >= 0: the others are never updated (:1131-1134), so
 * their density need not be queried (outdoor recipe, 5 cascades: 37 % of the cells are valid).
 * n2m_occupancy_update: grid = max(grid * decay, tmp) where both are >= 0 (:1133-1134), mean of max(grid, 0) (:1136) and the threshold
 * min(mean, density_thresh) (:1140) left on the device for n2m_packbits_dev -- no host read-back (the reference reads the mean with .item()).
 * partials: n2m_occupancy_update_partials(n) floats of scratch; ticket: one zero-initialised u32 the kernel resets; n floats, 16-byte aligned. */
int n2m_occupancy_points(const float* cells, const float* u, const int32_t* idx, float inner, float half_grid_size, float* xyz, uint32_t n_points,
                         void* stream);
uint32_t n2m_occupancy_update_partials(uint32_t n);
int n2m_occupancy_update(float* grid, const float* tmp, float decay, uint32_t n, float density_thresh, float* partials, uint32_t* ticket,
                         float* mean, float* thresh, void* stream);

/* raymarching.h:12  flatten_rays   kernel raymarching.cu:303-319.
 * rays [N,2] = (offset,count); res [M] int32: res[offset .. offset+count) = n.  Other entries untouched. */
int n2m_flatten_rays(const int32_t* rays, uint32_t N, uint32_t M, int32_t* res, void* stream);

/* raymarching.h:14  march_rays_train   kernel raymarching.cu:337-475, host :477-489.
 * Two-pass protocol of raymarching/raymarching.py:229-241:
 *   pass 1: xyzs == dirs == ts == NULL.  Counts the kept samples of every ray and writes
 *           rays[n] = (offset, count), counter[0] += sum(count).
 *           Offsets are the EXCLUSIVE PREFIX SUM of the counts IN RAY ORDER starting at the value counter[0]
 *           held on entry (the reference hands out offsets with atomicAdd in thread-arrival order,
 *           raymarching.cu:471: any consistent packing is valid there; ray order is the deterministic one and
 *           is what a serial execution of the reference produces).
 *   pass 2: xyzs [M,3], dirs [M,3], ts [M,2] given.  Re-marches with the same arithmetic and writes the
 *           samples of ray n at rays[n].offset.  counter is not touched.
 * grid = density bitfield [C*H^3/8] u8, Morton order inside each cascade.  noises [N] in [0,1). */
int n2m_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound,
                         int contract, float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                         const float* nears, const float* fars, float* xyzs, float* dirs, float* ts,
                         int32_t* rays, int32_t* counter, const float* noises, void* stream);
/* Pass 2 alone, into sample buffers of max_points rows: a ray whose (offset, count) range does not fit is not written
 * (the reference's `if (point_index + num_steps > M) return`, raymarching.cu:417).  With it the write pass can be queued
 * before the host has read the sample count back: buffers sized from the previous batch, exact re-run on overflow. */
int n2m_march_rays_train_write(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, int contract,
                               float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                               const float* nears, const float* fars, float* xyzs, float* dirs, float* ts,
                               const int32_t* rays, const float* noises, uint32_t max_points, void* stream);

/* The protocol of raymarching.py:229-241 with ONE march per ray: the count pass also records, per ray, the chunks of 64 candidates
 * that kept samples (kept-lane mask + T of the first lane, 16 B each); a replay kernel derives every ray's offset -- the ray-order
 * prefix sum of the counts, from 0 -- from per-group totals and recomputes only the kept candidates into the sample rows (no second
 * walk of the occupancy grid, no scan kernel).  Outputs: rays [N,2] = (offset, count), counter[0] = sample count, xyzs/dirs/ts rows
 * of every ray whose range fits max_points rows (raymarching.cu:417): bit-identical to n2m_march_rays_train (zeroed counter) +
 * n2m_march_rays_train_write.  workspace: n2m_march_fused_workspace_bytes(N) bytes of device memory, contents irrelevant. */
uint64_t n2m_march_fused_workspace_bytes(uint32_t N);
int n2m_march_rays_train_fused(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, int contract,
                               float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                               const float* nears, const float* fars, float* xyzs, float* dirs, float* ts,
                               int32_t* rays, int32_t* counter, const float* noises, uint32_t max_points,
                               void* workspace, uint64_t workspace_bytes, void* stream);

/* Settings of the binned table backward for the CALLING THREAD (all n2m_grid_encode_backward_binned* / n2m_grad_total_variation_binned calls
 * this thread makes afterwards; thread-local, default (1, 1.0) -- state them before each call and no other engine in the process can interfere).  tv_stride: floats between consecutive rows of the TV table (1 = an fp32 [rows,1] tensor as in gridencoder.h:15; 2 = the
 * density column of the packed {fp32, half2} table n2m_grid_encode_forward_packed reads -- a caller that shards the optimizer over ranks
 * keeps only that copy complete).  overflow_div: a gradient row whose magnitude exceeds max_of_type / overflow_div raises found_inf;
 * a caller that sums the tables over W ranks passes W, so that an overflow only the cross-rank sum would produce is caught before it. */
int n2m_grid_backward_config(int tv_stride, float overflow_div);

/* event: a hipEvent_t (or NULL).  The calling thread's NEXT n2m_grid_encode_backward_binned_pair* call records it on its stream between its
 * fill and its accumulate kernels (one-shot; behind the call when the path taken has no such point).  What a step executor hands to a
 * second stream as its go-ahead: work that needs neither this step's gradients nor its updated parameters (the next batch's ray generation
 * and march, nerf/provider.py:302-330 + raymarching.py:195-262) starts beside the accumulate. */
int n2m_grid_backward_mid_event(void* event);

/* Measurement aid for the shared fill of n2m_grid_encode_backward_binned_pair: on != 0 arms shader-clock stamps in one of its workgroups
 * (first 8 tile iterations x 6 phase boundaries) and in two work items of each accumulate kernel (5 boundaries); out (may be NULL,
 * else 116 words) receives the stamps of the last armed launch.  Synchronises. */
int n2m_debug_fill_times(int on, unsigned long long* out);

/* Diagnostics.  march_rays_train resolves which candidates a ray visits with a wave-wide prefix maximum whose result is provably the
 * serial chain's whenever its check passes; a ray that fails the check is re-marched with the serial resolution.  Returns in *out
 * the number of such rays (count passes) since the previous call and resets it.  Synchronises the device. */
int n2m_march_fallback_count(uint32_t* out);

/* raymarching.h:15  composite_rays_train_forward   kernel raymarching.cu:500-578.
 * sigmas [M], rgbs [M,3], ts [M,2], rays [N,2] -> weights [M], weights_sum [N], depth [N], image [N,3].
 * weights: every sample of a ray's (offset, count) range inside [0,M) is written -- zero after the early stop, and a
 * ray cut off by M (skipped like :521) zeroes its part.  The reference zero-fills instead (raymarching.py:262); a caller
 * whose ranges tile [0,M), as march_rays_train's do, can skip that, any other caller zero-fills as before.  The same
 * holds for grad_sigmas / grad_rgbs of the backward. */
int n2m_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* ts,
                                     const int32_t* rays, uint32_t M, uint32_t N, float T_thresh,
                                     int alpha_mode, float* weights, float* weights_sum, float* depth,
                                     float* image, void* stream);

/* raymarching.h:16  composite_rays_train_backward   kernel raymarching.cu:604-694.
 * grad_sigmas [M], grad_rgbs [M,3] are caller-zero-filled; samples after the early stop stay zero. */
int n2m_composite_rays_train_backward(const float* grad_weights, const float* grad_weights_sum,
                                      const float* grad_depth, const float* grad_image, const float* sigmas,
                                      const float* rgbs, const float* ts, const int32_t* rays,
                                      const float* weights_sum, const float* depth, const float* image,
                                      uint32_t M, uint32_t N, float T_thresh, int alpha_mode,
                                      float* grad_sigmas, float* grad_rgbs, void* stream);

/* raymarching.h:18  march_rays (inference)   kernel raymarching.cu:712-828.
 * Fixed n_step slots per alive ray; xyzs/dirs/ts [n_alive*n_step, .] are caller-zero-filled, a slot with
 * ts[.,0] == 0 is an empty slot.  noises [n_alive]. */
int n2m_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                   const float* rays_o, const float* rays_d, float bound, int contract, float dt_gamma,
                   uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid, const float* nears,
                   const float* fars, float* xyzs, float* dirs, float* ts, const float* noises, void* stream);

/* raymarching.h:19  composite_rays (inference)   kernel raymarching.cu:841-924.
 * Accumulates into weights_sum/depth/image [N] in place; rays_alive[n] = -1 when the ray ended, otherwise
 * rays_t[ray] = last t. */
int n2m_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int alpha_mode, int32_t* rays_alive,
                       float* rays_t, const float* sigmas, const float* rgbs, const float* ts,
                       float* weights_sum, float* depth, float* image, void* stream);

/* New (no reference counterpart): stream compaction of the alive list,
 * out[0..n_out) = the entries of rays_alive[0..n_alive) that are >= 0, order preserved; *n_out_dev = count.
 * Replaces `rays_alive = rays_alive[rays_alive >= 0]` (nerf/renderer.py:798). */
int n2m_compact_alive(const int32_t* rays_alive, uint32_t n_alive, int32_t* out, int32_t* n_out_dev,
                      void* stream);

/* New (no reference counterpart): out_idx[0..n_out) = the indices i in [0, n) with values[i * stride] > 0, ascending (int64, what torch indexing
 * takes); *n_out_dev = their number.  Replaces `torch.nonzero(mask > 0)` in front of the stage-1 shading (the covered pixels of
 * nerf/renderer.py:873-875, `mask` = the interpolated coverage) by the library's own scan: same indices, three launches, no host read inside. */
int n2m_select_positive(const float* values, uint32_t n, uint32_t stride, int64_t* out_idx, int32_t* n_out_dev, void* stream);

/* The inference loop of nerf/renderer.py:764-802 with the ray count kept ON THE DEVICE.  The reference reads n_alive back every round
 * (`rays_alive = rays_alive[rays_alive >= 0]`, :799) to size the next round's launches and to derive n_step = max(min(N // n_alive, 8), 1)
 * (:775).  Here `state` = {n_alive, step} (int32 x 2, device) carries both; the three kernels of a round read it, derive n_step the same way
 * and skip the round when n_alive == 0 or step >= max_steps; n2m_compact_alive_dev writes the next round's state {survivors, step + n_step}
 * into state_out (state and the ray lists ping-pong between two buffers).  The host passes n_alive_ub >= n_alive (the count never grows, so
 * any earlier value is a bound) for the grid sizes and may learn the true count as late as it likes.  Buffers xyzs / dirs / ts / sigmas /
 * rgbs need N rows (n_alive * n_step <= N for every round); N = the ray count of the image.  Per-ray arithmetic is that of n2m_march_rays /
 * n2m_composite_rays: same image, bit for bit. */
int n2m_march_rays_dev(const int32_t* state, uint32_t n_alive_ub, uint32_t N, const int32_t* rays_alive, const float* rays_t,
                       const float* rays_o, const float* rays_d, float bound, int contract, float dt_gamma, uint32_t max_steps, uint32_t C,
                       uint32_t H, const uint8_t* grid, const float* fars, float* xyzs, float* dirs, float* ts, const float* noises,
                       void* stream);
int n2m_composite_rays_dev(const int32_t* state, uint32_t n_alive_ub, uint32_t N, uint32_t max_steps, float T_thresh, int alpha_mode,
                           int32_t* rays_alive, float* rays_t, const float* sigmas, const float* rgbs, const float* ts, float* weights_sum,
                           float* depth, float* image, void* stream);
int n2m_compact_alive_dev(const int32_t* rays_alive, const int32_t* state, uint32_t n_alive_ub, uint32_t N, uint32_t max_steps, int32_t* out,
                          int32_t* state_out, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * gridencoder   (reference: gridencoder/src/gridencoder.h:12-15, gridencoder/src/bindings.cpp:5-9)
 * ---------------------------------------------------------------------------------------------------- */

/* gridencoder.h:12  grid_encode_forward   kernel gridencoder.cu:87-244, host :371-399,447-470.
 * inputs [B,D] fp32 in [0,1]; embeddings [rows,C] (dtype); offsets [L+1] int32 (device);
 * outputs [L,B,C] (dtype, LEVEL-MAJOR as in the reference); levels >= max_level are not written.
 * dy_dx: NULL or [B,L,D,C] (dtype).  S = log2(per_level_scale), H = base resolution.
 * gridtype 0 hash / 1 tiled; interp 0 linear / 1 smoothstep.  D in {2,3,4,5}, C in {1,2,4,8}. */
int n2m_grid_encode_forward(const float* inputs, const void* embeddings, const int32_t* offsets,
                            void* outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                            uint32_t max_level, float S, uint32_t H, void* dy_dx, uint32_t gridtype,
                            int align_corners, uint32_t interp, int dtype, void* stream);

/* gridencoder.h:13  grid_encode_backward   kernels gridencoder.cu:247-339 (+ :342-368), host :401-443,472-502.
 * grad [L,B,C] (dtype); grad_embeddings [rows,C] (dtype) is caller-zero-filled and accumulated with atomics;
 * dy_dx/grad_inputs: both NULL, or dy_dx [B,L,D,C] and grad_inputs [B,D] (dtype). */
int n2m_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings,
                             const int32_t* offsets, void* grad_embeddings, uint32_t B, uint32_t D,
                             uint32_t C, uint32_t L, uint32_t max_level, float S, uint32_t H,
                             const void* dy_dx, void* grad_inputs, uint32_t gridtype, int align_corners,
                             uint32_t interp, int dtype, void* stream);

/* gridencoder.h:15  grad_total_variation   kernel gridencoder.cu:505-609.
 * inputs [B,D] (dtype! the reference reads them as scalar_t, gridencoder.cu:507,642) in [0,1];
 * adds the TV gradient of the cell containing each input into `grad` [rows,C] in place (atomics). */
int n2m_grad_total_variation(const void* inputs, const void* embeddings, void* grad, const int32_t* offsets,
                             float weight, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                             uint32_t H, uint32_t gridtype, int align_corners, int dtype, void* stream);

/* New, opt-in layout variant used by the host operator (not by the reference's grid.py): identical
 * arithmetic to n2m_grid_encode_forward but writes outputs SAMPLE-MAJOR [B, L*C] directly, which is what
 * gridencoder/grid.py:63 produces with an extra permute pass.  Levels >= max_level are zero-filled. */
int n2m_grid_encode_forward_bm(const float* inputs, const void* embeddings, const int32_t* offsets,
                               void* outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                               uint32_t max_level, float S, uint32_t H, uint32_t gridtype,
                               int align_corners, uint32_t interp, int dtype, void* stream);
/* ... and the matching backward taking grad SAMPLE-MAJOR [B, L*C] (saves the permute of grid.py:81). */
int n2m_grid_encode_backward_bm(const void* grad, const float* inputs, const void* embeddings,
                                const int32_t* offsets, void* grad_embeddings, uint32_t B, uint32_t D,
                                uint32_t C, uint32_t L, uint32_t max_level, float S, uint32_t H,
                                uint32_t gridtype, int align_corners, uint32_t interp, int dtype,
                                void* stream);

/* New (no reference counterpart): the production backward of the host operator.  Same result as
 * n2m_grid_encode_backward / n2m_grad_total_variation (gridencoder.cu:247-339 / :505-609) for D = 3 and the two table
 * formats nerf2mesh trains with (fp32 C=1, fp16 C=2), computed by binning the per-vertex updates by table partition
 * and summing them in 64-bit fixed point: every (sample, level) is visited once, the sums are exact
 * from 2^-38 of the level's largest gradient down and, for the hashed levels, bit-reproducible run to run.
 *   host_offsets : HOST pointer to the L+1 int32 level offsets (the list gridencoder/grid.py:117-128 builds)
 *   workspace    : device scratch of at least n2m_grid_binned_workspace_bytes(...) bytes, 256-byte aligned; contents
 *                  are undefined on entry and exit, the calls are stream-ordered so one buffer can be shared
 *   B            : any; batches above 2^20 samples run in passes of 2^20 over the same workspace
 * n2m_grid_binned_workspace_bytes returns 0 when the configuration is not covered (callers then use the generic
 * entry points above). */
uint64_t n2m_grid_binned_workspace_bytes(uint32_t B, uint32_t D, uint32_t C, uint32_t max_level,
                                         const int32_t* host_offsets, int dtype, int tv);
int n2m_grid_encode_backward_binned(const void* grad, const float* inputs, const int32_t* host_offsets,
                                    void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                    uint32_t max_level, float S, uint32_t H, uint32_t gridtype, int align_corners,
                                    uint32_t interp, int dtype, const float* tv_embeddings, float tv_weight,
                                    float tv_weight_outer, float tv_inner01, const float* tv_scale,
                                    float* found_inf, void* workspace, uint64_t workspace_bytes, void* stream);
/* found_inf (device float, may be NULL) is set to 1 when a gradient read or a sum written is not finite (GradScaler's
 * check, done where the values pass through anyway; never cleared here).
 * tv_embeddings != NULL (fp32 C=1 table, max_level == L) folds grad_total_variation over the same inputs into the
 * backward: the TV cell floor(x*scale+0.5) is vertex 000 of the interpolation cell, so its term rides on that vertex's
 * update at no extra entry.  TV weighting, here and in n2m_grad_total_variation_binned: `weight` for inputs with
 * |x - 0.5|_inf <= inner01, `weight_outer` for the rest (nerf/utils.py:815-821 runs TV twice, x10 outside the unit box
 * when bound > 1; inner01 = 0.5/bound, >= 0.5 disables the split), both multiplied by *scale when that device pointer is
 * given (the GradScaler factor: the term can then be added to gradients that are still scaled). */
int n2m_grad_total_variation_binned(const float* inputs, const float* embeddings, float* grad,
                                    const int32_t* host_offsets, float weight, float weight_outer, float inner01,
                                    const float* weight_scale, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                    uint32_t H, uint32_t gridtype, int align_corners, void* workspace,
                                    uint64_t workspace_bytes, void* stream);

/* Forward of both field encoders in one launch (same inputs, same level geometry): embeddings1 [rows,1] f32 -> outputs1 [L,B]
 * f32, embeddings2 [rows,2] f16 -> outputs2 [L,B,2] f16, level-major; bit-identical to two n2m_grid_encode_forward calls.
 * Levels >= max_level are not written. */
int n2m_grid_encode_forward_pair(const float* inputs, const float* embeddings1, const void* embeddings2,
                                 const int32_t* offsets, float* outputs1, void* outputs2, uint32_t B, uint32_t L,
                                 uint32_t max_level, float S, uint32_t H, uint32_t gridtype, int align_corners,
                                 uint32_t interp, float in_scale, float in_offset, void* stream);
/* in_scale / in_offset (here and in the pair backward): the kernels read x * in_scale + in_offset, i.e. (1, 0) for inputs already
 * in [0,1], or (1/(2*bound), 0.5) to consume the renderer's points in [-bound, bound] directly -- for a power-of-two bound that is
 * bit-identical to grid.py:156's (x + bound) / (2 * bound) and saves two elementwise passes per call. */

/* n2m_grid_encode_forward_pair on a PACKED copy of the two tables: packed [rows] of 8 bytes {fp32 density feature, 2 x fp16 colour
 * features}, 16-byte aligned.  One L2 line then serves both encoders' gather of a vertex pair (the forward is bound by line traffic,
 * not by bytes used).  Outputs are bit-identical to the pair call.  n2m_adam_step keeps such a copy fresh (shadow modes 2 and 3). */
int n2m_grid_encode_forward_packed(const float* inputs, const void* packed, const int32_t* offsets, float* outputs1,
                                   void* outputs2, uint32_t B, uint32_t L, uint32_t max_level, float S, uint32_t H,
                                   uint32_t gridtype, int align_corners, uint32_t interp, float in_scale, float in_offset,
                                   void* stream);
/* n2m_grid_encode_forward_packed that also LEAVES, per hashed level and sample, the density column of corners 000 / 100 / 010 / 001 of the sample's
 * interpolation cell: tv_corners [L, B, 4] fp32, 16-byte aligned (rows of dense levels and of samples outside the unit cube are not written;
 * NULL = the plain call).  Those four values are the centre and the +x / +y / +z neighbours of the TV stencil of gridencoder.cu:505-609 (the TV
 * cell floor(x * scale + 0.5) is the interpolation cell's vertex 000), so a table backward over the SAME samples and the SAME table state that
 * folds the TV term in (tv_embeddings != NULL) can take them from here -- n2m_grid_backward_tv_corners -- and gather three neighbours per
 * (sample, level) instead of six: the fill's fine levels are bound by exactly those scattered requests.  Outputs bit-identical to the plain call. */
int n2m_grid_encode_forward_packed_tv(const float* inputs, const void* packed, const int32_t* offsets, float* outputs1,
                                      void* outputs2, uint32_t B, uint32_t L, uint32_t max_level, float S, uint32_t H,
                                      uint32_t gridtype, int align_corners, uint32_t interp, float in_scale, float in_offset,
                                      float* tv_corners, void* stream);
/* Round 6.  The packed lookup that also leaves the FINISHED total-variation terms of its samples: tv_out [L, B] fp32, tv_out[level, b] = the term
 * gridencoder.cu:505-609 (kernel_grad_tv) adds to the gradient of the cell of sample b on `level`, weighted like nerf/utils.py:800-823 (tv_weight inside
 * the inner region |x - 0.5| <= tv_inner01, tv_weight_outer outside, both times *tv_scale when given = the GradScaler factor) -- bit for bit what
 * n2m_grid_tv_terms computes from the packed table's density column.  Four of the stencil's seven values are corners 000 / 100 / 010 / 001 of the
 * interpolation cell, which the lookup holds in registers; at most three more rows are gathered.  n2m_grid_encode_backward_binned_pair_tvt
 * consumes tv_out: the table backward's fill then reads 4 coalesced bytes per (sample, level) instead of gathering the stencil.  max_level == L.
 * outputs1 / outputs2 are bit-identical to n2m_grid_encode_forward_packed's. */
int n2m_grid_encode_forward_packed_tvterms(const float* inputs, const void* packed, const int32_t* offsets, float* outputs1,
                                           void* outputs2, uint32_t B, uint32_t L, uint32_t max_level, float S, uint32_t H,
                                           uint32_t gridtype, int align_corners, uint32_t interp, float in_scale, float in_offset,
                                           float tv_weight, float tv_weight_outer, float tv_inner01, const float* tv_scale,
                                           float* tv_out, void* stream);
/* Sticky, per thread; NULL clears it.  The corner records the next n2m_grid_encode_backward_binned_pair[_half] calls of this thread may use for
 * their folded TV term (see above).  Used when the call computes the term itself, runs in one pass (B <= 2^20) over one point list in input
 * order without folded copies; ignored otherwise.  The caller guarantees: same B, same inputs, same table values as the forward that wrote them.
 * The gradients are bit-identical with and without the records. */
int n2m_grid_backward_tv_corners(const float* tv_corners);

/* The same lookup for the levels [level_begin, level_begin + n_levels) only (the rows of outputs1 / outputs2 of those levels, bit-identical
 * to the full call's).  For a caller whose packed rows arrive in level chunks -- multi-GPU with the optimizer sharded over the ranks
 * (nerf/utils.py:517-519 is the reference's DDP wrapper): every rank refreshes its own rows, the others' arrive by all-gather,
 * coarse levels first, and the lookup of the half that has landed runs while the other half is still on the wire. */
int n2m_grid_encode_forward_packed_levels(const float* inputs, const void* packed, const int32_t* offsets, float* outputs1,
                                          void* outputs2, uint32_t B, uint32_t L, uint32_t max_level, float S, uint32_t H,
                                          uint32_t gridtype, int align_corners, uint32_t interp, float in_scale, float in_offset,
                                          uint32_t level_begin, uint32_t n_levels, void* stream);

/* Both encoders of nerf2mesh's field in one call: the density table (fp32, C=1) and the colour table (fp16, C=2) share
 * their level geometry and are queried at the same inputs (nerf/network.py:92-108), so the index arithmetic and the partition
 * sort of the binned backward are done once and two update logs are written.  grad1 [L,B] f32, grad2 [L,B,2] f16, one
 * host_offsets for both tables; TV (tv_embeddings = the fp32 table) and found_inf as in n2m_grid_encode_backward_binned.
 * Entries of consecutive samples that fall into the same cell of a coarse level (samples along a ray do) are summed in fp32 before
 * they enter the exact fixed-point sum -- the log halves; the result differs from the unmerged sum by the rounding of those
 * partial sums (fp16 table: of sums of the half-rounded products), far below the order noise of the reference's atomics.
 * overwrite = 0: the sums are added onto grad_embeddings1/2 (zero-filled by the caller, or running sums) like the reference's
 * atomicAdd (gridencoder.cu:324-334); overwrite = 1: the call DEFINES both tables completely (all host_offsets[L] rows, zeros
 * included) -- no zero-fill beforehand and no read-modify-write in the flush.  * grad1 == NULL together with grad_embeddings1 == NULL: the COLOUR table alone through the same kernels (stage 1 shades with the colour
 * networks only, nerf/renderer.py:875-881): no density gradient is read, no fp32 log written, the fp32 accumulate not launched; no TV.
 * grad2 == NULL together with grad_embeddings2 == NULL: the DENSITY table alone (the stacked finite-difference evaluations of the SDF head,
 * nerf/network.py:143-154), TV allowed. */
uint64_t n2m_grid_binned_pair_workspace_bytes(uint32_t B, uint32_t max_level, const int32_t* host_offsets);
int n2m_grid_encode_backward_binned_pair(const float* grad1, const void* grad2, const float* inputs,
                                         const int32_t* host_offsets, float* grad_embeddings1,
                                         void* grad_embeddings2, uint32_t B, uint32_t L, uint32_t max_level, float S,
                                         uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                                         const float* tv_embeddings, float tv_weight, float tv_weight_outer,
                                         float tv_inner01, const float* tv_scale, float* found_inf, float in_scale,
                                         float in_offset, int overwrite, void* workspace, uint64_t workspace_bytes,
                                         void* stream);
/* The same for ONE half of the levels (max_level == L == 16 only): half = 1 -> levels 8..15, half = 2 -> levels 0..7; the two calls
 * together equal the full call bit for bit (levels are independent).  For callers that exchange the table gradients between GPUs: the
 * rows of the fine half are final after the first call, their collective can run while the second call computes (engine.py). */
int n2m_grid_encode_backward_binned_pair_half(const float* grad1, const void* grad2, const float* inputs,
                                         const int32_t* host_offsets, float* grad_embeddings1,
                                         void* grad_embeddings2, uint32_t B, uint32_t L, uint32_t max_level, float S,
                                         uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                                         const float* tv_embeddings, float tv_weight, float tv_weight_outer,
                                         float tv_inner01, const float* tv_scale, float* found_inf, float in_scale,
                                         float in_offset, int overwrite, void* workspace, uint64_t workspace_bytes,
                                         void* stream, int half);

/* The table backward with the OPTIMIZER PASS of the hashed levels inside it.  On every level whose partitions one accumulate work item owns
 * (n2m_grid_pair_fuse_plan: all levels from first_level on, for batches up to max_samples) the flush holds the final gradient row in registers;
 * n2m_grid_encode_backward_binned_pair_adam lets it do what n2m_adam_step would do for that row -- same arithmetic, element for element --
 * instead of storing the gradient for a later pass: the 98 MB gradient round trip disappears.  MEASURED SLOWER than the separate pass on
 * MI355X (the accumulates move the optimizer state at a third of n2m_adam_step's rate: step 0.632 -> 0.682 ms); the
 * engine uses it only with N2M_FUSE_ADAM=1.  GradScaler's skip is all-or-nothing and the verdict is known only when the last item has flushed, so
 * the update is written BESIDE the old state: parameter and both moments of the two tables exist twice, the flush reads the *_in buffers and
 * writes the *_out buffers (and its column of the packed table).  Rows below first_row (small dense levels, split over tile groups) and all
 * other tensors keep n2m_adam_step, in place on the *_in buffers.  n2m_adam_fuse_restore, launched behind BOTH (and in front of the scaler
 * update that clears found_inf), then copies rows [0, first_row) from *_in to *_out and, when found_inf is set, the remaining rows too
 * (re-packing the old parameters): after it the *_out buffers hold the complete state of the step -- taken or skipped -- and the caller
 * swaps the roles of the two buffer sets without reading the verdict.
 * [0] = density table (C = 1, fp32 gradient), [1] = colour table (C = 2, gradient rounded to fp16 like the unfused path's). */
typedef struct {
    const float* p_in[2]; const float* m_in[2]; const float* v_in[2];      /* live buffers, whole tables: element row * C + c */
    float* p_out[2]; float* m_out[2]; float* v_out[2];                    /* the other set */
    void* packed;                                                         /* the packed table the lookup reads (all rows) */
    uint32_t first_level;
    float lr[2]; int32_t slot[2];                                          /* learning rate / step-count slot (bias[2 slot], bias[2 slot + 1]) per table */
    double beta1, beta2; float eps;
    const float* scale; const float* bias;                                /* device: loss scale (may be NULL), bias corrections as n2m_adam_step */
} N2mAdamFuse;
int n2m_grid_pair_fuse_plan(uint32_t max_samples, uint32_t L, const int32_t* host_offsets, uint32_t* first_level, uint32_t* first_row);
int n2m_grid_encode_backward_binned_pair_adam(const float* grad1, const void* grad2, const float* inputs, const int32_t* host_offsets,
                                              float* grad_embeddings1, void* grad_embeddings2, uint32_t B, uint32_t L, uint32_t max_level, float S,
                                              uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, const float* tv_embeddings,
                                              float tv_weight, float tv_weight_outer, float tv_inner01, const float* tv_scale, float* found_inf,
                                              float in_scale, float in_offset, void* workspace, uint64_t workspace_bytes, const N2mAdamFuse* fuse,
                                              void* stream);
int n2m_adam_fuse_restore(const N2mAdamFuse* fuse, const int32_t* host_offsets, uint32_t L, const float* found_inf, void* stream);

/* SDF recipe (nerf/network.py:143-154: normals by finite differences, six copies x +- eps e_a of every sample through the density encoder):
 * the copies' TABLE BACKWARD folded into the batch's own.  At the end of the schedule eps = 1e-4 is a tenth of the finest cell, so on a given
 * level a copy nearly always shares its centre sample's cell (99.9 % on level 0 ... 90 % on level 15) -- it updates the same eight rows with
 * slightly different weights.
 *   n2m_sdf_fold_plan    flags[l, m] bit c (c = 2 axis + (minus ? 1 : 0), n2m_sdf_offsets' order): copy c of sample m lies in the centre's cell
 *                        on level l (0 for a centre outside the unit cube); per level the OTHER copies as a compact list -- left_pts01
 *                        [L, cap, 3] ([0,1] coordinates as n2m_sdf_offsets writes them), left_src [L, cap] (= 6 m + c), their number added to
 *                        counters[parity][l] (counters [2][32]; the kernel clears counters[parity ^ 1]: alternate parity from step to step;
 *                        cap >= 6 M).
 *   n2m_sdf_fold_gather  out [levels, B] = grad6[level, left_src[level, j]] for the listed copies (grad6 = the stacked copies' feature gradient
 *                        [levels, 6 M]; counters = this step's row), zero beyond a level's list, whose points are set outside the unit
 *                        cube up to B (the caller's common list length, >= every counter): ready for n2m_grid_encode_backward_binned_lists.
 *   n2m_grid_encode_backward_binned_pair_fold   n2m_grid_encode_backward_binned_pair (one pass, B <= 2^20) that adds, for every (copy, level)
 *                        with its flag set, w_copy(corner) * grad6[level, 6 s + c] to the centre sample's eight density entries (weights from
 *                        the copy's own position, recomputed as n2m_sdf_offsets does: the values the stacked pass would have used).
 *   n2m_grid_encode_backward_binned_lists       the density-only backward with one point list PER LEVEL (inputs [L, level_stride, 3], grad1
 *                        [L, B]); points outside the unit cube are skipped.
 * The caller reads the row of counters back (they are final long before the backward is enqueued: the plan only needs the samples) and runs the
 * lists call with B = their maximum.  Sums equal the stacked pass's up to fp32 association (centre + copies in one thread instead of a lane scan). */
int n2m_sdf_fold_plan(const float* xyz, uint32_t M, float eps, float bound, uint32_t L, uint32_t max_level, float S, uint32_t H, int align_corners,
                      uint8_t* flags, float* left_pts01, uint32_t* left_src, uint32_t cap, uint32_t* counters, uint32_t parity, void* stream);
int n2m_sdf_fold_gather(const float* grad6, uint32_t M, uint32_t levels, const uint32_t* left_src, float* left_pts01, uint32_t cap,
                        const uint32_t* counters, uint32_t B, float* out, void* stream);
int n2m_grid_encode_backward_binned_pair_fold(const float* grad1, const void* grad2, const float* inputs, const int32_t* host_offsets,
                                              float* grad_embeddings1, void* grad_embeddings2, uint32_t B, uint32_t L, uint32_t max_level, float S,
                                              uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, const float* tv_embeddings,
                                              float tv_weight, float tv_weight_outer, float tv_inner01, const float* tv_scale, float* found_inf,
                                              float in_scale, float in_offset, int overwrite, void* workspace, uint64_t workspace_bytes,
                                              const uint8_t* fold_flags, const float* fold_grad6, float fold_eps, float fold_bound, void* stream);
int n2m_grid_encode_backward_binned_lists(const float* grad1, const float* inputs, uint32_t level_stride, const int32_t* host_offsets,
                                          float* grad_embeddings1, uint32_t B, uint32_t L, uint32_t max_level, float S, uint32_t H, uint32_t gridtype,
                                          int align_corners, uint32_t interp, float* found_inf, int overwrite, void* workspace,
                                          uint64_t workspace_bytes, void* stream);

/* The TV terms of a batch on their own: tv_out[level, s] (f32 [L, B]) = the total-variation term of (sample s, level) exactly as the shared
 * fill evaluates it in place when it is handed tv_embeddings (gridencoder.cu:505-609 on the cell floor(x * scale + 0.5); weight / weight_outer /
 * inner01 / scale as in n2m_grid_encode_backward_binned_pair) -- and n2m_grid_encode_backward_binned_pair_tvt, the same backward consuming
 * those terms instead of gathering the 7-point stencil itself (tv_terms [L, B]; half: 0 all levels, 1 levels 8..15, 2 levels 0..7).  The
 * terms depend on the samples and the density table only, so a training step can evaluate them on a second stream beside its field kernels:
 * the stencil's scattered gathers (45 us of the fill) leave the backward's critical path.  Results are bit-identical to the in-place form. */
int n2m_grid_tv_terms(const float* inputs, const float* tv_embeddings, const int32_t* host_offsets, uint32_t B, uint32_t L, float S, uint32_t H,
                      uint32_t gridtype, int align_corners, uint32_t interp, float tv_weight, float tv_weight_outer, float tv_inner01,
                      const float* tv_scale, float in_scale, float in_offset, float* tv_out, void* stream);
int n2m_grid_encode_backward_binned_pair_tvt(const float* grad1, const void* grad2, const float* inputs, const int32_t* host_offsets,
                                             float* grad_embeddings1, void* grad_embeddings2, uint32_t B, uint32_t L, uint32_t max_level, float S,
                                             uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, const float* tv_terms,
                                             float* found_inf, float in_scale, float in_offset, int overwrite, void* workspace,
                                             uint64_t workspace_bytes, void* stream, int half);

/* ------------------------------------------------------------------------------------------------------
 * marching cubes   (reference: `mcubes.marching_cubes(volume, isovalue)` -- PyMCubes, an un-vendored dependency; call sites
 * nerf/renderer.py:524-527 (stage-0 mesh), :563 and :616 (outer cascades).  SURVEY.md section 8f-4.)
 * The reference copies the volume to the host (:518) and runs PyMCubes there; here the volume stays in HBM.
 * volume [R0][R1][R2] f32 in C order; solid = !(value < iso) (PyMCubes marks value < iso: same surface); one vertex per crossed grid
 * edge at PyMCubes' interpolation x1 + (iso - f1) / (f2 - f1), evaluated in double with the lower corner first; triangles from the
 * 256-case table of nerf2mesh_amd/csrc/mc_table.inc (rule-generated, tools/gen_mc_table.py: PyMCubes' literal table is not available
 * here; ambiguous faces never join their solid corners, which makes every extracted surface closed), normals pointing towards lower
 * values.  Deterministic order: vertices by grid node (C order), then axis; triangles by cell (C order), then table order.
 * Two calls with a host read in between, like march_rays_train: count -> totals -> allocate -> emit.
 * ---------------------------------------------------------------------------------------------------- */

/* bytes of device scratch n2m_marching_cubes_count/_emit need for a volume (0: unsupported size, see _count) */
uint64_t n2m_marching_cubes_workspace_bytes(uint32_t R0, uint32_t R1, uint32_t R2);

/* Pass 1.  totals (device, 2 x u64) <- {number of vertices, number of triangles}.  Fills `workspace` for the emit pass.
 * Each dimension <= 2048 and fewer than 2^31 nodes; an emit needs totals[0] < 2^29 and totals[1] < 2^32. */
int n2m_marching_cubes_count(const float* volume, uint32_t R0, uint32_t R1, uint32_t R2, double iso, void* workspace,
                             uint64_t workspace_bytes, uint64_t* totals, void* stream);

/* Pass 2 (same volume, iso and workspace).  vertices [cap_v][3] (f32, or f64 when vertices_f64 != 0) <-
 * ((index-space position / div) * mul) + add, evaluated in double (the reference: `vertices / (resolution - 1.0) * 2 - 1` on
 * PyMCubes' doubles, then astype(float32), nerf/renderer.py:529-530); div = mul = 1, add = 0 gives PyMCubes' own coordinates.
 * triangles [cap_t][3] i32.  Entries beyond a capacity are dropped (pass the totals of pass 1). */
int n2m_marching_cubes_emit(const float* volume, uint32_t R0, uint32_t R1, uint32_t R2, double iso, const void* workspace,
                            uint64_t workspace_bytes, double div, double mul, double add, void* vertices, int vertices_f64,
                            uint32_t cap_v, int32_t* triangles, uint32_t cap_t, void* stream);

/* Texture-bake padding (reference: the host-side kd-tree fill of nerf/renderer.py:371-387, `NearestNeighbors(n_neighbors=1)` over texel
 * coordinates).  feats [H][W][C] u8, in place; role [H][W] u8: bit 0 = source texel (chart boundary ring), bit 1 = destination texel
 * (the band around the charts).  Every destination takes the features of the nearest source within `radius` texels (Euclidean on
 * (row, column); ties: smallest row, then column); destinations with no source in range are left as they are. */
int n2m_texture_pad_nearest(uint8_t* feats, const uint8_t* role, uint32_t H, uint32_t W, uint32_t C, uint32_t radius, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * freqencoder   (reference: freqencoder/src/freqencoder.h:6-10, freqencoder/src/bindings.cpp:5-8)
 * ---------------------------------------------------------------------------------------------------- */

/* freqencoder.h:7  freq_encode_forward   kernel freqencoder.cu:30-61.  inputs [B,D] f32 -> outputs [B,C] f32,
 * C = D + 2*deg*D: the input, then per frequency f: sin(2^f x) and sin(2^f x + pi/2). */
int n2m_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                            float* outputs, void* stream);
/* freqencoder.h:10  freq_encode_backward   kernel freqencoder.cu:66-94.  grad, outputs [B,C] -> grad_inputs [B,D] (written). */
int n2m_freq_encode_backward(const float* grad, const float* outputs, uint32_t B, uint32_t D, uint32_t deg,
                             uint32_t C, float* grad_inputs, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * training-step helpers (no reference kernel: the reference composes these from torch ops)
 * ---------------------------------------------------------------------------------------------------- */

/* Rays of a batch of (view, pixel) pairs, get_rays of nerf/utils.py:242-290 for explicit indices plus the ground-truth
 * gather of nerf/provider.py:330: poses [V,4,4] f32 (camera-to-world), cam [N] / pix [N] int64 (pix = j*W + i),
 * pinhole intrinsics; rays_d = R * ((i+0.5-cx)/fx, -(j+0.5-cy)/fy, -1) (un-normalised), rays_o = translation;
 * images [V,H*W,4] f32 with rgba [N,4] out, or both NULL. */
int n2m_get_rays(const float* poses, const int64_t* cam, const int64_t* pix, uint32_t N, uint32_t H, uint32_t W,
                 float fx, float fy, float cx, float cy, const float* images, float* rays_o, float* rays_d,
                 float* rgba, void* stream);

/* A whole training batch from ONE tensor of uniforms [N,6] in [0,1): view = floor(u0 V), pixel = floor(u1 H W) (random pixels over
 * random views, nerf/provider.py:302-303, nerf/utils.py:271), rays + ground truth as n2m_get_rays, near/far as n2m_near_far_from_aabb,
 * march jitter noises [N] = u2, random background bg [N,3] = u3..u5 (may be NULL), and *counter = 0 (may be NULL).  H*W < 2^24. */
int n2m_batch_rays(const float* poses, const float* uniforms, uint32_t V, uint32_t N, uint32_t H, uint32_t W, float fx, float fy, float cx,
                   float cy, const float* images, const float* aabb, float min_near, float* rays_o, float* rays_d, float* rgba,
                   float* nears, float* fars, float* noises, float* bg, int32_t* counter, void* stream);
/* The same with the per-view near / far clamp of `--enable_cam_near_far` (nerf/renderer.py:689-691; colmap_provider.py:270,563-565 hands
 * every ray the (near, far) of its camera): cam_near_far [V,2] f32, nears = max(nears, cnf[view,0]), fars = min(fars, cnf[view,1]); NULL =
 * n2m_batch_rays. */
int n2m_batch_rays_cnf(const float* poses, const float* uniforms, uint32_t V, uint32_t N, uint32_t H, uint32_t W, float fx, float fy, float cx,
                       float cy, const float* images, const float* aabb, float min_near, float* rays_o, float* rays_d, float* rgba,
                       float* nears, float* fars, float* noises, float* bg, int32_t* counter, const float* cam_near_far, void* stream);

/* Photometric loss head of the stage-0 step in one launch per direction:
 *   pred   = image + (1 - weights_sum) * bg                      nerf/renderer.py:747
 *   target = gt.rgb * gt.a + bg * (1 - gt.a)                     nerf/utils.py:663-664
 *   loss   = mean_r [ lambda_rgb * mean_c (pred - target)^2 + lambda_mask * (weights_sum - gt.a)^2 ]   :679-683,797
 * image [N,3] (composited colour BEFORE the background blend), weights_sum [N], gt_rgba [N,4]; bg [N,3] or NULL
 * (then bg_scalar is the uniform background, 1 = white).  partial: scratch of ceil(N/256) floats; ticket: one
 * uint32 that is zero on entry (left zero on exit); loss: [1].  The sum order is fixed, so the value is reproducible.
 * backward: grad_loss [1] (device; the GradScaler factor) -> d_image [N,3], d_weights_sum [N]. */
int n2m_photo_loss_forward(const float* image, const float* weights_sum, const float* gt_rgba, const float* bg,
                           float bg_scalar, float lambda_rgb, float lambda_mask, uint32_t N, float* partial,
                           uint32_t* ticket, float* loss, void* stream);
int n2m_photo_loss_backward(const float* image, const float* weights_sum, const float* gt_rgba, const float* bg,
                            float bg_scalar, float lambda_rgb, float lambda_mask, uint32_t N,
                            const float* grad_loss, float* d_image, float* d_weights_sum, void* stream);

/* Training fast path: n2m_composite_rays_train_forward -> n2m_photo_loss_forward -> n2m_photo_loss_backward ->
 * n2m_composite_rays_train_backward as ONE launch (density mode, the plain rgb + mask loss: no gradient into weights or depth).
 * A wave composites its ray, forms the ray's loss term and gradients and runs the backward scan right away (the seed gradient
 * *grad_loss / N does not depend on the loss value).  grad_sigmas [M] / grad_rgbs [M,3] are bit-identical to the four-call chain
 * (every sample of a ray's range is written); weights_sum [N] and image [N,3] (colour before the background blend) may be NULL;
 * partial: ceil(N/16) floats scratch; ticket: one zero uint32 (left zero); loss [1] = the mean over rays; loss_sum (may be NULL) += loss.
 * ticket == NULL: the kernel only leaves the per-workgroup partials (no fence, no arrival atomics); n2m_scaler_update_slots_loss sums them. */
int n2m_composite_loss_train(const float* sigmas, const float* rgbs, const float* ts, const int32_t* rays, uint32_t M, uint32_t N,
                             float T_thresh, const float* gt_rgba, const float* bg, float bg_scalar, float lambda_rgb, float lambda_mask,
                             const float* grad_loss, float* weights_sum, float* image, float* grad_sigmas, float* grad_rgbs, float* partial,
                             uint32_t* ticket, float* loss, float* loss_sum, void* stream);
/* The same + the entropy regulariser of nerf/utils.py:728-733 (config 4: `--lambda_entropy 1e-3`, scripts/runall_360_outdoor.sh:2):
 * loss += lambda_entropy * (mean over the M samples of H(clamp(w)) + mean over the N rays of H(clamp(weights_sum))),
 * H(p) = -p log2 p - (1 - p) log2 (1 - p), clamp to [1e-5, 1 - 1e-5].  Its gradient w.r.t. a sample's weight is the `grad_weights` input of
 * composite_rays_train's backward and enters grad_sigmas exactly where the reference kernel reads grad_weights[i] (raymarching.cu:676).
 * lambda_entropy == 0: identical to n2m_composite_loss_train. */
int n2m_composite_loss_train_ent(const float* sigmas, const float* rgbs, const float* ts, const int32_t* rays, uint32_t M, uint32_t N,
                                 float T_thresh, const float* gt_rgba, const float* bg, float bg_scalar, float lambda_rgb, float lambda_mask,
                                 const float* grad_loss, float* weights_sum, float* image, float* grad_sigmas, float* grad_rgbs, float* partial,
                                 uint32_t* ticket, float* loss, float* loss_sum, float lambda_entropy, void* stream);
/* ... and with alpha_mode != 0 the SDF recipe's compositing (raymarching.cu:534,671: `sigmas` ARE the alphas, backward scale 1 / (1 - alpha);
 * nerf/renderer.py:739-741).  The entropy term is not available in alpha mode. */
int n2m_composite_loss_train_ex(const float* sigmas, const float* rgbs, const float* ts, const int32_t* rays, uint32_t M, uint32_t N,
                                float T_thresh, const float* gt_rgba, const float* bg, float bg_scalar, float lambda_rgb, float lambda_mask,
                                const float* grad_loss, float* weights_sum, float* image, float* grad_sigmas, float* grad_rgbs, float* partial,
                                uint32_t* ticket, float* loss, float* loss_sum, float lambda_entropy, int alpha_mode, void* stream);

/* Live-first sample order for the table backward (no reference counterpart; raymarching.cu:553,640: composite_rays_train stops a ray at
 * T < T_thresh, every later sample of the ray gets weight 0 and gradient 0 -- about half of the samples of a trained batch).
 * n2m_composite_live_counts: optional outputs of the NEXT n2m_composite_loss_train* calls of this thread (sticky; clear with NULL, NULL):
 *   live [N] <- per ray the number of samples up to and including the one the early stop fell on, block_live [ceil(N/16)] <- their sums per 16 rays.
 * n2m_sample_order_live_first: perm [M] <- sample indices with every ray's live prefix first (ray order), then every ray's remaining samples.
 * n2m_grid_backward_sample_order (include below, gridencoder): the binned pair backward then visits the samples in that order. */
int n2m_composite_live_counts(int32_t* live, uint32_t* block_live);
int n2m_sample_order_live_first(const int32_t* rays, const int32_t* live, const uint32_t* block_live, uint32_t N, uint32_t M,
                                uint32_t* perm, void* stream);
/* Sticky, per thread; NULL clears it.  perm [B]: the i-th sample the next n2m_grid_encode_backward_binned_pair[_half|_tvt] calls visit.  The
 * result is the same sum (fixed point) whatever the order; only the fp32 rounding of the same-cell run merge on the coarse levels follows it.
 * Waves whose 64 samples all have zero feature gradients on a level take a TV-only path (one log entry per sample instead of eight).
 * Needs the partition-major path, one pass (B <= 2^20), no folded copies, no per-level point lists. */
int n2m_grid_backward_sample_order(const uint32_t* perm);

/* Sticky, per thread; 0 restores the default (9).  The number of levels, from the coarsest, on which the fill of the next
 * n2m_grid_encode_backward_binned_pair* calls (partition-major path) merges same-cell runs of consecutive samples into one log entry per vertex.
 * Marched samples stop sharing cells around level 9; the covered pixels of a rasterised frame (stage 1) share them on every level: engine_stage1
 * sets 16 around its call.  The sums are the same fixed-point sums; a merged run is added in fp32 before it enters them. */
int n2m_grid_backward_merge_levels(uint32_t levels);

/* Adam + GradScaler for the whole parameter set in two launches (torch.optim.Adam(fused=True) + torch.amp.GradScaler of
 * main.py:221 / nerf/utils.py:506,1187-1190).  All tensors fp32 and 16-byte aligned, except grad which may be fp16
 * (grad_is_half) and half_shadow (fp16 copy of the updated parameter, or NULL).  Gradients are still multiplied by *scale
 * (NULL = 1); no parameter or moment is touched when *found_inf != 0; bias [2] = (1 - beta1^t, sqrt(1 - beta2^t)) of this step, kept up to
 * date by n2m_scaler_update (initialise it to (1 - beta1, sqrt(1 - beta2)) for t = 1). */
#define N2M_ADAM_MAX 16
typedef struct {
    void* param[N2M_ADAM_MAX]; const void* grad[N2M_ADAM_MAX]; void* exp_avg[N2M_ADAM_MAX]; void* exp_avg_sq[N2M_ADAM_MAX];
    void* half_shadow[N2M_ADAM_MAX];
    uint32_t numel[N2M_ADAM_MAX]; float lr[N2M_ADAM_MAX]; int32_t grad_is_half[N2M_ADAM_MAX];
    int32_t shadow_mode[N2M_ADAM_MAX];   /* 1: half_shadow is a plain fp16 copy; 2: parameter is a [rows,1] fp32 table, written to column 0
                                            of a packed table (8 B rows); 3: parameter is a [rows,2] table, written as half2 to column 1 */
    int32_t clear_grad[N2M_ADAM_MAX];    /* 1: the (fp32) gradient is set to zero once read -- also when the step is skipped -- so that a
                                            producer adding into a persistent buffer needs no zero-fill launch of its own */
    int32_t slot[N2M_ADAM_MAX];          /* 0: bias corrections bias[0..1] (one step count for all tensors); s in 1..N2M_ADAM_MAX: this
                                            tensor's own count, corrections bias[2s..2s+1] (n2m_scaler_update_slots) */
    uint32_t count;
} N2mAdamDesc;   /* HOST struct */
int n2m_adam_step(const N2mAdamDesc* desc, double beta1, double beta2, float eps, const float* scale,
                  const float* found_inf, const float* bias, void* stream);
/* torch_ema.ExponentialMovingAverage.update() (torch-ema is an un-vendored dependency of the reference, requirements.txt:11, imported at
 * nerf/utils.py:29; the Trainer builds it over model.parameters() with decay 0.95 for stage 0 -- nerf/utils.py:544-545, main.py:241 -- and
 * updates it once per epoch, :1213-1214) for up to N2M_EMA_MAX fp32 tensors in one launch:
 *     shadow[i] = shadow[i] - (shadow[i] - param[i]) * one_minus_decay
 * in that association and without FMA contraction, i.e. bit-identical to the library's `tmp = s - p; tmp.mul_(1 - decay); s.sub_(tmp)`.
 * The decay schedule (min(decay, (1 + n) / (10 + n)) at the n-th update) is host logic: nerf2mesh_amd/ema.py. */
#define N2M_EMA_MAX 16
typedef struct {
    void* shadow[N2M_EMA_MAX]; const void* param[N2M_EMA_MAX];
    uint32_t numel[N2M_EMA_MAX];
    uint32_t count;
} N2mEmaDesc;   /* HOST struct */
int n2m_ema_update(const N2mEmaDesc* desc, float one_minus_decay, void* stream);

/* GradScaler.update() on device scalars (floats): found_inf != 0 -> scale *= backoff, tracker = 0; else step += 1,
 * tracker += 1 and scale *= growth every growth_interval good steps; found_inf is reset to 0 and bias [2] recomputed (in
 * double) for the next step.  scale / growth_tracker / step / bias may be NULL. */
int n2m_scaler_update(float* scale, float* growth_tracker, float* found_inf, float* step, float* bias, double beta1,
                      double beta2, float growth_factor, float backoff_factor, float growth_interval, void* stream);
/* The same with one step count per tensor slot, as torch.optim.Adam counts (state[p]["step"]: a parameter that receives its first
 * gradient late starts its bias corrections at t = 1).  steps [1 + N2M_ADAM_MAX] and bias [1 + N2M_ADAM_MAX][2], slot 0 = the
 * global count; participants: bit s-1 set = slot s took part in this step.  Initialise every bias row to (1-beta1, sqrt(1-beta2)). */
int n2m_scaler_update_slots(float* scale, float* growth_tracker, float* found_inf, float* steps, float* bias,
                            uint32_t participants, double beta1, double beta2, float growth_factor, float backoff_factor,
                            float growth_interval, void* stream);

/* n2m_scaler_update_slots + loss = sum(loss_partial[0 .. n_partial)) / n_rays, *loss_sum += loss (either may be NULL). */
int n2m_scaler_update_slots_loss(float* scale, float* growth_tracker, float* found_inf, float* steps, float* bias,
                                 uint32_t participants, double beta1, double beta2, float growth_factor, float backoff_factor,
                                 float growth_interval, const float* loss_partial, uint32_t n_partial, uint32_t n_rays, float* loss,
                                 float* loss_sum, void* stream);
/* The same + a second term with its own normalisation: loss += extra_scale * sum(extra_partial[0 .. n_extra)).  The training step's
 * specular regulariser lambda_specular * mean_m sum_c specular^2 (nerf/utils.py:733-737) arrives this way: extra_partial = the
 * per-workgroup sums n2m_field_forward_train leaves, extra_scale = lambda_specular / M.  extra_partial == NULL: no second term. */
int n2m_scaler_update_slots_loss2(float* scale, float* growth_tracker, float* found_inf, float* steps, float* bias,
                                  uint32_t participants, double beta1, double beta2, float growth_factor, float backoff_factor,
                                  float growth_interval, const float* loss_partial, uint32_t n_partial, uint32_t n_rays, float* loss,
                                  float* loss_sum, const float* extra_partial, uint32_t n_extra, float extra_scale, void* stream);
/* ... and a third term, loss += extra2_scale * sum(extra2_partial[0 .. n_extra2)) (SDF recipe: the eikonal loss from n2m_sdf_alpha_forward's partials). */
int n2m_scaler_update_slots_loss3(float* scale, float* growth_tracker, float* found_inf, float* steps, float* bias,
                                  uint32_t participants, double beta1, double beta2, float growth_factor, float backoff_factor,
                                  float growth_interval, const float* loss_partial, uint32_t n_partial, uint32_t n_rays, float* loss,
                                  float* loss_sum, const float* extra_partial, uint32_t n_extra, float extra_scale,
                                  const float* extra2_partial, uint32_t n_extra2, float extra2_scale, void* stream);

/* n2m_adam_step + n2m_scaler_update_slots_loss3 in ONE launch (round 6): the optimizer pass whose last workgroup to finish also does the
 * GradScaler / per-slot step count / bias-correction / loss-value bookkeeping (GradScaler.update of nerf/utils.py:1176-1177 behind
 * optimizer.step; main.py:221's torch.optim.Adam semantics as n2m_adam_step) -- the same code on the same 256 threads, bit-identical state and
 * loss value; scale / found_inf / bias are the arrays n2m_adam_step reads (here also written: the next step's values).  `ticket`: N2M_TAIL_TICKET_WORDS
 * uint32 of device memory (64 arrival counters, one 128-byte line each), zero before the first call; the launch leaves them zero.  One launch less on the step's critical path. */
#define N2M_TAIL_TICKET_WORDS (64 * 32)
typedef struct {
    float* growth_tracker; float* steps;
    uint32_t participants; float growth_factor, backoff_factor, growth_interval;
    const float* loss_partial; uint32_t n_partial; uint32_t n_rays; float* loss; float* loss_sum;
    const float* extra_partial; uint32_t n_extra; float extra_scale;           /* NULL: no second term */
    const float* extra2_partial; uint32_t n_extra2; float extra2_scale;        /* NULL: no third term */
    uint32_t* ticket;
} N2mScalerTail;   /* HOST struct */
int n2m_adam_step_scaler(const N2mAdamDesc* desc, double beta1, double beta2, float eps, float* scale, float* found_inf, float* bias,
                         const N2mScalerTail* tail, void* stream);

/* SDF head of the step executor (config 5; the caller-side arithmetic of nerf/renderer.py:724-739, nerf/network.py:143-154 and the eikonal
 * loss of nerf/utils.py:740-743 -- the torch statement in nerf2mesh_amd/{renderer,network}.py is the parity baseline):
 *  n2m_sdf_offsets        pts [M, 6, 3] = clamp(xyz +- eps e_axis, -bound, bound) (k = 2 axis + (minus ? 1 : 0)), pts01 = (pts + bound) / (2 bound);
 *                         SAMPLE-major: the six copies of a sample are adjacent (they share their cell on almost every level)
 *  n2m_sdf_alpha_forward  sdf [M], sdf6 [M, 6] (the field at pts), dirs [M,3] (raw ray directions), ts [M,2], variance (device scalar)
 *                         -> alpha [M] (NeuS-style, clipped to [0,1]), normal [M,3] (raw finite-difference normal, may be NULL), eik_partial
 *                            [ceil(M/256)] = per-workgroup sums of (|normal| - 1)^2 (may be NULL)
 *  n2m_sdf_alpha_backward d_alpha [M] -> d_sdf [M], d_sdf6 [M, 6], d_variance [1] (= sum of var_partial [ceil(M/256)], fixed order; found_inf is
 *                         raised when it is not finite); adds the eikonal term's gradient *seed * eik_coef * (|n| - 1) n / |n| with
 *                         eik_coef = lambda_eikonal * 2 / M on the host. */
int n2m_sdf_offsets(const float* xyz, uint32_t M, float eps, float bound, float* pts, float* pts01, void* stream);
int n2m_sdf_alpha_forward(const float* sdf, const float* sdf6, const float* dirs, const float* ts, uint32_t M, const float* variance, float eps,
                          float cos_anneal_ratio, float* alpha, float* normal, float* eik_partial, void* stream);
int n2m_sdf_alpha_backward(const float* d_alpha, const float* sdf, const float* sdf6, const float* dirs, const float* ts, uint32_t M,
                           const float* variance, float eps, float cos_anneal_ratio, const float* seed, float eik_coef, float* d_sdf,
                           float* d_sdf6, float* var_partial, float* d_variance, float* found_inf, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * shencoder   (reference: shencoder/src/shencoder.h:9-10, shencoder/src/bindings.cpp:5-8)
 * ---------------------------------------------------------------------------------------------------- */

/* shencoder.h:9  sh_encode_forward   kernel shencoder.cu:27-356.
 * inputs [B,3] unit vectors; outputs [B,degree^2]; dy_dx NULL or [B,3,degree^2].  degree in 1..8, D == 3. */
int n2m_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t degree,
                          float* dy_dx, void* stream);

/* shencoder.h:10  sh_encode_backward   kernel shencoder.cu:358-382.
 * grad_inputs[b,d] += sum_ch grad[b,ch] * dy_dx[b,d,ch]   (accumulates, as the reference does). */
int n2m_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t degree,
                           const float* dy_dx, float* grad_inputs, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * per-kernel timing (measurement support for bench.py; not part of the reference surface)
 * ---------------------------------------------------------------------------------------------------- */

/* Kernel ids for n2m_prof_*: one per launch site family. */
enum {
    N2M_K_GRID_FWD = 0, N2M_K_GRID_BWD, N2M_K_GRID_TV, N2M_K_MARCH_COUNT, N2M_K_MARCH_WRITE,
    N2M_K_COMPOSITE_FWD, N2M_K_COMPOSITE_BWD, N2M_K_NEAR_FAR, N2M_K_PACKBITS, N2M_K_MLP_FWD, N2M_K_MLP_BWD,
    N2M_K_RASTER, N2M_K_GRID_FWD_PACKED /* the training lookup: both field tables from the packed copy */, N2M_K_ADAM,
    N2M_K_INTERP_FWD, N2M_K_INTERP_BWD, N2M_K_AA_FWD, N2M_K_AA_BWD, N2M_K_RASTER_BWD, N2M_K_COUNT
};
/* on = 1: every launch of a profiled kernel is bracketed by a hipEvent pair recorded on the launch stream; on = n > 1:
 * every n-th launch of each kernel id (sampled timing, keeps the event overhead out of a timed region); 0: off.
 * Events come from a fixed pool; launches beyond the pool are counted but not timed. */
int n2m_prof_enable(int on);
int n2m_prof_reset(void);
/* Synchronises the recorded events, then returns the number of timed launches of `kernel_id`, their summed
 * duration in milliseconds and the algorithmic bytes the library attributed to them. */
int n2m_prof_read(int kernel_id, uint64_t* launches, double* total_ms, double* algo_bytes);
/* Launches of `kernel_id` SEEN since n2m_prof_reset (timed or not): with sampled timing, per-step cost = mean timed duration x seen / steps. */
int n2m_prof_seen(int kernel_id, uint64_t* launches_seen);
const char* n2m_prof_name(int kernel_id);

/* Device-to-device copy of `bytes` (multiple of 16, 16-byte aligned) as a grid-stride kernel of 16 bytes per lane (nontemporal load and
 * store); workgroups = 0 picks 8 per CU.  bench.py times it as `peak_measured`: the streaming ceiling HBM-bound kernels are priced against
 * next to the nominal 8 TB/s (MI355X_MICROARCH.md: 6.29 TB/s for this form). */
int n2m_stream_copy(const void* src, void* dst, uint64_t bytes, uint32_t workgroups, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* N2M_HIP_H */
