/*
 * n2m_mlp.h -- fused field MLPs of nerf2mesh (the "M" row of SURVEY.md section 8a).
 *
 * New operator without a reference boundary: the reference evaluates its three tiny bias-free MLPs with
 * nn.Linear + F.relu under fp16 autocast (nerf/network.py:10-54,92-108,159-189), i.e. 7 GEMM launches forward and
 * ~14 backward with K <= 64, several of them degenerate tall-skinny shapes.  These entry points evaluate the whole
 * field head in one kernel per direction with the weights resident in LDS and the layer chain in MFMA registers.
 * The unfused torch path stays the parity baseline (tests/test_mlp_parity.py).
 *
 * Fixed architecture (nerf/network.py:66-75):
 *   sigma_net    : [x(3) | h1(16)] -> 32 -> 1            sigma = exp(.)            (trunc_exp, activation.py:5-17)
 *   color_net    : [x(3) | h2(32)] -> 64 -> 64 -> 6      geo = sigmoid(.) ; diffuse = geo[0:3], feat = geo[3:6]
 *   specular_net : [d(3) | feat(3)] -> 32 -> 3           specular = sigmoid(.)
 *   color = diffuse                       (shading 0, "diffuse")
 *         = clamp(specular + diffuse,0,1) (shading 1, "full")
 *         = specular                      (shading 2, "specular")
 * Numerics follow autocast: layer inputs, weights and layer outputs are rounded to fp16, accumulation is fp32
 * (v_mfma_f32_32x32x8_f16), exp runs in fp32.
 *
 * Weight pointers are the fp32 master parameters, row-major [out, in] exactly as nn.Linear stores them:
 *   w_sigma0 [32,19]  w_sigma1 [1,32]  w_color0 [64,35]  w_color1 [64,64]  w_color2 [6,64]  w_spec0 [32,6]  w_spec1 [3,32]
 * All pointers are device pointers; conventions as in n2m_hip.h.
 */
#ifndef N2M_MLP_H
#define N2M_MLP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Forward.  xyz [M,3] f32 (grid-bound coordinates, as fed to the encoders), dirs [M,3] f32 (may be NULL when
 * shading == 0; unit vectors, or with bit 0 of normalize_dirs set the raw ray directions march_rays_train hands out, which are
 * then normalised on load exactly like safe_normalize, nerf/renderer.py:704; bit 1 of normalize_dirs selects the SDF head: sigma is the
 * fp16 output of sigma_net as it is, `h[..., 0].float()` of nerf/network.py:100-101, instead of trunc_exp of it -- forward and backward), h1 [16,M] f32 (density features, LEVEL-major as n2m_grid_encode_forward writes them), h2 [16,M,2] f16
 * (colour features, level-major; may be NULL
 * when rgb == NULL: density-only evaluation as in update_extra_state).
 * Outputs: sigma [M] f32; rgb [M,3] f32 and specular [M,3] f32 (either may be NULL; specular is not written for
 * shading 0).  sigma == NULL selects the colour-only evaluation of stage 1 (NeRFNetwork.rgb, nerf/renderer.py:875-881):
 * h1 and the sigma_net weights are then not read. */
int n2m_field_forward(const float* xyz, const float* dirs, const float* h1, const void* h2, const float* w_sigma0,
                      const float* w_sigma1, const float* w_color0, const float* w_color1, const float* w_color2,
                      const float* w_spec0, const float* w_spec1, uint32_t M, int shading, int normalize_dirs, float* sigma, float* rgb,
                      float* specular, void* stream);

/* Backward of n2m_field_forward (activations are recomputed, nothing is saved but the inputs).
 * Incoming: d_sigma [M], d_rgb [M,3], d_specular [M,3] or NULL (gradient of the loss w.r.t. the three outputs).
 * Outgoing:
 *   d_h1 [16, M] f32      LEVEL-major, the layout n2m_grid_encode_backward consumes for the C=1 encoder
 *   d_h2 [16, M, 2] f16   LEVEL-major for the C=2 encoder (fp16 like autocast's grad of an fp16 tensor)
 *   d_w_* : fp32, same shapes as the weights, ACCUMULATED into (caller zero-fills or keeps running sums)
 * d_sigma == NULL (colour only) or d_rgb == NULL (density only) skip the corresponding branch and its outputs.
 * Pass already-scaled upstream gradients (GradScaler) as they are.  found_inf (device float, may be NULL) is set to 1 when a
 * weight-gradient value flushed to HBM is not finite; it is never cleared here. */
int n2m_field_backward(const float* xyz, const float* dirs, const float* h1, const void* h2, const float* w_sigma0,
                       const float* w_sigma1, const float* w_color0, const float* w_color1, const float* w_color2,
                       const float* w_spec0, const float* w_spec1, uint32_t M, int shading, int normalize_dirs, const float* d_sigma,
                       const float* d_rgb, const float* d_specular, float* d_h1, void* d_h2, float* d_w_sigma0,
                       float* d_w_sigma1, float* d_w_color0, float* d_w_color1, float* d_w_color2, float* d_w_spec0,
                       float* d_w_spec1, float* found_inf, void* stream);

/* Training forms with the specular regulariser of the reference's train_step folded in (nerf/utils.py:733-737:
 * loss += lambda_specular * (specular ** 2).sum(-1).mean(), present whenever shading != diffuse, i.e. for 29 000 of the 30 000 iterations
 * of main.py:59).  The reference forms the term with four elementwise/reduction launches over the [M,3] specular tensor and autograd sends
 * 2 lambda / M * specular back; here the forward leaves per-workgroup sums of specular^2 (fp32 squares of the fp16 sigmoid outputs, summed
 * in a fixed order) in spec_sq_partial[0 .. n2m_field_spec_partials()) -- unused slots are written as zeros -- and the backward adds
 * specular * (*seed * spec_reg) to d loss / d specular of the recomputed activations (seed: device scalar, the seed gradient of the step =
 * loss scale [/ world]; spec_reg = 2 lambda_specular / M on the host).  `specular` may then be NULL in the forward and `d_specular` NULL in
 * the backward: the [M,3] tensor is neither written nor read.  spec_sq_partial == NULL / seed == NULL or spec_reg == 0: identical to the
 * plain entry points.  Ignored with shading == 0. */
int n2m_field_forward_train(const float* xyz, const float* dirs, const float* h1, const void* h2, const float* w_sigma0,
                            const float* w_sigma1, const float* w_color0, const float* w_color1, const float* w_color2,
                            const float* w_spec0, const float* w_spec1, uint32_t M, int shading, int normalize_dirs, float* sigma, float* rgb,
                            float* specular, float* spec_sq_partial, void* stream);
int n2m_field_backward_train(const float* xyz, const float* dirs, const float* h1, const void* h2, const float* w_sigma0,
                             const float* w_sigma1, const float* w_color0, const float* w_color1, const float* w_color2,
                             const float* w_spec0, const float* w_spec1, uint32_t M, int shading, int normalize_dirs, const float* d_sigma,
                             const float* d_rgb, const float* d_specular, float* d_h1, void* d_h2, float* d_w_sigma0,
                             float* d_w_sigma1, float* d_w_color0, float* d_w_color1, float* d_w_color2, float* d_w_spec0,
                             float* d_w_spec1, float* found_inf, float spec_reg, const float* seed, void* stream);
uint32_t n2m_field_spec_partials(void);      /* slots the forward writes into spec_sq_partial (512) */

#ifdef __cplusplus
}
#endif
#endif /* N2M_MLP_H */
