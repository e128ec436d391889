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
#include "n2m_common.hpp"

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
