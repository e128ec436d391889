// Frequency (positional) encoding for gfx950 -- replaces the reference's _freqencoder extension
// (freqencoder/src/freqencoder.cu:30-94; entry points in include/n2m_hip.h).  Dormant in nerf2mesh's networks (the
// `frequency` option of encoding.py:84-86) but part of the operator surface.
//
// outputs [B, C], C = D + 2*deg*D: block 0 = input, block 1+2f = sin(2^f x), block 2+2f = sin(2^f x + pi/2) -- the
// reference spells its cosine that way (:58-60) and so does this kernel, with sinf instead of the fast-math __sinf.
// One thread per output element forward (coalesced stores), one per input element backward.
#include "n2m_common.hpp"

namespace {

__global__ void __launch_bounds__(256)
freq_forward_kernel(const float* __restrict__ inputs, uint32_t B, uint32_t D, uint32_t C, float* __restrict__ outputs) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (uint64_t)B * C) return;
    const uint32_t b = (uint32_t)(t / C), c = (uint32_t)(t - (uint64_t)b * C);
    const float* __restrict__ in = inputs + (size_t)b * D;
    if (c < D) { outputs[t] = in[c]; return; }
    const uint32_t col = c / D - 1u, d = c % D, freq = col >> 1;
    const float phase_shift = (float)(col & 1u) * (3.141592653589793f / 2);
    outputs[t] = sinf(scalbnf(in[d], (int)freq) + phase_shift);
}

__global__ void __launch_bounds__(256)
freq_backward_kernel(const float* __restrict__ grad, const float* __restrict__ outputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                     float* __restrict__ grad_inputs) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (uint64_t)B * D) return;
    const uint32_t b = (uint32_t)(t / D), d = (uint32_t)(t - (uint64_t)b * D);
    const float* __restrict__ g = grad + (size_t)b * C;
    const float* __restrict__ o = outputs + (size_t)b * C;
    float result = g[d];
    g += D; o += D;
    for (uint32_t f = 0; f < deg; ++f) {     // d sin = 2^f cos, d cos = -2^f sin, from the stored outputs (:85-89)
        result += scalbnf(1.0f, (int)f) * (g[d] * o[D + d] - g[D + d] * o[d]);
        g += 2 * D; o += 2 * D;
    }
    grad_inputs[t] = result;
}

}  // namespace

extern "C" int n2m_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float* outputs, void* stream) {
    N2M_REQUIRE(inputs && outputs, N2M_ENULL, "freq_encode_forward: NULL tensor");
    N2M_REQUIRE(D >= 1 && C == D + 2 * deg * D, N2M_EINVAL, "freq_encode_forward: C must be D + 2*deg*D (D=%u deg=%u C=%u)", D, deg, C);
    if (B == 0) return 0;
    freq_forward_kernel<<<n2m_ceil_div((uint64_t)B * C, 256), 256, 0, (hipStream_t)stream>>>(inputs, B, D, C, outputs);
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_freq_encode_backward(const float* grad, const float* outputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                                        float* grad_inputs, void* stream) {
    N2M_REQUIRE(grad && outputs && grad_inputs, N2M_ENULL, "freq_encode_backward: NULL tensor");
    N2M_REQUIRE(D >= 1 && C == D + 2 * deg * D, N2M_EINVAL, "freq_encode_backward: C must be D + 2*deg*D (D=%u deg=%u C=%u)", D, deg, C);
    if (B == 0) return 0;
    freq_backward_kernel<<<n2m_ceil_div((uint64_t)B * D, 256), 256, 0, (hipStream_t)stream>>>(grad, outputs, B, D, deg, C, grad_inputs);
    N2M_CHECK_LAUNCH();
    return 0;
}
