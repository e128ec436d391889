// Real spherical-harmonics direction encoding (degree 1..8) for gfx950.  Replaces the reference's _shencoder
// extension (shencoder/src/shencoder.cu); entry points are declared in include/n2m_hip.h.
//
// The reference spells the 64 basis functions and their 192 partial derivatives out as literal polynomials.
// Here they are generated at compile time from the factorisation every entry has,
//     Y(l, m) = N(l,|m|) * Pbar(l,|m|)(z) * { Re, Im }((x + i y)^|m|),        index l*l + l + m,
// with Pbar = P_l^m(z) / (1-z^2)^(m/2) from the usual three-term recurrences and the Condon-Shortley sign
// (-1)^m folded into N (so out[1] = -c*y, out[2] = c*z, out[3] = -c*x like shencoder.cu:52-54).  The loops
// are fully unrolled for a compile-time degree, so the kernel is straight-line FMAs like the reference's,
// and agrees with it to fp32 rounding (tests state the tolerance).
#include "n2m_common.hpp"

namespace {

// sqrt((2l+1)/(4 pi) * (l-m)!/(l+m)!) * (m ? sqrt(2) * (-1)^m : 1), evaluated in double at compile time
constexpr double cx_sqrt(double v) {
    double r = v > 1 ? v : 1.0;
    for (int i = 0; i < 64; ++i) r = 0.5 * (r + v / r);
    return r;
}
constexpr double sh_norm(int l, int m) {
    double ratio = 1.0;
    for (int k = l - m + 1; k <= l + m; ++k) ratio /= (double)k;
    double v = cx_sqrt((2.0 * l + 1.0) / (4.0 * 3.14159265358979323846) * ratio);
    if (m > 0) v *= cx_sqrt(2.0) * ((m & 1) ? -1.0 : 1.0);
    return v;
}

template <int DEG, bool WITH_GRAD>
__global__ void __launch_bounds__(256)
sh_forward_kernel(const float* __restrict__ inputs, float* __restrict__ outputs, uint32_t B, uint32_t D,
                  float* __restrict__ dy_dx) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    constexpr int C2 = DEG * DEG;
    const float x = inputs[(size_t)b * D], y = inputs[(size_t)b * D + 1], z = inputs[(size_t)b * D + 2];

    float re[DEG + 1], im[DEG + 1];          // (x + i y)^m
    re[0] = 1.0f; im[0] = 0.0f;
#pragma unroll
    for (int m = 1; m <= DEG; ++m) {
        re[m] = re[m - 1] * x - im[m - 1] * y;
        im[m] = re[m - 1] * y + im[m - 1] * x;
    }
    float P[DEG][DEG + 1];                   // Pbar(l, m), m <= l; column l+1 is the zero used by d/dz
#pragma unroll
    for (int m = 0; m < DEG; ++m) {
        float pmm = 1.0f;
#pragma unroll
        for (int k = 1; k <= m; ++k) pmm *= (float)(2 * k - 1);
        P[m][m] = pmm;
        if (m + 1 < DEG) P[m + 1][m] = (float)(2 * m + 1) * z * pmm;
#pragma unroll
        for (int l = m + 2; l < DEG; ++l)
            P[l][m] = ((float)(2 * l - 1) * z * P[l - 1][m] - (float)(l + m - 1) * P[l - 2][m]) * (1.0f / (float)(l - m));
    }
    float* out = outputs + (size_t)b * C2;
    float* gx = WITH_GRAD ? dy_dx + (size_t)b * D * C2 : nullptr;
    float* gy = WITH_GRAD ? gx + C2 : nullptr;
    float* gz = WITH_GRAD ? gy + C2 : nullptr;
#pragma unroll
    for (int l = 0; l < DEG; ++l) {
#pragma unroll
        for (int m = 0; m <= l; ++m) {
            const float n = (float)sh_norm(l, m);
            const float p = P[l][m];
            const float dp = (m + 1 <= l) ? P[l][m + 1] : 0.0f;
            const int ip = l * l + l + m, in = l * l + l - m;
            const float np = n * p;
            out[ip] = np * re[m];
            if (m) out[in] = np * im[m];
            if (WITH_GRAD) {
                const float fm = (float)m;
                gx[ip] = m ? np * (fm * re[m - 1]) : 0.0f;
                gy[ip] = m ? np * (-fm * im[m - 1]) : 0.0f;
                gz[ip] = n * dp * re[m];
                if (m) {
                    gx[in] = np * (fm * im[m - 1]);
                    gy[in] = np * (fm * re[m - 1]);
                    gz[in] = n * dp * im[m];
                }
            }
        }
    }
}

// grad_inputs[b,d] += sum_ch grad[b,ch] * dy_dx[b,d,ch]
__global__ void sh_backward_kernel(const float* __restrict__ grad, uint32_t B, uint32_t D, uint32_t C2,
                                   const float* __restrict__ dy_dx, float* __restrict__ grad_inputs) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D;
    const float* g = grad + (size_t)b * C2;
    const float* j = dy_dx + (size_t)t * C2;
    float acc = grad_inputs[t];
    for (uint32_t ch = 0; ch < C2; ++ch) acc += g[ch] * j[ch];
    grad_inputs[t] = acc;
}

template <int DEG>
void launch_sh(const float* inputs, float* outputs, uint32_t B, uint32_t D, float* dy_dx, hipStream_t s) {
    const uint32_t nb = n2m_ceil_div(B, 256);
    if (dy_dx) sh_forward_kernel<DEG, true><<<nb, 256, 0, s>>>(inputs, outputs, B, D, dy_dx);
    else sh_forward_kernel<DEG, false><<<nb, 256, 0, s>>>(inputs, outputs, B, D, nullptr);
}

}  // namespace

extern "C" int n2m_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t degree,
                                     float* dy_dx, void* stream) {
    N2M_NOTNULL(inputs); N2M_NOTNULL(outputs);
    N2M_REQUIRE(D == 3, N2M_EINVAL, "sh_encode_forward: SH encoder only supports input dim == 3 (got %u)", D);
    N2M_REQUIRE(degree >= 1 && degree <= 8, N2M_EINVAL, "sh_encode_forward: SH encoder only supports degree in [1, 8] (got %u)", degree);
    if (B == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    switch (degree) {
        case 1: launch_sh<1>(inputs, outputs, B, D, dy_dx, s); break;
        case 2: launch_sh<2>(inputs, outputs, B, D, dy_dx, s); break;
        case 3: launch_sh<3>(inputs, outputs, B, D, dy_dx, s); break;
        case 4: launch_sh<4>(inputs, outputs, B, D, dy_dx, s); break;
        case 5: launch_sh<5>(inputs, outputs, B, D, dy_dx, s); break;
        case 6: launch_sh<6>(inputs, outputs, B, D, dy_dx, s); break;
        case 7: launch_sh<7>(inputs, outputs, B, D, dy_dx, s); break;
        default: launch_sh<8>(inputs, outputs, B, D, dy_dx, s); break;
    }
    N2M_CHECK_LAUNCH();
    return 0;
}

extern "C" int n2m_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t degree,
                                      const float* dy_dx, float* grad_inputs, void* stream) {
    (void)inputs;
    N2M_NOTNULL(grad); N2M_NOTNULL(dy_dx); N2M_NOTNULL(grad_inputs);
    N2M_REQUIRE(D == 3, N2M_EINVAL, "sh_encode_backward: SH encoder only supports input dim == 3 (got %u)", D);
    N2M_REQUIRE(degree >= 1 && degree <= 8, N2M_EINVAL, "sh_encode_backward: degree must be in [1, 8] (got %u)", degree);
    if (B == 0) return 0;
    sh_backward_kernel<<<n2m_ceil_div((uint64_t)B * D, 256), 256, 0, (hipStream_t)stream>>>(grad, B, D, degree * degree, dy_dx,
                                                                                           grad_inputs);
    N2M_CHECK_LAUNCH();
    return 0;
}
