// Texture-bake padding on the device: the nearest-neighbour fill the reference runs on the host with a kd-tree over the texel
// coordinates (`NearestNeighbors(n_neighbors=1).fit(search_coords)`, nerf/renderer.py:371-387) after copying the baked atlas down.
// Every texel of the band around the charts (role bit 1) takes the features of the nearest chart-boundary texel (role bit 0),
// Euclidean distance on (row, column).  The band is the 32-step dilation of the covered texels, so the nearest source is never farther
// than `radius` = 32: each destination texel walks the square rings around itself outwards and stops as soon as the next ring cannot
// beat what it has (a ring of Chebyshev radius r holds no texel closer than r).  Ties (the kd-tree's choice among equidistant texels
// is arbitrary): smallest row, then smallest column.  Sources and destinations are disjoint, so the fill is done in place.
#include "n2m_common.hpp"

namespace {

__global__ void __launch_bounds__(256)
texture_pad_kernel(uint8_t* __restrict__ feats, const uint8_t* __restrict__ role, uint32_t H, uint32_t W, uint32_t C, int radius) {
    const uint32_t px = blockIdx.x * 64u + (threadIdx.x & 63u), py = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (px >= W || py >= H || !(role[(size_t)py * W + px] & 2u)) return;
    int best = 0x7fffffff, by = 0, bx = 0;
    auto probe = [&](int y, int x) {
        if (y < 0 || x < 0 || y >= (int)H || x >= (int)W || !(role[(size_t)y * W + x] & 1u)) return;
        const int dy = y - (int)py, dx = x - (int)px, d2 = dy * dy + dx * dx;
        if (d2 < best || (d2 == best && (y < by || (y == by && x < bx)))) { best = d2; by = y; bx = x; }
    };
    for (int r = 1; r <= radius && r * r <= best; ++r) {
        for (int x = (int)px - r; x <= (int)px + r; ++x) { probe((int)py - r, x); probe((int)py + r, x); }
        for (int y = (int)py - r + 1; y <= (int)py + r - 1; ++y) { probe(y, (int)px - r); probe(y, (int)px + r); }
    }
    if (best == 0x7fffffff) return;
    const uint8_t* __restrict__ src = feats + ((size_t)by * W + bx) * C;
    uint8_t* __restrict__ dst = feats + ((size_t)py * W + px) * C;
    for (uint32_t c = 0; c < C; ++c) dst[c] = src[c];
}

}  // namespace

extern "C" int n2m_texture_pad_nearest(uint8_t* feats, const uint8_t* role, uint32_t H, uint32_t W, uint32_t C, uint32_t radius, void* stream) {
    N2M_REQUIRE(feats != nullptr && role != nullptr, N2M_ENULL, "texture_pad_nearest: NULL feats / role");
    N2M_REQUIRE(H > 0 && W > 0 && C > 0 && C <= 16 && radius <= 1024, N2M_EINVAL, "texture_pad_nearest: H, W > 0, 1 <= C <= 16, radius <= 1024");
    texture_pad_kernel<<<dim3(n2m_ceil_div(W, 64), n2m_ceil_div(H, 4)), 256, 0, (hipStream_t)stream>>>(feats, role, H, W, C, (int)radius);
    N2M_CHECK_LAUNCH();
    return 0;
}
