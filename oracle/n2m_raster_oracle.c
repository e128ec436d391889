/*
 * n2m_raster_oracle.c -- TEST INFRASTRUCTURE ONLY.  Scalar CPU statement of the stage-1 primitives declared in
 * include/n2m_raster.h (rasterize / interpolate / antialias as nerf2mesh uses them, nerf/renderer.py:860-887).
 *
 * PARITY UNPINNED: the reference delegates these operators to nvdiffrast, which is neither vendored under
 * /root/reference nor installed here (unpinned git HEAD, readme.md:28-29).  This file states the published semantics
 *  as plain serial loops -- a per-pixel loop over ALL triangles with an explicit nearest-depth
 * test instead of the product's atomic z-buffer, a sorted edge list instead of its hash table -- so that the HIP kernels
 * can be checked against an independent evaluation order, and the analytic tests in tests/test_raster_oracle.py check
 * this file against closed forms (single-triangle coverage, watertight shared edges, finite differences).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x[3], y[3], z[3], w[3]; int ok; } tri_t;

static tri_t fetch(const float* pos, const int32_t* tri, uint32_t f, uint32_t V) {
    tri_t t; t.ok = 1;
    for (int k = 0; k < 3; ++k) {
        const int32_t i = tri[3 * f + k];
        if (i < 0 || (uint32_t)i >= V) { t.ok = 0; return t; }
        t.x[k] = pos[4 * i]; t.y[k] = pos[4 * i + 1]; t.z[k] = pos[4 * i + 2]; t.w[k] = pos[4 * i + 3];
    }
    return t;
}

/* homogeneous edge functions at NDC point (fx,fy): perspective-correct barycentrics, z/w, interpolated w */
static int shade(const tri_t* t, float fx, float fy, float* b0, float* b1, float* zw, float* wp) {
    float px[3], py[3];
    for (int k = 0; k < 3; ++k) { px[k] = t->x[k] - fx * t->w[k]; py[k] = t->y[k] - fy * t->w[k]; }
    const float a0 = px[1] * py[2] - py[1] * px[2], a1 = px[2] * py[0] - py[2] * px[0], a2 = px[0] * py[1] - py[0] * px[1];
    const float S = a0 + a1 + a2;
    if (S == 0.0f) return 0;
    const float iw = 1.0f / S;
    *b0 = a0 * iw; *b1 = a1 * iw;
    const float b2 = a2 * iw;
    const float z = t->z[0] * *b0 + t->z[1] * *b1 + t->z[2] * b2;
    *wp = t->w[0] * *b0 + t->w[1] * *b1 + t->w[2] * b2;
    *zw = z / *wp;
    return 1;
}

static int covers(const tri_t* t, int ix, int iy, uint32_t H, uint32_t W, float* zw_out, float* b0_out, float* b1_out) {
    const float fx = ((float)ix + 0.5f) * (2.0f / (float)W) - 1.0f, fy = ((float)iy + 0.5f) * (2.0f / (float)H) - 1.0f;
    float b0, b1, zw, wp;
    int fixed = t->w[0] > 1e-12f && t->w[1] > 1e-12f && t->w[2] > 1e-12f;
    long long X[3], Y[3];
    if (fixed) {
        for (int k = 0; k < 3; ++k) {
            const float px = (t->x[k] / t->w[k] * 0.5f + 0.5f) * (float)W, py = (t->y[k] / t->w[k] * 0.5f + 0.5f) * (float)H;
            if (!(fabsf(px) < 1048576.f && fabsf(py) < 1048576.f)) fixed = 0;
            X[k] = (long long)lrintf(px * 256.0f); Y[k] = (long long)lrintf(py * 256.0f);
        }
    }
    if (fixed) {
        const long long area2 = (X[1] - X[0]) * (Y[2] - Y[0]) - (Y[1] - Y[0]) * (X[2] - X[0]);
        if (area2 == 0) return 0;
        const long long sg = area2 > 0 ? 1 : -1;
        const long long px = (long long)ix * 256 + 128, py = (long long)iy * 256 + 128;
        for (int k = 0; k < 3; ++k) {
            const int a = k, b = (k + 1) % 3;
            const long long dx = sg * (X[b] - X[a]), dy = sg * (Y[b] - Y[a]);
            const long long e = dx * (py - Y[a]) - dy * (px - X[a]);
            if (e < 0) return 0;
            if (e == 0 && !(dy > 0 || (dy == 0 && dx > 0))) return 0;
        }
        if (!shade(t, fx, fy, &b0, &b1, &zw, &wp)) return 0;
    } else {
        if (!shade(t, fx, fy, &b0, &b1, &zw, &wp)) return 0;
        const float b2 = 1.0f - b0 - b1;
        if (!(b0 >= 0.f && b1 >= 0.f && b2 >= 0.f && wp > 0.f)) return 0;
    }
    if (!(zw >= -1.0f && zw <= 1.0f)) return 0;
    *zw_out = zw; *b0_out = b0; *b1_out = b1;
    return 1;
}

/* per pixel: nearest covering triangle, ties -> lower id */
void n2m_oracle_rasterize(const float* pos, const int32_t* tri, uint32_t V, uint32_t F, uint32_t H, uint32_t W, float* rast) {
    memset(rast, 0, sizeof(float) * 4 * (size_t)H * W);
    float* best = (float*)malloc(sizeof(float) * (size_t)H * W);
    for (size_t i = 0; i < (size_t)H * W; ++i) best[i] = INFINITY;
    for (uint32_t f = 0; f < F; ++f) {
        const tri_t t = fetch(pos, tri, f, V);
        if (!t.ok) continue;
#pragma omp parallel for schedule(static)
        for (int iy = 0; iy < (int)H; ++iy)
            for (int ix = 0; ix < (int)W; ++ix) {
                float zw, b0, b1;
                if (!covers(&t, ix, iy, H, W, &zw, &b0, &b1)) continue;
                const size_t i = (size_t)iy * W + ix;
                if (zw < best[i]) {   /* strict: an equal depth keeps the lower id drawn earlier */
                    best[i] = zw;
                    rast[4 * i] = b0; rast[4 * i + 1] = b1; rast[4 * i + 2] = fminf(fmaxf(zw, -1.f), 1.f); rast[4 * i + 3] = (float)(f + 1);
                }
            }
    }
    free(best);
}

/* The same image with each triangle confined to the pixel box of its vertices (+1 px: the 1/256-pixel snap of `covers`): a pixel outside
 * that box cannot pass the edge tests, so the result equals n2m_oracle_rasterize bit for bit (tests/test_raster_oracle.py) at a cost of
 * O(covered pixels) instead of O(F H W).  Triangles with a vertex at w <= 0 (float homogeneous path) keep the full image.  This is the
 * form bench.py times as the stage-1 CPU baseline (a scalar software rasteriser on the same inputs). */
void n2m_oracle_rasterize_bbox(const float* pos, const int32_t* tri, uint32_t V, uint32_t F, uint32_t H, uint32_t W, float* rast) {
    memset(rast, 0, sizeof(float) * 4 * (size_t)H * W);
    float* best = (float*)malloc(sizeof(float) * (size_t)H * W);
    for (size_t i = 0; i < (size_t)H * W; ++i) best[i] = INFINITY;
    for (uint32_t f = 0; f < F; ++f) {
        const tri_t t = fetch(pos, tri, f, V);
        if (!t.ok) continue;
        int x0 = 0, x1 = (int)W - 1, y0 = 0, y1 = (int)H - 1;
        if (t.w[0] > 1e-12f && t.w[1] > 1e-12f && t.w[2] > 1e-12f) {
            float lox = INFINITY, hix = -INFINITY, loy = INFINITY, hiy = -INFINITY;
            int finite = 1;
            for (int k = 0; k < 3; ++k) {
                const float px = (t.x[k] / t.w[k] * 0.5f + 0.5f) * (float)W, py = (t.y[k] / t.w[k] * 0.5f + 0.5f) * (float)H;
                if (!(fabsf(px) < 1048576.f && fabsf(py) < 1048576.f)) finite = 0;
                lox = fminf(lox, px); hix = fmaxf(hix, px); loy = fminf(loy, py); hiy = fmaxf(hiy, py);
            }
            if (finite) {
                x0 = (int)fmaxf(0.f, floorf(lox) - 1.f); x1 = (int)fminf((float)W - 1.f, ceilf(hix) + 1.f);
                y0 = (int)fmaxf(0.f, floorf(loy) - 1.f); y1 = (int)fminf((float)H - 1.f, ceilf(hiy) + 1.f);
            }
        }
        for (int iy = y0; iy <= y1; ++iy)
            for (int ix = x0; ix <= x1; ++ix) {
                float zw, b0, b1;
                if (!covers(&t, ix, iy, H, W, &zw, &b0, &b1)) continue;
                const size_t i = (size_t)iy * W + ix;
                if (zw < best[i]) {
                    best[i] = zw;
                    rast[4 * i] = b0; rast[4 * i + 1] = b1; rast[4 * i + 2] = fminf(fmaxf(zw, -1.f), 1.f); rast[4 * i + 3] = (float)(f + 1);
                }
            }
    }
    free(best);
}

void n2m_oracle_interpolate(const float* attr, const float* rast, const int32_t* tri, uint32_t V, uint32_t F, uint32_t A,
                            uint32_t H, uint32_t W, float* out) {
    (void)V;
    for (size_t i = 0; i < (size_t)H * W; ++i) {
        const int f = (int)rast[4 * i + 3] - 1;
        for (uint32_t a = 0; a < A; ++a) out[i * A + a] = 0.f;
        if (f < 0 || (uint32_t)f >= F) continue;
        const float b0 = rast[4 * i], b1 = rast[4 * i + 1], b2 = 1.0f - b0 - b1;
        const int i0 = tri[3 * f], i1 = tri[3 * f + 1], i2 = tri[3 * f + 2];
        for (uint32_t a = 0; a < A; ++a)
            out[i * A + a] = b0 * attr[(size_t)i0 * A + a] + b1 * attr[(size_t)i1 * A + a] + b2 * attr[(size_t)i2 * A + a];
    }
}

/* ---- antialias: sorted (lo, hi, opposite) list instead of a hash table */
typedef struct { int32_t lo, hi, op; } edge_t;
static int edge_cmp(const void* a, const void* b) {
    const edge_t* x = (const edge_t*)a; const edge_t* y = (const edge_t*)b;
    if (x->lo != y->lo) return x->lo < y->lo ? -1 : 1;
    if (x->hi != y->hi) return x->hi < y->hi ? -1 : 1;
    return 0;
}
static int other_vertex(const edge_t* e, size_t n, int32_t a, int32_t b, int32_t c) {
    const int32_t lo = a < b ? a : b, hi = a < b ? b : a;
    size_t l = 0, r = n;
    while (l < r) { const size_t m = (l + r) / 2; if (e[m].lo < lo || (e[m].lo == lo && e[m].hi < hi)) l = m + 1; else r = m; }
    /* entries of this edge are consecutive; return an opposite vertex different from c (first two only, like the product) */
    int first = -1, second = -1, cnt = 0;
    for (size_t i = l; i < n && e[i].lo == lo && e[i].hi == hi; ++i) { if (cnt == 0) first = e[i].op; else if (cnt == 1) second = e[i].op; ++cnt; }
    if (cnt == 0) return -1;
    if (cnt == 1) return first != c ? first : -1;
    /* two (or more) triangles share the edge: the one that is not c; insertion order in the product's table is not
       defined, so a non-manifold edge (cnt > 2) is ambiguous there as well */
    return first != c ? first : second;
}

static float cross2(float ux, float uy, float vx, float vy) { return ux * vy - uy * vx; }

void n2m_oracle_antialias(const float* color, const float* rast, const float* pos, const int32_t* tri, uint32_t V, uint32_t F,
                          uint32_t C, uint32_t H, uint32_t W, float* out) {
    edge_t* edges = (edge_t*)malloc(sizeof(edge_t) * 3 * (size_t)F + 1);
    size_t ne = 0;
    for (uint32_t f = 0; f < F; ++f)
        for (int k = 0; k < 3; ++k) {
            const int32_t a = tri[3 * f + k], b = tri[3 * f + (k + 1) % 3], c = tri[3 * f + (k + 2) % 3];
            if (a < 0 || b < 0 || c < 0 || a == b) continue;
            edges[ne].lo = a < b ? a : b; edges[ne].hi = a < b ? b : a; edges[ne].op = c; ++ne;
        }
    qsort(edges, ne, sizeof(edge_t), edge_cmp);
    memcpy(out, color, sizeof(float) * (size_t)H * W * C);
    for (uint32_t p = 0; p < H * W; ++p)
        for (int dir = 0; dir < 2; ++dir) {
            const uint32_t ix = p % W, iy = p / W;
            if (dir == 0 ? ix + 1 >= W : iy + 1 >= H) continue;
            const uint32_t q = dir == 0 ? p + 1 : p + W;
            const int ta = (int)rast[4 * p + 3] - 1, tb = (int)rast[4 * q + 3] - 1;
            if (ta == tb) continue;
            const int use_a = tb < 0 || (ta >= 0 && rast[4 * p + 2] <= rast[4 * q + 2]);
            const int f = use_a ? ta : tb;
            const uint32_t P = use_a ? p : q, O = use_a ? q : p;
            const float Px = (float)(P % W) + 0.5f, Py = (float)(P / W) + 0.5f, Ox = (float)(O % W) + 0.5f, Oy = (float)(O / W) + 0.5f;
            int id[3]; float X[3], Y[3]; int ok = 1;
            for (int k = 0; k < 3; ++k) {
                id[k] = tri[3 * f + k];
                if (id[k] < 0 || (uint32_t)id[k] >= V) { ok = 0; break; }
                const float* v = pos + 4 * (size_t)id[k];
                if (!(v[3] > 1e-12f)) { ok = 0; break; }
                X[k] = (v[0] / v[3] * 0.5f + 0.5f) * (float)W; Y[k] = (v[1] / v[3] * 0.5f + 0.5f) * (float)H;
            }
            if (!ok) continue;
            float best = 2.0f;
            for (int k = 0; k < 3; ++k) {
                const int a = k, b = (k + 1) % 3, c = (k + 2) % 3;
                const float ex = X[b] - X[a], ey = Y[b] - Y[a];
                const float sc = cross2(ex, ey, X[c] - X[a], Y[c] - Y[a]);
                if (sc == 0.f) continue;
                const int other = other_vertex(edges, ne, id[a], id[b], id[c]);
                int sil = 1;
                if (other >= 0 && (uint32_t)other < V) {
                    const float* v = pos + 4 * (size_t)other;
                    if (v[3] > 1e-12f) {
                        const float qx = (v[0] / v[3] * 0.5f + 0.5f) * (float)W, qy = (v[1] / v[3] * 0.5f + 0.5f) * (float)H;
                        sil = cross2(ex, ey, qx - X[a], qy - Y[a]) * sc > 0.f;
                    }
                }
                if (!sil) continue;
                const float eP = cross2(ex, ey, Px - X[a], Py - Y[a]), eO = cross2(ex, ey, Ox - X[a], Oy - Y[a]);
                if (!(eP * sc >= 0.f && eO * sc < 0.f)) continue;
                const float d = eP / (eP - eO);
                const float qx = Px + d * (Ox - Px), qy = Py + d * (Oy - Py);
                const float tt = ((qx - X[a]) * ex + (qy - Y[a]) * ey) / (ex * ex + ey * ey);
                if (!(tt >= 0.f && tt <= 1.f)) continue;
                if (d < best) best = d;
            }
            if (best > 1.5f) continue;
            const uint32_t dst = best < 0.5f ? P : O, src = best < 0.5f ? O : P;
            const float wgt = fabsf(0.5f - best);
            for (uint32_t c = 0; c < C; ++c) out[(size_t)dst * C + c] += wgt * (color[(size_t)src * C + c] - color[(size_t)dst * C + c]);
        }
    free(edges);
}
