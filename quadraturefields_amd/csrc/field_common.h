// Device helpers shared by the fp32 and bf16 field kernels: hash-grid level table, corner indexing, trilinear
// blend weights, SH basis, sigmoid.  tiny-cuda-nn semantics per SURVEY.md Appendix A.1 / A.3.
#pragma once
#include "qf_common.h"

#define QF_PRIME_Y 2654435761u
#define QF_PRIME_Z 805459861u

namespace {

struct GridArgs {
    uint32_t offset[QF_MAX_LEVELS];
    uint32_t rows[QF_MAX_LEVELS];
    uint32_t res[QF_MAX_LEVELS];
    float scale[QF_MAX_LEVELS];
    uint32_t hashed_mask;
};

// column of the (64-wide) hidden input consumed at k-step s by lane quartet kq (fp32 16x16x4 MFMA chaining)
__device__ __forceinline__ int hidden_col(int s, int kq) { return 16 * (s >> 2) + 4 * kq + (s & 3); }

struct LevelConst {
    uint32_t offset, rows, res, hashed;
    float scale;
};

// Table rows of the 8 corners of the cell containing (x,y,z) at one level, and the fractional position in it.
__device__ __forceinline__ void level_indices(const LevelConst &lc, float x, float y, float z,
                                              uint32_t idx[8], float frac[3])
{
    const float px = fmaf(lc.scale, x, 0.5f), py = fmaf(lc.scale, y, 0.5f), pz = fmaf(lc.scale, z, 0.5f);
    const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    frac[0] = px - fx;
    frac[1] = py - fy;
    frac[2] = pz - fz;
    const uint32_t gx = (uint32_t)(int32_t)fx, gy = (uint32_t)(int32_t)fy, gz = (uint32_t)(int32_t)fz;

    uint32_t cx[2] = {gx, gx + 1u}, cy[2] = {gy, gy + 1u}, cz[2] = {gz, gz + 1u};
    if (lc.hashed) {
        const uint32_t hy[2] = {cy[0] * QF_PRIME_Y, cy[1] * QF_PRIME_Y};
        const uint32_t hz[2] = {cz[0] * QF_PRIME_Z, cz[1] * QF_PRIME_Z};
#pragma unroll
        for (int c = 0; c < 8; ++c)
            idx[c] = (cx[c & 1] ^ hy[(c >> 1) & 1] ^ hz[c >> 2]) & (lc.rows - 1u);   // rows is 2^T when hashed
    } else {
        // dense: stride walks 1, res, res^2 while stride <= rows (A.1); res^2 <= rows always holds
        // for a dense level except the degenerate res^2 > rows case handled by the host check.
        const uint32_t r2 = lc.res * lc.res;
        const uint32_t sy[2] = {cy[0] * lc.res, cy[1] * lc.res};
        const uint32_t sz[2] = {cz[0] * r2, cz[1] * r2};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            uint32_t v = cx[c & 1] + sy[(c >> 1) & 1] + sz[c >> 2];
            if (v >= lc.rows) v %= lc.rows;   // rare: cell on the upper faces, or point outside the aabb
            idx[c] = v;
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) idx[c] += lc.offset;
}

// Trilinear blend of the 8 gathered corners; weight = ((1 * wx?) * wy?) * wz?, corner order 0..7
// (bit d of the corner index selects the upper cell along axis d) -- as tcnn / the oracle.
__device__ __forceinline__ void level_blend(const float2 val[8], const float frac[3], float *f0, float *f1)
{
    const float wx = frac[0], wy = frac[1], wz = frac[2];
    const float wx0 = 1.0f - wx, wy0 = 1.0f - wy, wz0 = 1.0f - wz;
    const float wxy[4] = {wx0 * wy0, wx * wy0, wx0 * wy, wx * wy};
    float a = 0.0f, b = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float w = wxy[c & 3] * ((c & 4) ? wz : wz0);
        a = fmaf(w, val[c].x, a);
        b = fmaf(w, val[c].y, b);
    }
    *f0 = a;
    *f1 = b;
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// 4 of the 16 degree-4 SH basis values: components 4g .. 4g+3 (A.3).
__device__ __forceinline__ void sh4_quartet(int g, float x, float y, float z, float out[4])
{
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    float v[16];
    v[0] = 0.28209479177387814f;
    v[1] = -0.48860251190291987f * y;
    v[2] = 0.48860251190291987f * z;
    v[3] = -0.48860251190291987f * x;
    v[4] = 1.0925484305920792f * xy;
    v[5] = -1.0925484305920792f * yz;
    v[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    v[7] = -1.0925484305920792f * xz;
    v[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    v[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
    v[10] = 2.8906114426405538f * xy * z;
    v[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
    v[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
    v[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
    v[14] = 1.4453057213202769f * z * (x2 - y2);
    v[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float a = v[r], b = v[4 + r], c = v[8 + r], d = v[12 + r];
        out[r] = g == 0 ? a : (g == 1 ? b : (g == 2 ? c : d));
    }
}

inline int fill_grid_args(const qf_grid_desc *d, GridArgs *ga)
{
    if (!d || d->n_levels != QF_MAX_LEVELS || d->n_features != 2) return QF_ERR_UNSUPPORTED;
    for (int l = 0; l < QF_MAX_LEVELS; ++l) {
        ga->offset[l] = d->offset[l];
        ga->rows[l] = d->offset[l + 1] - d->offset[l];
        ga->res[l] = d->resolution[l];
        ga->scale[l] = d->scale[l];
        const bool hashed = (d->hashed_mask >> l) & 1u;
        if (hashed && (ga->rows[l] & (ga->rows[l] - 1u))) return QF_ERR_UNSUPPORTED;
        // dense levels must take all three strides (res^2 <= rows), true for rows >= res^3
        if (!hashed && (uint64_t)d->resolution[l] * d->resolution[l] > ga->rows[l]) return QF_ERR_UNSUPPORTED;
    }
    ga->hashed_mask = d->hashed_mask;
    return QF_OK;
}

}  // namespace
