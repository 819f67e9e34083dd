// Backward of the multi-resolution hash-grid encoding (training side).
//
// Replaces the backward of tcnn GridEncoding (kernel_grid_backward / kernel_grid_backward_input) that the
// reference reaches through torch autograd when it trains NGPRadianceField / Field
// (examples/train_finetune.py:465-533, examples/field.py:229-238).
//   * table gradient: every (point, level) scatter-adds weight_c * dL/dfeat into the 8 corner rows with no-return
//     fp32 atomics (global_atomic_add_f32), four lanes per (point, level) so that same-line atomics share a request.
//     Atomic sums depend on arrival order: results are reproducible to fp32 rounding, not bitwise.
//   * input gradient: dL/dx = sum_levels scale_l * sum_f dL/dfeat_f * sum_c (d weight_c / d frac) * table[c][f]
//     (linear interpolation: d pos / d x = scale), one lane per point, no atomics.
#include "field_common.h"

namespace {

__device__ __forceinline__ LevelConst level_const(const GridArgs &ga, int level)
{
    LevelConst lc;
    lc.offset = ga.offset[level];
    lc.rows = ga.rows[level];
    lc.res = ga.res[level];
    lc.scale = ga.scale[level];
    lc.hashed = (ga.hashed_mask >> level) & 1u;
    return lc;
}

// Table scatter: four lanes per (point, level), lane sub = (corner x bit, feature).  The four lanes of a quad address
// 16 contiguous bytes (rows idx, idx+1 x two features, whenever the x-neighbour is the next row) in the SAME atomic
// instruction, and the memory pipeline carries same-line lanes of one instruction as one request: 3.7 ms per 2^20
// points against 13.3 ms with one lane per (point, level) issuing its 16 atomics one after the other.
__global__ void grid_backward_table_kernel(GridArgs ga, const float *x01, const float *dfeat, int64_t n, float *grad_table)
{
    const int64_t total = n * QF_MAX_LEVELS * 4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int sub = (int)(e & 3), cx = sub >> 1, f = sub & 1;
        const int64_t pl = e >> 2, pt = pl >> 4;
        const int level = (int)(pl & 15);
        const float gf = dfeat[pt * 32 + 2 * level + f];
        if (gf == 0.0f) continue;
        const LevelConst lc = level_const(ga, level);
        uint32_t idx[8];
        float frac[3];
        level_indices(lc, x01[pt * 3], x01[pt * 3 + 1], x01[pt * 3 + 2], idx, frac);
        const float wx = cx ? frac[0] : 1.0f - frac[0], wy = frac[1], wz = frac[2];
#pragma unroll
        for (int yz = 0; yz < 4; ++yz) {
            const float w = (wx * ((yz & 1) ? wy : 1.0f - wy)) * ((yz & 2) ? wz : 1.0f - wz);
            atomicAdd(grad_table + 2 * (int64_t)idx[cx | (yz << 1)] + f, w * gf);
        }
    }
}

__global__ void grid_backward_input_kernel(GridArgs ga, const float2 *table, const float *x01, const float *dfeat,
                                           int64_t n, float *dx)
{
    for (int64_t pt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pt < n; pt += (int64_t)gridDim.x * blockDim.x) {
        const float x = x01[pt * 3], y = x01[pt * 3 + 1], z = x01[pt * 3 + 2];
        float gx = 0.0f, gy = 0.0f, gz = 0.0f;
        for (int level = 0; level < QF_MAX_LEVELS; ++level) {
            const float g0 = dfeat[pt * 32 + 2 * level], g1 = dfeat[pt * 32 + 2 * level + 1];
            const LevelConst lc = level_const(ga, level);
            uint32_t idx[8];
            float frac[3];
            level_indices(lc, x, y, z, idx, frac);
            const float wx = frac[0], wy = frac[1], wz = frac[2];
            float sx = 0.0f, sy = 0.0f, sz = 0.0f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float2 v = table[idx[c]];
                const float s = g0 * v.x + g1 * v.y;
                const float ax = (c & 1) ? wx : 1.0f - wx, ay = (c & 2) ? wy : 1.0f - wy, az = (c & 4) ? wz : 1.0f - wz;
                sx += ((c & 1) ? 1.0f : -1.0f) * ay * az * s;
                sy += ((c & 2) ? 1.0f : -1.0f) * ax * az * s;
                sz += ((c & 4) ? 1.0f : -1.0f) * ax * ay * s;
            }
            gx += lc.scale * sx;
            gy += lc.scale * sy;
            gz += lc.scale * sz;
        }
        dx[pt * 3 + 0] = gx;
        dx[pt * 3 + 1] = gy;
        dx[pt * 3 + 2] = gz;
    }
}

// Second order (Field.field_grad(create_graph=True), examples/field.py:229-238, and the losses on it): the backward
// of  gx = J(x; table)^T dfeat  given v = dL/dgx [n,3].  With D_l(c) = sum_d v_d scale_l (dw_c / dfrac_d), the
// derivative of corner c's weight along v:
//   dL/ddfeat_{l,f}      = sum_c D_l(c) table[c][f]                     (one lane per point, no atomics)
//   dL/dtable[c][f]     += D_l(c) dfeat_{l,f}                           (atomic scatter, like the first order)
//   dL/dx_e              = sum_l scale_l sum_{d != e} v_d scale_l sum_c (d2 w_c / dfrac_d dfrac_e) g_c,  g_c = dfeat . table[c]
// (the weights are products of one linear factor per axis: the pure second derivatives vanish, the mixed ones are
// sign_d sign_e times the third factor).
// table part of the second order: dL/dtable[c][f] += D_l(c) dfeat_{l,f}, four lanes per (point, level) like
// grid_backward_table_kernel
__global__ void grid_double_backward_table_kernel(GridArgs ga, const float *x01, const float *dfeat, const float *v,
                                                  int64_t n, float *grad_table)
{
    const int64_t total = n * QF_MAX_LEVELS * 4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int sub = (int)(e & 3), cx = sub >> 1, f = sub & 1;
        const int64_t pl = e >> 2, pt = pl >> 4;
        const int level = (int)(pl & 15);
        const float df = dfeat[pt * 32 + 2 * level + f];
        if (df == 0.0f) continue;
        const LevelConst lc = level_const(ga, level);
        uint32_t idx[8];
        float frac[3];
        level_indices(lc, x01[pt * 3], x01[pt * 3 + 1], x01[pt * 3 + 2], idx, frac);
        const float vx = v[pt * 3], vy = v[pt * 3 + 1], vz = v[pt * 3 + 2];
        const float wx = frac[0], wy = frac[1], wz = frac[2];
        const float ax = cx ? wx : 1.0f - wx, sx = cx ? 1.0f : -1.0f;
#pragma unroll
        for (int yz = 0; yz < 4; ++yz) {
            const float ay = (yz & 1) ? wy : 1.0f - wy, az = (yz & 2) ? wz : 1.0f - wz;
            const float sy = (yz & 1) ? 1.0f : -1.0f, sz = (yz & 2) ? 1.0f : -1.0f;
            const float D = lc.scale * (vx * sx * ay * az + vy * sy * ax * az + vz * sz * ax * ay);
            atomicAdd(grad_table + 2 * (int64_t)idx[cx | (yz << 1)] + f, D * df);
        }
    }
}

__global__ void grid_double_backward_kernel(GridArgs ga, const float2 *table, const float *x01, const float *dfeat,
                                            const float *v, int64_t n, float *g_dfeat, float *g_x)
{
    for (int64_t pt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pt < n; pt += (int64_t)gridDim.x * blockDim.x) {
        const float x = x01[pt * 3], y = x01[pt * 3 + 1], z = x01[pt * 3 + 2];
        const float vx = v[pt * 3], vy = v[pt * 3 + 1], vz = v[pt * 3 + 2];
        float hx = 0.0f, hy = 0.0f, hz = 0.0f;
        for (int level = 0; level < QF_MAX_LEVELS; ++level) {
            const float d0 = dfeat[pt * 32 + 2 * level], d1 = dfeat[pt * 32 + 2 * level + 1];
            const LevelConst lc = level_const(ga, level);
            uint32_t idx[8];
            float frac[3];
            level_indices(lc, x, y, z, idx, frac);
            const float wx = frac[0], wy = frac[1], wz = frac[2];
            float o0 = 0.0f, o1 = 0.0f, mxy = 0.0f, mxz = 0.0f, myz = 0.0f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float2 t = table[idx[c]];
                const float ax = (c & 1) ? wx : 1.0f - wx, ay = (c & 2) ? wy : 1.0f - wy, az = (c & 4) ? wz : 1.0f - wz;
                const float sx = (c & 1) ? 1.0f : -1.0f, sy = (c & 2) ? 1.0f : -1.0f, sz = (c & 4) ? 1.0f : -1.0f;
                const float D = lc.scale * (vx * sx * ay * az + vy * sy * ax * az + vz * sz * ax * ay);
                o0 += D * t.x;
                o1 += D * t.y;
                const float g = d0 * t.x + d1 * t.y;
                mxy += sx * sy * az * g;
                mxz += sx * sz * ay * g;
                myz += sy * sz * ax * g;
            }
            if (g_dfeat) { g_dfeat[pt * 32 + 2 * level] = o0; g_dfeat[pt * 32 + 2 * level + 1] = o1; }
            const float s2 = lc.scale * lc.scale;
            hx += s2 * (vy * mxy + vz * mxz);
            hy += s2 * (vx * mxy + vz * myz);
            hz += s2 * (vx * mxz + vy * myz);
        }
        if (g_x) { g_x[pt * 3] = hx; g_x[pt * 3 + 1] = hy; g_x[pt * 3 + 2] = hz; }
    }
}

}  // namespace

extern "C" int qf_grid_encode_backward(const qf_grid_desc *desc, const float *table, const float *x01,
                                       const float *dfeat, int64_t n, float *grad_table, float *grad_x01, void *stream)
{
    if (!desc || n < 0) return QF_ERR_INVALID_ARGUMENT;
    GridArgs ga;
    int rc = fill_grid_args(desc, &ga);
    if (rc != QF_OK) return rc;
    if (n == 0) return QF_OK;
    if (!x01 || !dfeat || (!grad_table && !grad_x01) || (grad_x01 && !table)) return QF_ERR_INVALID_ARGUMENT;
    if (grad_table) {
        hipLaunchKernelGGL(grid_backward_table_kernel, dim3(qf_grid_1d(n * 64, 256, 32)), dim3(256), 0, qf_stream(stream),
                           ga, x01, dfeat, n, grad_table);
        QF_LAUNCH_CHECK();
    }
    if (grad_x01) {
        hipLaunchKernelGGL(grid_backward_input_kernel, dim3(qf_grid_1d(n, 256)), dim3(256), 0, qf_stream(stream), ga,
                           reinterpret_cast<const float2 *>(table), x01, dfeat, n, grad_x01);
        QF_LAUNCH_CHECK();
    }
    return QF_OK;
}

extern "C" int qf_grid_encode_double_backward(const qf_grid_desc *desc, const float *table, const float *x01,
                                              const float *dfeat, const float *v, int64_t n, float *g_dfeat,
                                              float *g_x01, float *grad_table, void *stream)
{
    if (!desc || n < 0) return QF_ERR_INVALID_ARGUMENT;
    GridArgs ga;
    int rc = fill_grid_args(desc, &ga);
    if (rc != QF_OK) return rc;
    if (n == 0) return QF_OK;
    if (!table || !x01 || !dfeat || !v || (!g_dfeat && !g_x01 && !grad_table)) return QF_ERR_INVALID_ARGUMENT;
    if (g_dfeat || g_x01) {
        hipLaunchKernelGGL(grid_double_backward_kernel, dim3(qf_grid_1d(n, 256)), dim3(256), 0, qf_stream(stream), ga,
                           reinterpret_cast<const float2 *>(table), x01, dfeat, v, n, g_dfeat, g_x01);
        QF_LAUNCH_CHECK();
    }
    if (grad_table) {
        hipLaunchKernelGGL(grid_double_backward_table_kernel, dim3(qf_grid_1d(n * 64, 256, 32)), dim3(256), 0,
                           qf_stream(stream), ga, x01, dfeat, v, n, grad_table);
        QF_LAUNCH_CHECK();
    }
    return QF_OK;
}
