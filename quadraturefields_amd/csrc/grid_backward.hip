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
// LevelSel: the levels this launch serves (all of them, or the ones the partitioned scatter below leaves to the atomics).
struct LevelSel {
    int n;
    int8_t level[QF_MAX_LEVELS];
};

__global__ void grid_backward_table_kernel(GridArgs ga, LevelSel sel, const float *x01, const float *dfeat, int64_t n,
                                           float *grad_table)
{
    const int64_t total = n * sel.n * 4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int sub = (int)(e & 3), cx = sub >> 1, f = sub & 1;
        const int64_t pl = e >> 2, pt = pl / sel.n;
        const int level = sel.level[(int)(pl - pt * sel.n)];
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

// ---------------------------------------------------------------------------------------------------------------------
// Table scatter WITHOUT global atomics on the data path ("LDS-partitioned scatter").  Scattered no-return atomics run
// at the memory side at ~1.9e10 requests/s chip-wide whatever the kernel does (MI355X_MICROARCH.md, "Global float
// atomics"; the quad kernel above sits at 1.8e10), so the table gradient of 2^20 points costs 3.7 ms.  Here every
// level's rows are cut into partitions of kLdsRows rows (160 KB of LDS = 20 000 float2 accumulators); a workgroup owns
// one (level, partition, point chunk): it walks ITS chunk of the points, recomputes their corner rows for that level
// and adds the contributions that fall into its partition to the LDS accumulators (ds_add_f32); at the end it adds
// its accumulators to the gradient table with CONTIGUOUS atomics (256 B per wave instruction: full rate, 1.3 TB/s).
// The price is that the points are read once per partition -- 8-byte dfeat pairs in level-major order (transposed
// once into the workspace) and 12-byte positions, out of L2 / the Infinity Cache -- and the corner rows are hashed
// once per partition; both are cheap next to a scattered atomic.  What bounds the kernel is the LDS atomic pipe, so
// the work is dealt by ATOMICS per workgroup: a coarse dense level is one partition in which every corner of every
// point lands, and gets ~100 point chunks; a hashed level's 26 partitions get 4 each (with 3 chunks everywhere the
// three workgroups of level 0 ran 7.1 ms while the rest of the chip idled).  1.97 ms per 2^20 points, T = 2^19,
// against 3.69 ms for the quad-atomic kernel (tools/train_bench.py).
//
// The walk is O(partitions x points) per level: ~2.9 us per partition and 10^6 points, against 0.21 ms per level and
// 10^6 points for the quad atomics (4 requests per point and level at 1.9e10/s).  A level therefore takes this route
// only while it has at most kMaxLdsParts partitions; above that -- every hashed level of the deformation field's
// T = 2^24 table: 839 partitions each -- it goes to grid_backward_table_kernel.  Round 2 sent ALL levels here and the
// reference-sized deformation table (train_finetune.py:387-399, 101.6 M rows = 5 300 partitions) cost 13.4 ms per
// 0.8 M points; with the split 1.3 ms.
constexpr int kLdsRows = 20000;
constexpr int kMaxLdsParts = 64;
constexpr int kScatterThreads = 1024;
constexpr int kScatterUnroll = 8;

struct ScatterPlan {
    int first_item[QF_MAX_LEVELS + 1];   // workgroups [first_item[l], first_item[l+1]) serve level l
    int chunks[QF_MAX_LEVELS];           // point chunks per partition of level l: a level with few partitions (coarse,
                                         // dense: EVERY corner of every point is an LDS atomic of its one partition)
                                         // gets many chunks, so that no workgroup carries more atomics than the others
};

// dfeat [n][32] -> level-major [16][n] float2 (one coalesced 512-B segment per level and 64 points)
__global__ __launch_bounds__(256) void dfeat_level_major_kernel(const float *__restrict__ dfeat, int64_t n, float2 *__restrict__ out)
{
    __shared__ float tile[64][33];
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    for (int e = threadIdx.x; e < 64 * 32; e += 256) {
        const int r = e >> 5, c = e & 31;
        tile[r][c] = (p0 + r < n) ? dfeat[(p0 + r) * 32 + c] : 0.0f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 16 * 64; e += 256) {
        const int l = e >> 6, r = e & 63;
        if (p0 + r < n) out[(int64_t)l * n + p0 + r] = make_float2(tile[r][2 * l], tile[r][2 * l + 1]);
    }
}

// kSecond: the table part of the SECOND order (grid_double_backward_table_kernel's sums): corner c contributes
// D_l(c) * dfeat instead of w_c * dfeat, D_l(c) = scale_l * sum_d v_d (dw_c / dfrac_d) with v [n,3] = dL/d(field_grad).
template <bool kSecond>
__global__ __launch_bounds__(kScatterThreads) void grid_backward_table_lds_kernel(GridArgs ga, ScatterPlan plan,
                                                                                const float *__restrict__ x01,
                                                                                const float2 *__restrict__ dfeat_lm,
                                                                                const float *__restrict__ v,
                                                                                int64_t n, float *__restrict__ grad_table)
{
    extern __shared__ float2 acc[];
    int level = 0;
    while (level < QF_MAX_LEVELS - 1 && (int)blockIdx.x >= plan.first_item[level + 1]) ++level;
    const int local = (int)blockIdx.x - plan.first_item[level];
    const int n_chunks = plan.chunks[level];
    const int part = local / n_chunks, chunk = local - part * n_chunks;
    const LevelConst lc = level_const(ga, level);
    const uint32_t row_lo = (uint32_t)part * kLdsRows;
    const uint32_t row_cnt = lc.rows - row_lo < (uint32_t)kLdsRows ? lc.rows - row_lo : (uint32_t)kLdsRows;
    for (uint32_t i = threadIdx.x; i < row_cnt; i += kScatterThreads) acc[i] = make_float2(0.0f, 0.0f);
    __syncthreads();
    const int64_t p_lo = n * chunk / n_chunks, p_hi = n * (chunk + 1) / n_chunks;
    const float2 *gl = dfeat_lm + (int64_t)level * n;
    // kScatterUnroll points per lane and trip, all their loads issued before the first is used (the walk is a chain of
    // dependent L2 / Infinity-Cache reads otherwise)
    for (int64_t base = p_lo + threadIdx.x; base < p_hi; base += (int64_t)kScatterUnroll * kScatterThreads) {
        float2 gf[kScatterUnroll];
        float px[kScatterUnroll], py[kScatterUnroll], pz[kScatterUnroll];
        float vx[kSecond ? kScatterUnroll : 1], vy[kSecond ? kScatterUnroll : 1], vz[kSecond ? kScatterUnroll : 1];
#pragma unroll
        for (int u = 0; u < kScatterUnroll; ++u) {
            const int64_t pt = base + (int64_t)u * kScatterThreads;
            const bool ok = pt < p_hi;
            const int64_t q = ok ? pt : p_lo;
            gf[u] = gl[q];
            if (!ok) gf[u] = make_float2(0.0f, 0.0f);
            px[u] = x01[q * 3];
            py[u] = x01[q * 3 + 1];
            pz[u] = x01[q * 3 + 2];
            if (kSecond) { vx[u] = v[q * 3]; vy[u] = v[q * 3 + 1]; vz[u] = v[q * 3 + 2]; }
        }
#pragma unroll
        for (int u = 0; u < kScatterUnroll; ++u) {
            uint32_t idx[8];
            float frac[3];
            level_indices(lc, px[u], py[u], pz[u], idx, frac);
            // corners of this point that fall into the partition, as a bit mask; a lane has 8 / partitions of them on
            // average, so the wave works the masks off one set bit per trip (2-3 trips) instead of issuing 16 LDS
            // atomics with one or two lanes each -- the LDS atomic pipe, not memory, bounds this kernel
            unsigned mask = 0;
            if (gf[u].x != 0.0f || gf[u].y != 0.0f) {
#pragma unroll
                for (int c = 0; c < 8; ++c) mask |= (idx[c] - lc.offset - row_lo < row_cnt) ? (1u << c) : 0u;
            }
            while (__any(mask != 0)) {
                if (mask) {
                    const int c = __ffs(mask) - 1;
                    mask &= mask - 1;
                    uint32_t id = idx[0];
#pragma unroll
                    for (int k = 1; k < 8; ++k) id = (c == k) ? idx[k] : id;
                    const uint32_t r = id - lc.offset - row_lo;
                    float w;
                    if (kSecond) {                     // the expression of grid_double_backward_table_kernel
                        const float ax = (c & 1) ? frac[0] : 1.0f - frac[0], sx = (c & 1) ? 1.0f : -1.0f;
                        const float ay = (c & 2) ? frac[1] : 1.0f - frac[1], sy = (c & 2) ? 1.0f : -1.0f;
                        const float az = (c & 4) ? frac[2] : 1.0f - frac[2], sz = (c & 4) ? 1.0f : -1.0f;
                        w = lc.scale * (vx[kSecond ? u : 0] * sx * ay * az + vy[kSecond ? u : 0] * sy * ax * az +
                                        vz[kSecond ? u : 0] * sz * ax * ay);
                    } else {
                        w = (((c & 1) ? frac[0] : 1.0f - frac[0]) * ((c & 2) ? frac[1] : 1.0f - frac[1])) *
                            ((c & 4) ? frac[2] : 1.0f - frac[2]);
                    }
                    atomicAdd(&acc[r].x, w * gf[u].x);
                    atomicAdd(&acc[r].y, w * gf[u].y);
                }
            }
        }
    }
    __syncthreads();
    float *dst = grad_table + 2 * ((int64_t)lc.offset + row_lo);
    const float *src = reinterpret_cast<const float *>(acc);
    for (uint32_t i = threadIdx.x; i < 2 * row_cnt; i += kScatterThreads) {
        const float v = src[i];
        if (v != 0.0f) atomicAdd(dst + i, v);          // contiguous across the wave: full-rate atomics
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
__global__ void grid_double_backward_table_kernel(GridArgs ga, LevelSel sel, const float *x01, const float *dfeat,
                                                  const float *v, int64_t n, float *grad_table)
{
    const int64_t total = n * sel.n * 4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int sub = (int)(e & 3), cx = sub >> 1, f = sub & 1;
        const int64_t pl = e >> 2, pt = pl / sel.n;
        const int level = sel.level[(int)(pl - pt * sel.n)];
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
        LevelSel all;
        all.n = QF_MAX_LEVELS;
        for (int l = 0; l < QF_MAX_LEVELS; ++l) all.level[l] = (int8_t)l;
        hipLaunchKernelGGL(grid_backward_table_kernel, dim3(qf_grid_1d(n * 64, 256, 32)), dim3(256), 0, qf_stream(stream),
                           ga, all, x01, dfeat, n, grad_table);
        QF_LAUNCH_CHECK();
    }
    if (grad_x01) {
        hipLaunchKernelGGL(grid_backward_input_kernel, dim3(qf_grid_1d(n, 256)), dim3(256), 0, qf_stream(stream), ga,
                           reinterpret_cast<const float2 *>(table), x01, dfeat, n, grad_x01);
        QF_LAUNCH_CHECK();
    }
    return QF_OK;
}

extern "C" int64_t qf_grid_backward_workspace_bytes(int64_t n) { return n < 0 ? -1 : n * 32 * (int64_t)sizeof(float); }

// The table scatter of a large batch, first (v == NULL) or second order: levels with few partitions through the LDS walk,
// the others through the quad atomics (see kMaxLdsParts).  workspace: qf_grid_backward_workspace_bytes(n) bytes.
static int table_scatter_ws(const GridArgs &ga, const float *x01, const float *dfeat, const float *v, int64_t n,
                            float *grad_table, void *workspace, hipStream_t st)
{
    float2 *lm = reinterpret_cast<float2 *>(workspace);
    ScatterPlan plan;
    // ~96 workgroups per level on 256 CUs (measured 32 / 64 / 96 / 128 / 256 per level: 2.35 / 2.05 / 1.96 / 1.98 / 2.07 ms)
    const int per_level = 3 * qf_cu_count_cached() / 8 > 16 ? 3 * qf_cu_count_cached() / 8 : 16;
    int item = 0;
    LevelSel atomics;                  // levels with too many partitions for the walk: quad atomics
    atomics.n = 0;
    for (int l = 0; l < QF_MAX_LEVELS; ++l) {
        const int parts = (int)qf_div_up(ga.rows[l], kLdsRows);
        plan.first_item[l] = item;
        if (parts > kMaxLdsParts) {
            plan.chunks[l] = 1;        // no workgroups: first_item[l + 1] == first_item[l]
            atomics.level[atomics.n++] = (int8_t)l;
            continue;
        }
        int chunks = (per_level + parts / 2) / parts;
        if (chunks < 1) chunks = 1;
        plan.chunks[l] = chunks;
        item += parts * chunks;
    }
    plan.first_item[QF_MAX_LEVELS] = item;
    if (atomics.n > 0) {
        if (v)
            hipLaunchKernelGGL(grid_double_backward_table_kernel, dim3(qf_grid_1d(n * atomics.n * 4, 256, 32)), dim3(256), 0, st,
                               ga, atomics, x01, dfeat, v, n, grad_table);
        else
            hipLaunchKernelGGL(grid_backward_table_kernel, dim3(qf_grid_1d(n * atomics.n * 4, 256, 32)), dim3(256), 0, st, ga,
                               atomics, x01, dfeat, n, grad_table);
        QF_LAUNCH_CHECK();
    }
    if (item == 0) return QF_OK;
    hipLaunchKernelGGL(dfeat_level_major_kernel, dim3((unsigned)qf_div_up(n, 64)), dim3(256), 0, st, dfeat, n, lm);
    QF_LAUNCH_CHECK();
    const size_t lds = (size_t)kLdsRows * sizeof(float2);
    if (v) {
        static QfLdsAttr attr2;                      // per device
        QF_HIP_TRY(qf_ensure_dynamic_lds(attr2, reinterpret_cast<const void *>(grid_backward_table_lds_kernel<true>), lds));
        hipLaunchKernelGGL(grid_backward_table_lds_kernel<true>, dim3((unsigned)item), dim3(kScatterThreads), lds, st, ga, plan,
                           x01, lm, v, n, grad_table);
    } else {
        static QfLdsAttr attr;                       // per device
        QF_HIP_TRY(qf_ensure_dynamic_lds(attr, reinterpret_cast<const void *>(grid_backward_table_lds_kernel<false>), lds));
        hipLaunchKernelGGL(grid_backward_table_lds_kernel<false>, dim3((unsigned)item), dim3(kScatterThreads), lds, st, ga, plan,
                           x01, lm, (const float *)nullptr, n, grad_table);
    }
    QF_LAUNCH_CHECK();
    return QF_OK;
}

extern "C" int qf_grid_encode_backward_ws(const qf_grid_desc *desc, const float *table, const float *x01,
                                          const float *dfeat, int64_t n, float *grad_table, float *grad_x01,
                                          void *workspace, int64_t workspace_bytes, void *stream)
{
    // small batches: the quad-atomic kernel (the partitioned scatter re-reads the points once per partition and
    // flushes 160 KB per workgroup: it only pays from a few tens of thousands of points on)
    if (!grad_table || !workspace || n < (1 << 15) || workspace_bytes < qf_grid_backward_workspace_bytes(n))
        return qf_grid_encode_backward(desc, table, x01, dfeat, n, grad_table, grad_x01, stream);
    if (!desc) return QF_ERR_INVALID_ARGUMENT;
    GridArgs ga;
    int rc = fill_grid_args(desc, &ga);
    if (rc != QF_OK) return rc;
    if (!x01 || !dfeat || (grad_x01 && !table)) return QF_ERR_INVALID_ARGUMENT;
    hipStream_t st = qf_stream(stream);
    rc = table_scatter_ws(ga, x01, dfeat, nullptr, n, grad_table, workspace, st);
    if (rc != QF_OK) return rc;
    if (grad_x01) {
        hipLaunchKernelGGL(grid_backward_input_kernel, dim3(qf_grid_1d(n, 256)), dim3(256), 0, st, ga,
                           reinterpret_cast<const float2 *>(table), x01, dfeat, n, grad_x01);
        QF_LAUNCH_CHECK();
    }
    return QF_OK;
}

extern "C" int qf_grid_encode_double_backward(const qf_grid_desc *desc, const float *table, const float *x01,
                                              const float *dfeat, const float *v, int64_t n, float *g_dfeat,
                                              float *g_x01, float *grad_table, void *workspace, int64_t workspace_bytes,
                                              void *stream)
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
        if (workspace && n >= (1 << 15) && workspace_bytes >= qf_grid_backward_workspace_bytes(n)) {
            rc = table_scatter_ws(ga, x01, dfeat, v, n, grad_table, workspace, qf_stream(stream));
            if (rc != QF_OK) return rc;
        } else {
            LevelSel all;
            all.n = QF_MAX_LEVELS;
            for (int l = 0; l < QF_MAX_LEVELS; ++l) all.level[l] = (int8_t)l;
            hipLaunchKernelGGL(grid_double_backward_table_kernel, dim3(qf_grid_1d(n * 64, 256, 32)), dim3(256), 0,
                               qf_stream(stream), ga, all, x01, dfeat, v, n, grad_table);
            QF_LAUNCH_CHECK();
        }
    }
    return QF_OK;
}
