// bf16 variant of the fused field evaluation (BASELINE config 3: bf16 tables and MLPs, fp32 accumulate).
//
// Same wave mapping as field_eval.hip -- lane (p = l & 15, g = l >> 4): point p of a 16-point group, level quartet
// g -- but the hash tables hold bf16x2 rows (4 B gathers) and every layer is ONE OR TWO v_mfma_f32_16x16x32_bf16:
// the B operand of that instruction is 8 consecutive k per lane quartet, which is exactly the 8 grid features
// (4 levels x 2) a lane has just blended, or 8 of the 16 hidden activations it holds after the previous layer
// (C/D layout: register r of lane (g,p) = neuron 16*mt + 4g + r of point p).  So, as in the fp32 kernel, activations
// go from accumulator to next operand without leaving the lane; they are rounded to bf16 (RNE) on the way.
// 20 MFMAs per 16 points instead of 160.  Weights: bf16 copies of the reference's fp32 parameters, re-laid once per
// workgroup into per-lane operand order in LDS (20 KB).
#include "field_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int kBlockB = 512;

struct FieldArgsB {
    GridArgs grid;
    float aabb_lo[3];
    float aabb_hi[3];
    const uint32_t *table;      // [rows] bf16x2: low half = feature 0
    const uint16_t *base_w;     // bf16 [64*32 | 16*64]
    const uint16_t *head_w;     // bf16 NGP head [64*32 | 64*64 | 16*64]
    const uint16_t *sg_w1, *sg_b1, *sg_w2, *sg_wout;   // bf16
    const float *sg_b2, *sg_bout;                      // fp32 (accumulator init)
    const float *xyz;
    const float *dirs;
    int64_t n;
    const int64_t *n_dev;   // optional device-side point count, see FieldArgs (field_eval.hip)
    const int32_t *order;
    float *rgb;
    float *sigma;
    float *geo;
    int32_t n_lobes, n_out, nt_out;
};

__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;       // round to nearest even (finite inputs)
}

__device__ __forceinline__ uint32_t pack2(float lo, float hi) { return f32_to_bf16_bits(lo) | (f32_to_bf16_bits(hi) << 16); }

__device__ __forceinline__ bf16x8 pack8(const float v[8])
{
    const uint4 u = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
    return __builtin_bit_cast(bf16x8, u);
}

__device__ __forceinline__ f32x4 mfma_bf16(uint4 a, bf16x8 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), b, c, 0, 0, 0);
}

// column of a 64-wide hidden input fed as element j of lane quartet kq at k-step s (s = 0, 1)
__device__ __forceinline__ int hidden_col_b(int s, int kq, int j) { return 16 * (2 * s + (j >> 2)) + 4 * kq + (j & 3); }

// bf16 bits of the weight that lane `lane` feeds as element j of the A operand of MFMA number m
template <int HEAD>
__device__ uint32_t weight_for_b(const FieldArgsB &a, int m, int lane, int j)
{
    const int i = lane & 15, kq = lane >> 4;
    if (m < 4) return a.base_w[(16 * m + i) * 32 + 2 * (4 * (j >> 1) + kq) + (j & 1)];          // 32 -> 64
    if (m < 6) return a.base_w[2048 + i * 64 + hidden_col_b(m - 4, kq, j)];                      // 64 -> 16
    m -= 6;
    if (HEAD == QF_HEAD_NGP) {
        if (m < 4) {                                                                             // [SH16|geo15|1] -> 64
            int col;
            if (j < 4) col = 4 * kq + j;
            else { const int o = 4 * kq + (j - 4); col = (o == 0) ? 31 : 15 + o; }
            return a.head_w[(16 * m + i) * 32 + col];
        }
        if (m < 12) { const int q = m - 4, s = q >> 2, mt = q & 3; return a.head_w[2048 + (16 * mt + i) * 64 + hidden_col_b(s, kq, j)]; }
        return a.head_w[2048 + 4096 + i * 64 + hidden_col_b(m - 12, kq, j)];                     // 64 -> 16
    }
    if (HEAD == QF_HEAD_SG) {
        if (m < 4) {                                                                             // [geo15|bias] (+16 zero k) -> 64
            if (j >= 4) return 0u;
            const int row = 16 * m + i, o = 4 * kq + j;
            return (o == 0) ? a.sg_b1[row] : a.sg_w1[row * 15 + (o - 1)];
        }
        if (m < 12) { const int q = m - 4, s = q >> 2, mt = q & 3; return a.sg_w2[(16 * mt + i) * 64 + hidden_col_b(s, kq, j)]; }
        const int q = m - 12, mt = q >> 1, s = q & 1, row = 16 * mt + i;
        return (row < a.n_out) ? a.sg_wout[row * 64 + hidden_col_b(s, kq, j)] : 0u;
    }
    return 0u;
}

template <int HEAD>
__device__ __forceinline__ int n_mfma_b(const FieldArgsB &a)
{
    if (HEAD == QF_HEAD_NGP) return 20;
    if (HEAD == QF_HEAD_SG) return 18 + 2 * a.nt_out;
    return 6;
}

__device__ __forceinline__ void relu8(const f32x4 &a, const f32x4 &b, float out[8])
{
#pragma unroll
    for (int r = 0; r < 4; ++r) { out[r] = fmaxf(a[r], 0.0f); out[4 + r] = fmaxf(b[r], 0.0f); }
}

template <int HEAD>
__global__ __launch_bounds__(kBlockB, 4) void field_kernel_bf16(const FieldArgsB a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t ldsb[];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, p = lane & 15;

    // A-operand images: [m][lane] uint4 (8 bf16), then fp32 biases (SG), then the level table
    const int n_m = n_mfma_b<HEAD>(a);
    for (int e = tid; e < n_m * 64 * 4; e += kBlockB) {
        const int m = e >> 8, l = (e >> 2) & 63, w = e & 3;
        ldsb[e] = weight_for_b<HEAD>(a, m, l, 2 * w) | (weight_for_b<HEAD>(a, m, l, 2 * w + 1) << 16);
    }
    float *bias_lds = reinterpret_cast<float *>(ldsb + n_m * 256);
    if (HEAD == QF_HEAD_SG) {
        if (tid < 64) bias_lds[tid] = a.sg_b2[tid];
        else if (tid < 128) bias_lds[tid] = (tid - 64 < a.n_out) ? a.sg_bout[tid - 64] : 0.0f;
    }
    uint32_t *lvl_lds = reinterpret_cast<uint32_t *>(bias_lds + 128);
    if (tid < QF_MAX_LEVELS) {
        lvl_lds[tid * 8 + 0] = a.grid.offset[tid];
        lvl_lds[tid * 8 + 1] = a.grid.rows[tid];
        lvl_lds[tid * 8 + 2] = a.grid.res[tid];
        lvl_lds[tid * 8 + 3] = (a.grid.hashed_mask >> tid) & 1u;
        lvl_lds[tid * 8 + 4] = __float_as_uint(a.grid.scale[tid]);
    }
    __syncthreads();
    const uint4 *img_base = reinterpret_cast<const uint4 *>(ldsb);

    // one contiguous eighth of the processing order per XCD (see field_kernel in field_eval.hip)
    int64_t n_pts = a.n;
    if (a.n_dev) { const int64_t nd = *a.n_dev; n_pts = nd < a.n ? (nd > 0 ? nd : 0) : a.n; }
    const int64_t n_groups = (n_pts + 15) >> 4;
    int64_t grp_begin, grp_end, wave_stride;
    if ((gridDim.x & 7) == 0) {
        const int64_t per_xcd = (n_groups + 7) >> 3;
        grp_begin = (int64_t)(blockIdx.x & 7) * per_xcd;
        grp_end = grp_begin + per_xcd < n_groups ? grp_begin + per_xcd : n_groups;
        grp_begin += (int64_t)(blockIdx.x >> 3) * (kBlockB / 64) + (tid >> 6);
        wave_stride = (int64_t)(gridDim.x >> 3) * (kBlockB / 64);
    } else {
        grp_begin = (int64_t)blockIdx.x * (kBlockB / 64) + (tid >> 6);
        grp_end = n_groups;
        wave_stride = (int64_t)gridDim.x * (kBlockB / 64);
    }
    for (int64_t grp = grp_begin; grp < grp_end; grp += wave_stride) {
        const int64_t pt_raw = grp * 16 + p;
        const bool valid = pt_raw < n_pts;
        int64_t pt = valid ? pt_raw : n_pts - 1;
        if (a.order) pt = a.order[pt];
        const float X = a.xyz[pt * 3 + 0], Y = a.xyz[pt * 3 + 1], Z = a.xyz[pt * 3 + 2];
        const float x01 = (X - a.aabb_lo[0]) / (a.aabb_hi[0] - a.aabb_lo[0]);
        const float y01 = (Y - a.aabb_lo[1]) / (a.aabb_hi[1] - a.aabb_lo[1]);
        const float z01 = (Z - a.aabb_lo[2]) / (a.aabb_hi[2] - a.aabb_lo[2]);
        const bool selector = x01 > 0.0f && x01 < 1.0f && y01 > 0.0f && y01 < 1.0f && z01 > 0.0f && z01 < 1.0f;

        int loff = lane, goff = g * 8;
        asm volatile("" : "+v"(loff), "+v"(goff));
        const uint4 *img = img_base + loff;

        float frac[4][3];
        uint32_t raw[4][8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t *lv = lvl_lds + 32 * j + goff;
            LevelConst lc;
            lc.offset = lv[0]; lc.rows = lv[1]; lc.res = lv[2]; lc.hashed = lv[3];
            lc.scale = __uint_as_float(lv[4]);
            uint32_t idx[8];
            level_indices(lc, x01, y01, z01, idx, frac[j]);
#pragma unroll
            for (int c = 0; c < 8; ++c) raw[j][c] = a.table[idx[c]];
        }
        float feat[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float2 val[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                val[c].x = __uint_as_float(raw[j][c] << 16);
                val[c].y = __uint_as_float(raw[j][c] & 0xffff0000u);
            }
            level_blend(val, frac[j], &feat[2 * j], &feat[2 * j + 1]);
        }

        // ---- base MLP
        const bf16x8 xb = pack8(feat);
        f32x4 h[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) h[mt] = mfma_bf16(img[mt * 64], xb, (f32x4){0.f, 0.f, 0.f, 0.f});
        float hv[8];
        f32x4 bo = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            relu8(h[2 * s], h[2 * s + 1], hv);
            bo = mfma_bf16(img[(4 + s) * 64], pack8(hv), bo);
        }
        const float density = selector ? expf(bo[0] - 1.0f) : 0.0f;
        if (g == 0 && valid && a.sigma) a.sigma[pt] = density;
        if (a.geo && valid) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = 4 * g + r;
                if (o > 0) a.geo[pt * 15 + (o - 1)] = bo[r];
            }
        }

        if (HEAD == QF_HEAD_NGP) {
            const float dx = a.dirs[pt * 3 + 0], dy = a.dirs[pt * 3 + 1], dz = a.dirs[pt * 3 + 2];
            const float ux = ((dx + 1.0f) / 2.0f) * 2.0f - 1.0f;
            const float uy = ((dy + 1.0f) / 2.0f) * 2.0f - 1.0f;
            const float uz = ((dz + 1.0f) / 2.0f) * 2.0f - 1.0f;
            float in[8];
            sh4_quartet(g, ux, uy, uz, in);
#pragma unroll
            for (int r = 0; r < 4; ++r) in[4 + r] = bo[r];
            if (g == 0) in[4] = 1.0f;
            const bf16x8 ib = pack8(in);
            f32x4 h1[4], h2[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                h1[mt] = mfma_bf16(img[(6 + mt) * 64], ib, (f32x4){0.f, 0.f, 0.f, 0.f});
                h2[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                relu8(h1[2 * s], h1[2 * s + 1], hv);
                const bf16x8 hb = pack8(hv);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) h2[mt] = mfma_bf16(img[(10 + 4 * s + mt) * 64], hb, h2[mt]);
            }
            f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                relu8(h2[2 * s], h2[2 * s + 1], hv);
                c = mfma_bf16(img[(18 + s) * 64], pack8(hv), c);
            }
            if (g == 0 && valid) {
                a.rgb[pt * 3 + 0] = sigmoidf(c[0]);
                a.rgb[pt * 3 + 1] = sigmoidf(c[1]);
                a.rgb[pt * 3 + 2] = sigmoidf(c[2]);
            }
        }

        if (HEAD == QF_HEAD_SG) {
            float in[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { in[r] = bo[r]; in[4 + r] = 0.0f; }
            if (g == 0) in[0] = 1.0f;
            const bf16x8 ib = pack8(in);
            const f32x4 *b2v = reinterpret_cast<const f32x4 *>(bias_lds);
            const f32x4 *bov = reinterpret_cast<const f32x4 *>(bias_lds + 64);
            f32x4 h1[4], h2[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                h1[mt] = mfma_bf16(img[(6 + mt) * 64], ib, (f32x4){0.f, 0.f, 0.f, 0.f});
                h2[mt] = b2v[4 * mt + g];
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                relu8(h1[2 * s], h1[2 * s + 1], hv);
                const bf16x8 hb = pack8(hv);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) h2[mt] = mfma_bf16(img[(10 + 4 * s + mt) * 64], hb, h2[mt]);
            }
            float hv0[8], hv1[8];
            relu8(h2[0], h2[1], hv0);
            relu8(h2[2], h2[3], hv1);
            const bf16x8 hb0 = pack8(hv0), hb1 = pack8(hv1);
            f32x4 out[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                out[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (mt < a.nt_out) {
                    out[mt] = mfma_bf16(img[(18 + 2 * mt) * 64], hb0, bov[4 * mt + g]);
                    out[mt] = mfma_bf16(img[(19 + 2 * mt) * 64], hb1, out[mt]);
                }
            }
            const float dx = a.dirs[pt * 3 + 0], dy = a.dirs[pt * 3 + 1], dz = a.dirs[pt * 3 + 2];
            auto fetch = [&](int o) -> float { return __shfl(out[o >> 4][o & 3], p + 16 * ((o >> 2) & 3), 64); };
            float acc_r = 0.0f, acc_g = 0.0f, acc_b = 0.0f;
#pragma unroll
            for (int l = 0; l < QF_MAX_LOBES; ++l) {
                if (l < a.n_lobes) {
                    const int o = 3 + 7 * l;
                    const float ax = fetch(o), ay = fetch(o + 1), az = fetch(o + 2);
                    const float lam = fabsf(fetch(o + 3));
                    const float cr = fetch(o + 4), cg = fetch(o + 5), cb = fetch(o + 6);
                    const float nrm = sqrtf(ax * ax + ay * ay + az * az);
                    const float dotp = (ax / nrm) * dx + (ay / nrm) * dy + (az / nrm) * dz;
                    const float e = expf(lam * (dotp - 1.0f));
                    acc_r += cr * e;
                    acc_g += cg * e;
                    acc_b += cb * e;
                }
            }
            const float d0 = fetch(0), d1 = fetch(1), d2 = fetch(2);
            if (g == 0 && valid) {
                a.rgb[pt * 3 + 0] = sigmoidf(d0 + acc_r);
                a.rgb[pt * 3 + 1] = sigmoidf(d1 + acc_g);
                a.rgb[pt * 3 + 2] = sigmoidf(d2 + acc_b);
            }
        }
    }
}

template <int HEAD>
int launch_field_b(const FieldArgsB &a, hipStream_t st)
{
    int n_m = 6;
    if (HEAD == QF_HEAD_NGP) n_m = 20;
    if (HEAD == QF_HEAD_SG) n_m = 18 + 2 * a.nt_out;
    const size_t lds_bytes = (size_t)n_m * 1024 + (128 + 8 * QF_MAX_LEVELS) * sizeof(float);
    int64_t blocks = qf_div_up((a.n + 15) / 16, kBlockB / 64);
    const int64_t cap = (int64_t)qf_cu_count_cached();   // one workgroup per CU, see launch_field in field_eval.hip
    if (blocks > cap) blocks = cap;
    if (blocks >= 64) blocks &= ~(int64_t)7;
    hipLaunchKernelGGL(field_kernel_bf16<HEAD>, dim3((unsigned)blocks), dim3(kBlockB), lds_bytes, st, a);
    QF_LAUNCH_CHECK();
    return QF_OK;
}

}  // namespace

extern "C" int qf_field_forward_bf16(const qf_field_desc *desc, const uint16_t *table, const uint16_t *base_w,
                                     const uint16_t *head_ngp_w, const qf_sg_head_bf16 *head_sg, const float *xyz,
                                     const float *dirs, int64_t n, const int64_t *n_device, const int32_t *order,
                                     float *rgb, float *sigma, float *geo, void *stream)
{
    if (!desc || !table || !base_w || n < 0 || n > 0x7fffffff) return QF_ERR_INVALID_ARGUMENT;
    FieldArgsB a = {};
    int rc = fill_grid_args(&desc->grid, &a.grid);
    if (rc != QF_OK) return rc;
    for (int k = 0; k < 3; ++k) {
        a.aabb_lo[k] = desc->aabb[k];
        a.aabb_hi[k] = desc->aabb[3 + k];
        if (!(desc->aabb[3 + k] > desc->aabb[k])) return QF_ERR_INVALID_ARGUMENT;
    }
    a.table = reinterpret_cast<const uint32_t *>(table);
    a.base_w = base_w;
    a.xyz = xyz;
    a.dirs = dirs;
    a.n = n;
    a.n_dev = n_device;
    a.order = order;
    a.rgb = rgb;
    a.sigma = sigma;
    a.geo = geo;
    if (n == 0) return QF_OK;
    if (!xyz) return QF_ERR_INVALID_ARGUMENT;
    hipStream_t st = qf_stream(stream);
    switch (desc->head) {
    case QF_HEAD_NONE:
        if (!sigma && !geo) return QF_ERR_INVALID_ARGUMENT;
        return launch_field_b<QF_HEAD_NONE>(a, st);
    case QF_HEAD_NGP:
        if (!head_ngp_w || !dirs || !rgb) return QF_ERR_INVALID_ARGUMENT;
        a.head_w = head_ngp_w;
        return launch_field_b<QF_HEAD_NGP>(a, st);
    case QF_HEAD_SG:
        if (!head_sg || !head_sg->w1 || !head_sg->b1 || !head_sg->w2 || !head_sg->b2 || !head_sg->wout ||
            !head_sg->bout || !dirs || !rgb)
            return QF_ERR_INVALID_ARGUMENT;
        if (desc->n_lobes < 1 || desc->n_lobes > QF_MAX_LOBES) return QF_ERR_UNSUPPORTED;
        a.sg_w1 = head_sg->w1; a.sg_b1 = head_sg->b1; a.sg_w2 = head_sg->w2; a.sg_wout = head_sg->wout;
        a.sg_b2 = head_sg->b2; a.sg_bout = head_sg->bout;
        a.n_lobes = desc->n_lobes;
        a.n_out = 3 + 7 * desc->n_lobes;
        a.nt_out = (a.n_out + 15) / 16;
        return launch_field_b<QF_HEAD_SG>(a, st);
    default:
        return QF_ERR_UNSUPPORTED;
    }
}
