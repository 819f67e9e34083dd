// Fused backward of the NGPRadianceField MLPs (training side).
//
// Replaces the backward of tcnn's FullyFusedMLP pair that torch autograd reaches when the reference trains
// NGPRadianceField (examples/radiance_fields/ngp.py:757-809 under examples/train_finetune.py:465-533):
//   enc [n,32] -> 64 (ReLU) -> 16 = [raw density | geo15];  density = exp(raw - 1) * selector
//   [SH16(dir) | geo15 | 1] -> 64 (ReLU) -> 64 (ReLU) -> 16;  rgb = sigmoid(out[:3])
// Given dL/drgb [n,3] and dL/ddensity [n], one launch recomputes the forward pass of 16 points per wave exactly as
// field_kernel does (same MFMA chain, activations in registers) and then
//   * back-propagates through the transposed weights with the SAME chained layout -- the D-layout accumulator of one
//     layer is the B operand of the next, the k-order permutation lives in the (transposed) weight images in LDS --
//     down to dL/denc [n,32], which goes to qf_grid_encode_backward;
//   * accumulates the five weight gradients dW = sum_p dz[:,p] a[:,p]^T on the matrix cores: both operands are moved
//     from the D layout (neuron quartet x point) to the A/B layout (neuron x point quartet) through a per-wave LDS
//     transpose, and the 40 16x16 tiles of dW (10 240 parameters) stay in 160 accumulator registers per lane for the
//     whole launch; at the end every wave adds its tiles to the fp32 gradient vectors with atomics.
// v_mfma_f32_16x16x4_f32 throughout (exact fp32 products, fp32 accumulate).
#include "field_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTrainBlock = 256;     // 4 waves: one per SIMD, so that a lane may use up to 512 registers
constexpr int kFwdMfma = 160;        // 48 base + 112 head, program order of field_kernel<NGP>
constexpr int kBwdMfma = 144;        // V3^T 16, V2^T 64, V1^T (geo rows) 16, W2^T 16, W1^T 32

struct TrainArgs {
    const float *enc, *dirs;
    const uint8_t *sel;
    const float *d_rgb, *d_sigma;
    const float *base_w, *head_w;
    int64_t n;
    float *d_enc, *g_base, *g_head;
};

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ int img_index(int m, int lane) { return ((((m >> 2) << 6) + lane) << 2) + (m & 3); }

// column of the head's first layer fed by register 4+r of lane quartet kq: [geo | 1] part of [SH16 | geo15 | 1]
__device__ __forceinline__ int geo_col(int o) { return o == 0 ? 31 : 15 + o; }

// A operand of forward MFMA m (identical to field_kernel<QF_HEAD_NGP>)
__device__ float fwd_weight(const TrainArgs &a, int m, int lane)
{
    const int i = lane & 15, kq = lane >> 4;
    if (m < 32) {
        const int s = m >> 2, mt = m & 3;
        return a.base_w[(16 * mt + i) * 32 + 2 * (4 * (s >> 1) + kq) + (s & 1)];
    }
    if (m < 48) return a.base_w[2048 + i * 64 + hidden_col(m - 32, kq)];
    m -= 48;
    if (m < 32) {
        const int s = m >> 2, mt = m & 3;
        const int col = s < 4 ? 4 * kq + s : geo_col(4 * kq + (s - 4));
        return a.head_w[(16 * mt + i) * 32 + col];
    }
    if (m < 96) {
        const int q = m - 32, s = q >> 2, mt = q & 3;
        return a.head_w[2048 + (16 * mt + i) * 64 + hidden_col(s, kq)];
    }
    return a.head_w[2048 + 4096 + i * 64 + hidden_col(m - 96, kq)];
}

// A operand of backward MFMA bm (a layer's transposed weights): rows = the layer's inputs, k = its outputs, in the
// register order the chained D layout delivers them
__device__ float bwd_weight(const TrainArgs &a, int bm, int lane)
{
    const int i = lane & 15, kq = lane >> 4;
    if (bm < 16) {                        // V3^T: [64 x 16], s outer (4), mt' inner (4); k = out row 4kq + s
        const int s = bm >> 2, mt = bm & 3;
        return a.head_w[2048 + 4096 + (4 * kq + s) * 64 + 16 * mt + i];
    }
    if (bm < 80) {                        // V2^T: [64 x 64], s outer (16), mt' inner (4)
        const int q = bm - 16, s = q >> 2, mt = q & 3;
        return a.head_w[2048 + hidden_col(s, kq) * 64 + 16 * mt + i];
    }
    if (bm < 96) {                        // V1^T, the 16 rows that carry [1 | geo15]: k-steps 16
        const int s = bm - 80;
        return a.head_w[hidden_col(s, kq) * 32 + geo_col(i)];
    }
    if (bm < 112) {                       // W2^T: [64 x 16], s outer (4), mt' inner (4)
        const int q = bm - 96, s = q >> 2, mt = q & 3;
        return a.base_w[2048 + (4 * kq + s) * 64 + 16 * mt + i];
    }
    const int q = bm - 112, s = q >> 1, mt = q & 1;   // W1^T: [32 x 64], s outer (16), mt' inner (2)
    return a.base_w[hidden_col(s, kq) * 32 + 16 * mt + i];
}

// D layout (lane (g,p): rows 4g..4g+3 of column p) -> A/B layout (lane (i,kq): row i, columns 4s+kq for s = 0..3),
// through a 16 x 17 float scratch private to the wave.
__device__ __forceinline__ f32x4 to_operand(const f32x4 d, volatile float *scratch, int lane)
{
    const int g = lane >> 4, p = lane & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) scratch[(4 * g + r) * 17 + p] = d[r];
    __builtin_amdgcn_wave_barrier();
    f32x4 o;
#pragma unroll
    for (int s = 0; s < 4; ++s) o[s] = scratch[p * 17 + 4 * s + g];   // row i = p, column 4s + kq, kq = g
    __builtin_amdgcn_wave_barrier();
    return o;
}

// acc += dz^T-tile x a^T-tile over the 16 points of the group (4 k-steps of 4 points)
__device__ __forceinline__ f32x4 outer_acc(const f32x4 dz_op, const f32x4 a_op, f32x4 acc)
{
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = mfma(dz_op[s], a_op[s], acc);
    return acc;
}

__global__ __launch_bounds__(kTrainBlock, 1) void ngp_mlp_backward_kernel(const TrainArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, p = lane & 15;
    for (int e = tid; e < kFwdMfma * 64; e += kTrainBlock) lds[img_index(e >> 6, e & 63)] = fwd_weight(a, e >> 6, e & 63);
    float *blds = lds + kFwdMfma * 64;
    for (int e = tid; e < kBwdMfma * 64; e += kTrainBlock) blds[img_index(e >> 6, e & 63)] = bwd_weight(a, e >> 6, e & 63);
    volatile float *scratch = blds + kBwdMfma * 64 + wave * (16 * 17);
    __syncthreads();
    const f32x4 *img = reinterpret_cast<const f32x4 *>(lds);
    const f32x4 *bimg = reinterpret_cast<const f32x4 *>(blds);

    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 aV3[4], aV2[4][4], aV1[4][2], aW2[4], aW1[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        aV3[i] = zero;
        aW2[i] = zero;
#pragma unroll
        for (int j = 0; j < 4; ++j) aV2[i][j] = zero;
        aV1[i][0] = aV1[i][1] = aW1[i][0] = aW1[i][1] = zero;
    }

    const int64_t n_groups = (a.n + 15) >> 4;
    const int64_t wave_global = (int64_t)blockIdx.x * (kTrainBlock / 64) + wave;
    const int64_t wave_stride = (int64_t)gridDim.x * (kTrainBlock / 64);
    for (int64_t grp = wave_global; grp < n_groups; grp += wave_stride) {
        const int64_t pt_raw = grp * 16 + p;
        const bool valid = pt_raw < a.n;
        const int64_t pt = valid ? pt_raw : a.n - 1;
        int loff = lane;
        asm volatile("" : "+v"(loff));              // keep the weight images in LDS (see field_kernel)
        const f32x4 *im = img + loff;
        const f32x4 *bi = bimg + loff;

        // ---------------------------------------------------------------- forward, as field_kernel<NGP>
        float feat[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) feat[s] = a.enc[pt * 32 + 2 * (4 * (s >> 1) + g) + (s & 1)];
        f32x4 h[4] = {zero, zero, zero, zero};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const f32x4 w4 = im[s * 64];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) h[mt] = mfma(w4[mt], feat[s], h[mt]);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) h[mt][r] = fmaxf(h[mt][r], 0.0f);
        f32x4 oa = zero, ob = zero;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 w4 = im[(8 + q) * 64];
            oa = mfma(w4[0], h[q][0], oa);
            ob = mfma(w4[1], h[q][1], ob);
            oa = mfma(w4[2], h[q][2], oa);
            ob = mfma(w4[3], h[q][3], ob);
        }
        const f32x4 base_out = oa + ob;
        const float density = a.sel[pt] ? expf(base_out[0] - 1.0f) : 0.0f;
        const float dx = a.dirs[pt * 3 + 0], dy = a.dirs[pt * 3 + 1], dzv = a.dirs[pt * 3 + 2];
        const float ux = ((dx + 1.0f) / 2.0f) * 2.0f - 1.0f, uy = ((dy + 1.0f) / 2.0f) * 2.0f - 1.0f,
                    uz = ((dzv + 1.0f) / 2.0f) * 2.0f - 1.0f;
        float in[8];
        sh4_quartet(g, ux, uy, uz, in);
#pragma unroll
        for (int r = 0; r < 4; ++r) in[4 + r] = base_out[r];
        if (g == 0) in[4] = 1.0f;
        f32x4 h1[4] = {zero, zero, zero, zero};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const f32x4 w4 = im[(12 + s) * 64];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) h1[mt] = mfma(w4[mt], in[s], h1[mt]);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) h1[mt][r] = fmaxf(h1[mt][r], 0.0f);
        f32x4 h2[4] = {zero, zero, zero, zero};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const f32x4 w4 = im[(20 + s) * 64];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) h2[mt] = mfma(w4[mt], h1[s >> 2][s & 3], h2[mt]);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) h2[mt][r] = fmaxf(h2[mt][r], 0.0f);
        f32x4 ca = zero, cb = zero;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 w4 = im[(36 + q) * 64];
            ca = mfma(w4[0], h2[q][0], ca);
            cb = mfma(w4[1], h2[q][1], cb);
            ca = mfma(w4[2], h2[q][2], ca);
            cb = mfma(w4[3], h2[q][3], cb);
        }
        const f32x4 c = ca + cb;

        // ---------------------------------------------------------------- backward through the head
        f32x4 dz3 = zero;                      // rows 0..2 of the 16-row output tile live in lane quartet 0
        if (g == 0 && valid) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float sg = sigmoidf(c[r]);
                dz3[r] = a.d_rgb[pt * 3 + r] * sg * (1.0f - sg);
            }
        }
        f32x4 dz2[4] = {zero, zero, zero, zero};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4 w4 = bi[s * 64];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) dz2[mt] = mfma(w4[mt], dz3[s], dz2[mt]);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dz2[mt][r] = h2[mt][r] > 0.0f ? dz2[mt][r] : 0.0f;
        f32x4 dz1[4] = {zero, zero, zero, zero};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const f32x4 w4 = bi[(4 + s) * 64];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) dz1[mt] = mfma(w4[mt], dz2[s >> 2][s & 3], dz1[mt]);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dz1[mt][r] = h1[mt][r] > 0.0f ? dz1[mt][r] : 0.0f;
        f32x4 da = zero, db = zero;            // d[1 | geo15]: lane (g,p) register r = d out16 row 4g + r (row 0 unused)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 w4 = bi[(20 + q) * 64];
            da = mfma(w4[0], dz1[q][0], da);
            db = mfma(w4[1], dz1[q][1], db);
            da = mfma(w4[2], dz1[q][2], da);
            db = mfma(w4[3], dz1[q][3], db);
        }
        f32x4 dout = da + db;
        if (g == 0) dout[0] = valid ? a.d_sigma[pt] * density : 0.0f;   // d exp(raw - 1) * selector
        if (!valid) dout = zero;

        // ---------------------------------------------------------------- head weight gradients
        {
            const f32x4 t3 = to_operand(dz3, scratch, lane);
            f32x4 th[4];
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) th[ti] = to_operand(h2[ti], scratch, lane);
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) aV3[ti] = outer_acc(t3, th[ti], aV3[ti]);
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) th[ti] = to_operand(h1[ti], scratch, lane);
#pragma unroll
            for (int to = 0; to < 4; ++to) {
                const f32x4 tz = to_operand(dz2[to], scratch, lane);
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) aV2[to][ti] = outer_acc(tz, th[ti], aV2[to][ti]);
            }
            f32x4 tin[2];
            tin[0] = to_operand((f32x4){in[0], in[1], in[2], in[3]}, scratch, lane);
            tin[1] = to_operand((f32x4){in[4], in[5], in[6], in[7]}, scratch, lane);
#pragma unroll
            for (int to = 0; to < 4; ++to) {
                const f32x4 tz = to_operand(dz1[to], scratch, lane);
                aV1[to][0] = outer_acc(tz, tin[0], aV1[to][0]);
                aV1[to][1] = outer_acc(tz, tin[1], aV1[to][1]);
            }
        }

        // ---------------------------------------------------------------- backward through the base MLP
        f32x4 dzh[4] = {zero, zero, zero, zero};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4 w4 = bi[(24 + s) * 64];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) dzh[mt] = mfma(w4[mt], dout[s], dzh[mt]);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dzh[mt][r] = h[mt][r] > 0.0f ? dzh[mt][r] : 0.0f;
        f32x4 de[2] = {zero, zero};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const f32x4 w4 = bi[(28 + (s >> 1)) * 64];
            de[0] = mfma(w4[2 * (s & 1) + 0], dzh[s >> 2][s & 3], de[0]);
            de[1] = mfma(w4[2 * (s & 1) + 1], dzh[s >> 2][s & 3], de[1]);
        }
        if (valid) {
            *reinterpret_cast<f32x4 *>(a.d_enc + pt * 32 + 4 * g) = de[0];
            *reinterpret_cast<f32x4 *>(a.d_enc + pt * 32 + 16 + 4 * g) = de[1];
        }
        {
            const f32x4 to_ = to_operand(dout, scratch, lane);
            f32x4 th[4];
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) th[ti] = to_operand(h[ti], scratch, lane);
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) aW2[ti] = outer_acc(to_, th[ti], aW2[ti]);
            f32x4 tf[2];
            tf[0] = to_operand((f32x4){feat[0], feat[1], feat[2], feat[3]}, scratch, lane);
            tf[1] = to_operand((f32x4){feat[4], feat[5], feat[6], feat[7]}, scratch, lane);
#pragma unroll
            for (int to = 0; to < 4; ++to) {
                const f32x4 tz = to_operand(dzh[to], scratch, lane);
                aW1[to][0] = outer_acc(tz, tf[0], aW1[to][0]);
                aW1[to][1] = outer_acc(tz, tf[1], aW1[to][1]);
            }
        }
    }

    // ---- add this wave's 40 tiles to the gradient vectors: lane (g,p) register r = dW[16 to + 4g + r][column(ti, p)]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r;
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            atomicAdd(a.g_head + 2048 + 4096 + row * 64 + 16 * ti + p, aV3[ti][r]);
            atomicAdd(a.g_base + 2048 + row * 64 + 16 * ti + p, aW2[ti][r]);
#pragma unroll
            for (int to = 0; to < 4; ++to) atomicAdd(a.g_head + 2048 + (16 * to + row) * 64 + 16 * ti + p, aV2[to][ti][r]);
        }
#pragma unroll
        for (int to = 0; to < 4; ++to) {
            atomicAdd(a.g_head + (16 * to + row) * 32 + p, aV1[to][0][r]);
            atomicAdd(a.g_head + (16 * to + row) * 32 + geo_col(p), aV1[to][1][r]);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int s = 4 * t + (p & 3);
                atomicAdd(a.g_base + (16 * to + row) * 32 + 2 * (4 * (s >> 1) + (p >> 2)) + (s & 1), aW1[to][t][r]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// The same for NGPRadianceFieldSGNew (ngp.py:404-470): base MLP + BasicDecoder 15 -> 64 -> 64 -> (3+7L) with biases.
// The spherical-Gaussian mixture is differentiated per point by sg_features_to_rgb_backward_kernel; this kernel takes
// dL/dfeatures [n, 3+7L] and dL/ddensity.  Biases: b1 rides in the constant-1 slot of the first layer's input (as in
// field_kernel), so its gradient is column 0 of that layer's weight-gradient tile; b2 / bout get theirs from one more
// tile per output tile whose "activation" is a row of ones.
constexpr int kSgFwdMfma = 128;      // 48 base + 16 (16 -> 64) + 64 (64 -> 64); the output layer is not recomputed
constexpr int kSgBwdMfma = 192;      // Wout^T 64, W2^T 64, W1^T 16, base W2^T 16, base W1^T 32

struct SgTrainArgs {
    const float *enc;
    const uint8_t *sel;
    const float *d_feat;
    int64_t d_stride;
    const float *d_sigma;
    const float *base_w;
    qf_sg_head sg;
    int32_t n_out, nt_out;
    int64_t n;
    float *d_enc, *g_base, *g_w1, *g_b1, *g_w2, *g_b2, *g_wout, *g_bout;
};

__device__ float sg_fwd_weight(const SgTrainArgs &a, int m, int lane)
{
    const int i = lane & 15, kq = lane >> 4;
    if (m < 32) {
        const int s = m >> 2, mt = m & 3;
        return a.base_w[(16 * mt + i) * 32 + 2 * (4 * (s >> 1) + kq) + (s & 1)];
    }
    if (m < 48) return a.base_w[2048 + i * 64 + hidden_col(m - 32, kq)];
    m -= 48;
    if (m < 16) {
        const int s = m >> 2, mt = m & 3, row = 16 * mt + i, o = 4 * kq + s;
        return o == 0 ? a.sg.b1[row] : a.sg.w1[row * 15 + (o - 1)];
    }
    const int q = m - 16, s = q >> 2, mt = q & 3;
    return a.sg.w2[(16 * mt + i) * 64 + hidden_col(s, kq)];
}

__device__ float sg_bwd_weight(const SgTrainArgs &a, int bm, int lane)
{
    const int i = lane & 15, kq = lane >> 4;
    if (bm < 64) {                        // Wout^T: output tile t outer (4), k-step s (4), mt' inner (4)
        const int t = bm >> 4, s = (bm >> 2) & 3, mt = bm & 3, row = 16 * t + 4 * kq + s;
        return row < a.n_out ? a.sg.wout[row * 64 + 16 * mt + i] : 0.0f;
    }
    if (bm < 128) {                       // W2^T
        const int q = bm - 64, s = q >> 2, mt = q & 3;
        return a.sg.w2[hidden_col(s, kq) * 64 + 16 * mt + i];
    }
    if (bm < 144) {                       // W1^T: row i = input slot i ([1 | geo15]); the constant's row is not needed
        const int s = bm - 128;
        return i == 0 ? 0.0f : a.sg.w1[hidden_col(s, kq) * 15 + (i - 1)];
    }
    if (bm < 160) {
        const int q = bm - 144, s = q >> 2, mt = q & 3;
        return a.base_w[2048 + (4 * kq + s) * 64 + 16 * mt + i];
    }
    const int q = bm - 160, s = q >> 1, mt = q & 1;
    return a.base_w[hidden_col(s, kq) * 32 + 16 * mt + i];
}

__global__ __launch_bounds__(kTrainBlock, 1) void sg_mlp_backward_kernel(const SgTrainArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, p = lane & 15;
    for (int e = tid; e < kSgFwdMfma * 64; e += kTrainBlock) lds[img_index(e >> 6, e & 63)] = sg_fwd_weight(a, e >> 6, e & 63);
    float *blds = lds + kSgFwdMfma * 64;
    for (int e = tid; e < kSgBwdMfma * 64; e += kTrainBlock) blds[img_index(e >> 6, e & 63)] = sg_bwd_weight(a, e >> 6, e & 63);
    float *b2_lds = blds + kSgBwdMfma * 64;
    if (tid < 64) b2_lds[tid] = a.sg.b2[tid];
    volatile float *scratch = b2_lds + 64 + wave * (16 * 17);
    __syncthreads();
    const f32x4 *img = reinterpret_cast<const f32x4 *>(lds);
    const f32x4 *bimg = reinterpret_cast<const f32x4 *>(blds);
    const f32x4 *b2v = reinterpret_cast<const f32x4 *>(b2_lds);

    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 ones_op = p == 0 ? (f32x4){1.f, 1.f, 1.f, 1.f} : zero;     // operand form of "row 0 = 1 for every point"
    f32x4 aWo[4][4], abo[4], aW2h[4][4], ab2[4], aW1h[4], aW2[4], aW1[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        abo[i] = ab2[i] = aW1h[i] = aW2[i] = zero;
        aW1[i][0] = aW1[i][1] = zero;
#pragma unroll
        for (int j = 0; j < 4; ++j) aWo[i][j] = aW2h[i][j] = zero;
    }

    const int64_t n_groups = (a.n + 15) >> 4;
    const int64_t wave_global = (int64_t)blockIdx.x * (kTrainBlock / 64) + wave;
    const int64_t wave_stride = (int64_t)gridDim.x * (kTrainBlock / 64);
    for (int64_t grp = wave_global; grp < n_groups; grp += wave_stride) {
        const int64_t pt_raw = grp * 16 + p;
        const bool valid = pt_raw < a.n;
        const int64_t pt = valid ? pt_raw : a.n - 1;
        int loff = lane, goff = g;
        asm volatile("" : "+v"(loff), "+v"(goff));
        const f32x4 *im = img + loff;
        const f32x4 *bi = bimg + loff;

        // ---------------------------------------------------------------- forward up to the second hidden layer
        float feat[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) feat[s] = a.enc[pt * 32 + 2 * (4 * (s >> 1) + g) + (s & 1)];
        f32x4 h[4] = {zero, zero, zero, zero};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const f32x4 w4 = im[s * 64];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) h[mt] = mfma(w4[mt], feat[s], h[mt]);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) h[mt][r] = fmaxf(h[mt][r], 0.0f);
        f32x4 oa = zero, ob = zero;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 w4 = im[(8 + q) * 64];
            oa = mfma(w4[0], h[q][0], oa);
            ob = mfma(w4[1], h[q][1], ob);
            oa = mfma(w4[2], h[q][2], oa);
            ob = mfma(w4[3], h[q][3], ob);
        }
        const f32x4 base_out = oa + ob;
        const float density = a.sel[pt] ? expf(base_out[0] - 1.0f) : 0.0f;
        f32x4 in = base_out;
        if (g == 0) in[0] = 1.0f;                  // bias slot
        f32x4 h1[4] = {zero, zero, zero, zero};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4 w4 = im[(12 + s) * 64];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) h1[mt] = mfma(w4[mt], in[s], h1[mt]);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) h1[mt][r] = fmaxf(h1[mt][r], 0.0f);
        f32x4 h2[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) h2[mt] = b2v[4 * mt + goff];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const f32x4 w4 = im[(16 + s) * 64];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) h2[mt] = mfma(w4[mt], h1[s >> 2][s & 3], h2[mt]);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) h2[mt][r] = fmaxf(h2[mt][r], 0.0f);

        // ---------------------------------------------------------------- backward through the decoder
        f32x4 dzo[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            dzo[t] = zero;
            if (t < a.nt_out && valid) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = 16 * t + 4 * g + r;
                    if (o < a.n_out) dzo[t][r] = a.d_feat[pt * a.d_stride + o];
                }
            }
        }
        f32x4 dz2[4] = {zero, zero, zero, zero};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < a.nt_out) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const f32x4 w4 = bi[(4 * t + s) * 64];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) dz2[mt] = mfma(w4[mt], dzo[t][s], dz2[mt]);
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dz2[mt][r] = h2[mt][r] > 0.0f ? dz2[mt][r] : 0.0f;
        f32x4 dz1[4] = {zero, zero, zero, zero};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const f32x4 w4 = bi[(16 + s) * 64];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) dz1[mt] = mfma(w4[mt], dz2[s >> 2][s & 3], dz1[mt]);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dz1[mt][r] = h1[mt][r] > 0.0f ? dz1[mt][r] : 0.0f;
        f32x4 da = zero, db = zero;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 w4 = bi[(32 + q) * 64];
            da = mfma(w4[0], dz1[q][0], da);
            db = mfma(w4[1], dz1[q][1], db);
            da = mfma(w4[2], dz1[q][2], da);
            db = mfma(w4[3], dz1[q][3], db);
        }
        f32x4 dout = da + db;                      // d out16 rows 4g + r (row 0: see below)
        if (g == 0) dout[0] = valid ? a.d_sigma[pt] * density : 0.0f;
        if (!valid) dout = zero;

        // ---------------------------------------------------------------- decoder weight / bias gradients
        {
            f32x4 th[4];
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) th[ti] = to_operand(h2[ti], scratch, lane);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < a.nt_out) {
                    const f32x4 tz = to_operand(dzo[t], scratch, lane);
#pragma unroll
                    for (int ti = 0; ti < 4; ++ti) aWo[t][ti] = outer_acc(tz, th[ti], aWo[t][ti]);
                    abo[t] = outer_acc(tz, ones_op, abo[t]);
                }
            }
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) th[ti] = to_operand(h1[ti], scratch, lane);
#pragma unroll
            for (int to = 0; to < 4; ++to) {
                const f32x4 tz = to_operand(dz2[to], scratch, lane);
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) aW2h[to][ti] = outer_acc(tz, th[ti], aW2h[to][ti]);
                ab2[to] = outer_acc(tz, ones_op, ab2[to]);
            }
            const f32x4 tin = to_operand(in, scratch, lane);
#pragma unroll
            for (int to = 0; to < 4; ++to) aW1h[to] = outer_acc(to_operand(dz1[to], scratch, lane), tin, aW1h[to]);
        }

        // ---------------------------------------------------------------- base MLP
        f32x4 dzh[4] = {zero, zero, zero, zero};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4 w4 = bi[(36 + s) * 64];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) dzh[mt] = mfma(w4[mt], dout[s], dzh[mt]);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dzh[mt][r] = h[mt][r] > 0.0f ? dzh[mt][r] : 0.0f;
        f32x4 de[2] = {zero, zero};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const f32x4 w4 = bi[(40 + (s >> 1)) * 64];
            de[0] = mfma(w4[2 * (s & 1) + 0], dzh[s >> 2][s & 3], de[0]);
            de[1] = mfma(w4[2 * (s & 1) + 1], dzh[s >> 2][s & 3], de[1]);
        }
        if (valid) {
            *reinterpret_cast<f32x4 *>(a.d_enc + pt * 32 + 4 * g) = de[0];
            *reinterpret_cast<f32x4 *>(a.d_enc + pt * 32 + 16 + 4 * g) = de[1];
        }
        {
            const f32x4 to_ = to_operand(dout, scratch, lane);
            f32x4 th[4];
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) th[ti] = to_operand(h[ti], scratch, lane);
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) aW2[ti] = outer_acc(to_, th[ti], aW2[ti]);
            f32x4 tf[2];
            tf[0] = to_operand((f32x4){feat[0], feat[1], feat[2], feat[3]}, scratch, lane);
            tf[1] = to_operand((f32x4){feat[4], feat[5], feat[6], feat[7]}, scratch, lane);
#pragma unroll
            for (int to = 0; to < 4; ++to) {
                const f32x4 tz = to_operand(dzh[to], scratch, lane);
                aW1[to][0] = outer_acc(tz, tf[0], aW1[to][0]);
                aW1[to][1] = outer_acc(tz, tf[1], aW1[to][1]);
            }
        }
    }

    // ---- this wave's tiles -> gradient vectors: lane (g,p) register r = d[16 tile + 4g + r][column p of the in-tile]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int orow = 16 * t + row;
            if (t < a.nt_out && orow < a.n_out) {
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) atomicAdd(a.g_wout + orow * 64 + 16 * ti + p, aWo[t][ti][r]);
                if (p == 0) atomicAdd(a.g_bout + orow, abo[t][r]);
            }
            const int hrow = 16 * t + row;          // t doubles as the tile of a 64-wide layer below
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) atomicAdd(a.g_w2 + hrow * 64 + 16 * ti + p, aW2h[t][ti][r]);
            if (p == 0) {
                atomicAdd(a.g_b2 + hrow, ab2[t][r]);
                atomicAdd(a.g_b1 + hrow, aW1h[t][r]);
            } else {
                atomicAdd(a.g_w1 + hrow * 15 + (p - 1), aW1h[t][r]);
            }
            atomicAdd(a.g_base + 2048 + row * 64 + 16 * t + p, aW2[t][r]);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int s = 4 * tt + (p & 3);
                atomicAdd(a.g_base + hrow * 32 + 2 * (4 * (s >> 1) + (p >> 2)) + (s & 1), aW1[t][tt][r]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// The deformation field's decoder (examples/field.py:186-203): cat[x01(3), grid(32)] -> 32 -> 32 -> 1, ReLU, biases.
// Same scheme; the forward chain is deform_kernel's (field_eval.hip).  dL/dout [n] -> dL/denc [n,32] (and, optionally,
// the part of dL/dx01 that flows through the three x01 inputs of the first layer), weight and bias gradients.
constexpr int kDfFwdMfma = 42, kDfBwdMfma = 42;

struct DeformTrainArgs {
    const float *enc, *x01, *d_out;
    const float *w1, *b1, *w2, *b2, *wout;
    int64_t n;
    float *d_enc, *d_x01;
    float *g_w1, *g_b1, *g_w2, *g_b2, *g_wout, *g_bout;
};

__device__ float df_fwd_weight(const DeformTrainArgs &a, int m, int lane)
{
    const int i = lane & 15, kq = lane >> 4;
    if (m < 18) {
        const int s = m >> 1, mt = m & 1, row = 16 * mt + i;
        if (s < 8) return a.w1[row * 35 + 3 + 2 * (4 * (s >> 1) + kq) + (s & 1)];
        return kq < 3 ? a.w1[row * 35 + kq] : a.b1[row];
    }
    if (m < 34) {
        const int q = m - 18, s = q >> 1, mt = q & 1;
        return a.w2[(16 * mt + i) * 32 + hidden_col(s, kq)];
    }
    return 0.0f;                              // the output layer is not recomputed
}

__device__ float df_bwd_weight(const DeformTrainArgs &a, int bm, int lane)
{
    const int i = lane & 15, kq = lane >> 4;
    if (bm < 2) return kq == 0 ? a.wout[16 * bm + i] : 0.0f;                    // Wout^T, the one k-step that carries row 0
    if (bm < 18) {                                                             // W2^T
        const int q = bm - 2, s = q >> 1, mt = q & 1;
        return a.w2[hidden_col(s, kq) * 32 + 16 * mt + i];
    }
    if (bm < 34) {                                                             // W1^T, the 32 grid columns
        const int q = bm - 18, s = q >> 1, mt = q & 1;
        return a.w1[hidden_col(s, kq) * 35 + 3 + 16 * mt + i];
    }
    const int s = bm - 34;                                                     // W1^T, the 3 x01 columns
    return i < 3 ? a.w1[hidden_col(s, kq) * 35 + i] : 0.0f;
}

__global__ __launch_bounds__(kTrainBlock, 1) void deform_mlp_backward_kernel(const DeformTrainArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, p = lane & 15;
    for (int e = tid; e < kDfFwdMfma * 64; e += kTrainBlock) lds[e] = df_fwd_weight(a, e >> 6, e & 63);
    float *blds = lds + kDfFwdMfma * 64;
    for (int e = tid; e < kDfBwdMfma * 64; e += kTrainBlock) blds[e] = df_bwd_weight(a, e >> 6, e & 63);
    float *bias = blds + kDfBwdMfma * 64;
    if (tid < 32) bias[tid] = a.b2[tid];
    volatile float *scratch = bias + 32 + wave * (16 * 17);
    __syncthreads();

    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 ones_op = p == 0 ? (f32x4){1.f, 1.f, 1.f, 1.f} : zero;
    f32x4 aWo[2] = {zero, zero}, abo = zero, aW2[2][2] = {{zero, zero}, {zero, zero}}, ab2[2] = {zero, zero};
    f32x4 aW1f[2][2] = {{zero, zero}, {zero, zero}}, aW1x[2] = {zero, zero};

    const int64_t n_groups = (a.n + 15) >> 4;
    const int64_t wave_global = (int64_t)blockIdx.x * (kTrainBlock / 64) + wave;
    const int64_t wave_stride = (int64_t)gridDim.x * (kTrainBlock / 64);
    for (int64_t grp = wave_global; grp < n_groups; grp += wave_stride) {
        const int64_t pt_raw = grp * 16 + p;
        const bool valid = pt_raw < a.n;
        const int64_t pt = valid ? pt_raw : a.n - 1;
        int loff = lane;
        asm volatile("" : "+v"(loff));
        const float *wl = lds + loff;
        const float *bl = blds + loff;

        float in[9];
#pragma unroll
        for (int s = 0; s < 8; ++s) in[s] = a.enc[pt * 32 + 2 * (4 * (s >> 1) + g) + (s & 1)];
        in[8] = g < 3 ? a.x01[pt * 3 + g] : 1.0f;
        f32x4 h1[2] = {zero, zero};
#pragma unroll
        for (int s = 0; s < 9; ++s)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) h1[mt] = mfma(wl[(2 * s + mt) * 64], in[s], h1[mt]);
        f32x4 h2[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { h1[mt][r] = fmaxf(h1[mt][r], 0.0f); h2[mt][r] = bias[16 * mt + 4 * g + r]; }
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) h2[mt] = mfma(wl[(18 + 2 * s + mt) * 64], h1[s >> 2][s & 3], h2[mt]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) h2[mt][r] = fmaxf(h2[mt][r], 0.0f);

        // ---- backward
        const float v = (g == 0 && valid) ? a.d_out[pt] : 0.0f;       // row 0 of the 16-row output tile
        f32x4 dz2[2], dz1[2] = {zero, zero}, de[2] = {zero, zero}, dxt = zero;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            dz2[mt] = mfma(bl[mt * 64], v, zero);
#pragma unroll
            for (int r = 0; r < 4; ++r) dz2[mt][r] = h2[mt][r] > 0.0f ? dz2[mt][r] : 0.0f;
        }
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) dz1[mt] = mfma(bl[(2 + 2 * s + mt) * 64], dz2[s >> 2][s & 3], dz1[mt]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dz1[mt][r] = h1[mt][r] > 0.0f ? dz1[mt][r] : 0.0f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) de[mt] = mfma(bl[(18 + 2 * s + mt) * 64], dz1[s >> 2][s & 3], de[mt]);
            dxt = mfma(bl[(34 + s) * 64], dz1[s >> 2][s & 3], dxt);
        }
        if (valid) {
            *reinterpret_cast<f32x4 *>(a.d_enc + pt * 32 + 4 * g) = de[0];
            *reinterpret_cast<f32x4 *>(a.d_enc + pt * 32 + 16 + 4 * g) = de[1];
            if (a.d_x01 && g == 0) {
                a.d_x01[pt * 3 + 0] = dxt[0];
                a.d_x01[pt * 3 + 1] = dxt[1];
                a.d_x01[pt * 3 + 2] = dxt[2];
            }
        }

        // ---- weight / bias gradients
        {
            const f32x4 t3 = to_operand((f32x4){v, 0.f, 0.f, 0.f}, scratch, lane);
            f32x4 th[2];
            th[0] = to_operand(h2[0], scratch, lane);
            th[1] = to_operand(h2[1], scratch, lane);
            aWo[0] = outer_acc(t3, th[0], aWo[0]);
            aWo[1] = outer_acc(t3, th[1], aWo[1]);
            abo = outer_acc(t3, ones_op, abo);
            th[0] = to_operand(h1[0], scratch, lane);
            th[1] = to_operand(h1[1], scratch, lane);
            f32x4 tin[3];
            tin[0] = to_operand((f32x4){in[0], in[1], in[2], in[3]}, scratch, lane);
            tin[1] = to_operand((f32x4){in[4], in[5], in[6], in[7]}, scratch, lane);
            tin[2] = to_operand((f32x4){in[8], 0.f, 0.f, 0.f}, scratch, lane);
#pragma unroll
            for (int to = 0; to < 2; ++to) {
                const f32x4 tz2 = to_operand(dz2[to], scratch, lane);
                aW2[to][0] = outer_acc(tz2, th[0], aW2[to][0]);
                aW2[to][1] = outer_acc(tz2, th[1], aW2[to][1]);
                ab2[to] = outer_acc(tz2, ones_op, ab2[to]);
                const f32x4 tz1 = to_operand(dz1[to], scratch, lane);
                aW1f[to][0] = outer_acc(tz1, tin[0], aW1f[to][0]);
                aW1f[to][1] = outer_acc(tz1, tin[1], aW1f[to][1]);
                aW1x[to] = outer_acc(tz1, tin[2], aW1x[to]);
            }
        }
    }

#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r;
        if (row == 0) {
            atomicAdd(a.g_wout + p, aWo[0][r]);
            atomicAdd(a.g_wout + 16 + p, aWo[1][r]);
            if (p == 0) atomicAdd(a.g_bout, abo[r]);
        }
#pragma unroll
        for (int to = 0; to < 2; ++to) {
            const int hrow = 16 * to + row;
            atomicAdd(a.g_w2 + hrow * 32 + p, aW2[to][0][r]);
            atomicAdd(a.g_w2 + hrow * 32 + 16 + p, aW2[to][1][r]);
            if (p == 0) atomicAdd(a.g_b2 + hrow, ab2[to][r]);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int s = 4 * t + (p & 3);
                atomicAdd(a.g_w1 + hrow * 35 + 3 + 2 * (4 * (s >> 1) + (p >> 2)) + (s & 1), aW1f[to][t][r]);
            }
            if ((p & 3) == 0) {
                if (p < 12) atomicAdd(a.g_w1 + hrow * 35 + (p >> 2), aW1x[to][r]);
                else atomicAdd(a.g_b1 + hrow, aW1x[to][r]);
            }
        }
    }
}

// Backward of features_to_rgb (ngp.py:371-393,456-461): rgb = sigmoid(diffuse + sum_l c_l exp(|lambda_l| (a_l/|a_l| . d - 1))).
// One lane per point; replaces ~20 elementwise torch kernels per lobe in the SG-fitting step (train_fit_sg.py:439-461).
__global__ void sg_features_to_rgb_backward_kernel(const float *features, int64_t stride, const float *dirs,
                                                   const float *d_rgb, int64_t n, int n_lobes, float *d_features,
                                                   int64_t d_stride)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float *f = features + i * stride;
        float *df = d_features + i * d_stride;
        const float dx = dirs[i * 3], dy = dirs[i * 3 + 1], dz = dirs[i * 3 + 2];
        float pre[3] = {f[0], f[1], f[2]};
        for (int l = 0; l < n_lobes; ++l) {
            const float *x = f + 3 + 7 * l;
            const float nrm = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
            const float dotp = (x[0] / nrm) * dx + (x[1] / nrm) * dy + (x[2] / nrm) * dz;
            const float e = expf(fabsf(x[3]) * (dotp - 1.0f));
            pre[0] += x[4] * e;
            pre[1] += x[5] * e;
            pre[2] += x[6] * e;
        }
        float g[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float sg = sigmoidf(pre[k]);
            g[k] = d_rgb[i * 3 + k] * sg * (1.0f - sg);
            df[k] = g[k];
        }
        for (int l = 0; l < n_lobes; ++l) {
            const float *x = f + 3 + 7 * l;
            float *o = df + 3 + 7 * l;
            const float nrm = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
            const float ax = x[0] / nrm, ay = x[1] / nrm, az = x[2] / nrm;
            const float t = (ax * dx + ay * dy + az * dz) - 1.0f;
            const float sh = fabsf(x[3]);
            const float e = expf(sh * t);
            const float dE = g[0] * x[4] + g[1] * x[5] + g[2] * x[6];
            o[4] = g[0] * e;
            o[5] = g[1] * e;
            o[6] = g[2] * e;
            const float dEe = dE * e;
            o[3] = dEe * t * (x[3] > 0.0f ? 1.0f : (x[3] < 0.0f ? -1.0f : 0.0f));      // d|lambda|: 0 at 0, as torch.abs
            const float dt = dEe * sh;                                                  // d(a_hat . d)
            const float proj = ax * dx + ay * dy + az * dz;                             // d a_hat = dt * d; d a = (I - a_hat a_hat^T) d a_hat / |a|
            o[0] = dt * (dx - ax * proj) / nrm;
            o[1] = dt * (dy - ay * proj) / nrm;
            o[2] = dt * (dz - az * proj) / nrm;
        }
    }
}

}  // namespace

extern "C" int qf_ngp_mlp_backward(const float *enc, const float *dirs, const uint8_t *selector, const float *d_rgb,
                                   const float *d_sigma, const float *base_w, const float *head_w, int64_t n,
                                   float *d_enc, float *grad_base_w, float *grad_head_w, void *stream)
{
    if (n < 0) return QF_ERR_INVALID_ARGUMENT;
    if (n == 0) return QF_OK;
    if (!enc || !dirs || !selector || !d_rgb || !d_sigma || !base_w || !head_w || !d_enc || !grad_base_w || !grad_head_w)
        return QF_ERR_INVALID_ARGUMENT;
    TrainArgs a;
    a.enc = enc; a.dirs = dirs; a.sel = selector; a.d_rgb = d_rgb; a.d_sigma = d_sigma;
    a.base_w = base_w; a.head_w = head_w; a.n = n; a.d_enc = d_enc; a.g_base = grad_base_w; a.g_head = grad_head_w;
    const size_t lds_bytes = (size_t)((kFwdMfma + kBwdMfma) * 64 + (kTrainBlock / 64) * 16 * 17) * sizeof(float);
    static QfLdsAttr attr;                           // per device
    QF_HIP_TRY(qf_ensure_dynamic_lds(attr, reinterpret_cast<const void *>(ngp_mlp_backward_kernel), lds_bytes));
    const int64_t n_groups = (n + 15) / 16;
    int64_t blocks = qf_div_up(n_groups, kTrainBlock / 64);
    const int64_t cap = (int64_t)qf_cu_count_cached();
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(ngp_mlp_backward_kernel, dim3((unsigned)blocks), dim3(kTrainBlock), lds_bytes, qf_stream(stream), a);
    QF_LAUNCH_CHECK();
    return QF_OK;
}

extern "C" int qf_sg_features_to_rgb_backward(const float *features, int64_t feat_stride, const float *dirs,
                                              const float *d_rgb, int64_t n, int32_t n_lobes, float *d_features,
                                              int64_t d_stride, void *stream)
{
    if (n < 0 || n_lobes < 1 || n_lobes > QF_MAX_LOBES || feat_stride < 3 + 7 * n_lobes || d_stride < 3 + 7 * n_lobes)
        return QF_ERR_INVALID_ARGUMENT;
    if (n == 0) return QF_OK;
    if (!features || !dirs || !d_rgb || !d_features) return QF_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(sg_features_to_rgb_backward_kernel, dim3(qf_grid_1d(n, 256)), dim3(256), 0, qf_stream(stream),
                       features, feat_stride, dirs, d_rgb, n, (int)n_lobes, d_features, d_stride);
    QF_LAUNCH_CHECK();
    return QF_OK;
}

extern "C" int qf_sg_mlp_backward(const float *enc, const uint8_t *selector, const float *d_features, int64_t d_stride,
                                  const float *d_sigma, const float *base_w, const qf_sg_head *head, int32_t n_lobes,
                                  int64_t n, float *d_enc, float *grad_base_w, const qf_sg_head *grad_head, void *stream)
{
    if (n < 0 || n_lobes < 1 || n_lobes > QF_MAX_LOBES || d_stride < 3 + 7 * n_lobes) return QF_ERR_INVALID_ARGUMENT;
    if (n == 0) return QF_OK;
    if (!enc || !selector || !d_features || !d_sigma || !base_w || !head || !d_enc || !grad_base_w || !grad_head)
        return QF_ERR_INVALID_ARGUMENT;
    if (!head->w1 || !head->b1 || !head->w2 || !head->b2 || !head->wout || !head->bout || !grad_head->w1 || !grad_head->b1 ||
        !grad_head->w2 || !grad_head->b2 || !grad_head->wout || !grad_head->bout)
        return QF_ERR_INVALID_ARGUMENT;
    SgTrainArgs a;
    a.enc = enc; a.sel = selector; a.d_feat = d_features; a.d_stride = d_stride; a.d_sigma = d_sigma; a.base_w = base_w;
    a.sg = *head;
    a.n_out = 3 + 7 * n_lobes;
    a.nt_out = (a.n_out + 15) / 16;
    a.n = n;
    a.d_enc = d_enc; a.g_base = grad_base_w;
    a.g_w1 = const_cast<float *>(grad_head->w1); a.g_b1 = const_cast<float *>(grad_head->b1);
    a.g_w2 = const_cast<float *>(grad_head->w2); a.g_b2 = const_cast<float *>(grad_head->b2);
    a.g_wout = const_cast<float *>(grad_head->wout); a.g_bout = const_cast<float *>(grad_head->bout);
    const size_t lds_bytes = (size_t)((kSgFwdMfma + kSgBwdMfma) * 64 + 64 + (kTrainBlock / 64) * 16 * 17) * sizeof(float);
    static QfLdsAttr attr;                           // per device
    QF_HIP_TRY(qf_ensure_dynamic_lds(attr, reinterpret_cast<const void *>(sg_mlp_backward_kernel), lds_bytes));
    const int64_t n_groups = (n + 15) / 16;
    int64_t blocks = qf_div_up(n_groups, kTrainBlock / 64);
    const int64_t cap = (int64_t)qf_cu_count_cached();
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(sg_mlp_backward_kernel, dim3((unsigned)blocks), dim3(kTrainBlock), lds_bytes, qf_stream(stream), a);
    QF_LAUNCH_CHECK();
    return QF_OK;
}

extern "C" int qf_deform_mlp_backward(const float *enc, const float *x01, const float *d_out, const float *w1, const float *b1,
                                      const float *w2, const float *b2, const float *wout, int64_t n, float *d_enc,
                                      float *d_x01, float *g_w1, float *g_b1, float *g_w2, float *g_b2, float *g_wout,
                                      float *g_bout, void *stream)
{
    if (n < 0) return QF_ERR_INVALID_ARGUMENT;
    if (n == 0) return QF_OK;
    if (!enc || !x01 || !d_out || !w1 || !b1 || !w2 || !b2 || !wout || !d_enc || !g_w1 || !g_b1 || !g_w2 || !g_b2 || !g_wout ||
        !g_bout)
        return QF_ERR_INVALID_ARGUMENT;
    DeformTrainArgs a;
    a.enc = enc; a.x01 = x01; a.d_out = d_out; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.wout = wout; a.n = n;
    a.d_enc = d_enc; a.d_x01 = d_x01; a.g_w1 = g_w1; a.g_b1 = g_b1; a.g_w2 = g_w2; a.g_b2 = g_b2; a.g_wout = g_wout;
    a.g_bout = g_bout;
    const size_t lds_bytes = (size_t)((kDfFwdMfma + kDfBwdMfma) * 64 + 32 + (kTrainBlock / 64) * 16 * 17) * sizeof(float);
    const int64_t n_groups = (n + 15) / 16;
    int64_t blocks = qf_div_up(n_groups, kTrainBlock / 64);
    const int64_t cap = (int64_t)qf_cu_count_cached() * 2;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(deform_mlp_backward_kernel, dim3((unsigned)blocks), dim3(kTrainBlock), lds_bytes, qf_stream(stream), a);
    QF_LAUNCH_CHECK();
    return QF_OK;
}
