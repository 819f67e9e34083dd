// Packed per-ray compositing for gfx950.
//
// Replaces kaolin.render.spc mark_pack_boundaries / exponential_integration / sum_reduce as called by
// derive_properties (examples/utils.py:863-898), and nerfacc pack_info / exclusive_sum /
// exclusive_prod + the index_add_ accumulations of examples/field_rendering.py:100-156,483-573.
//
// A ray of the mesh path carries at most max_hits (25) samples, so every segmented scan here is a
// short sequential loop owned by one lane: deterministic summation order, no atomics, no
// intermediate cumsum tensors.  The kernels are HBM-bound on 20 B/sample (sigma, rgb, t).
#include "qf_common.h"

namespace {

__global__ void mark_boundaries_kernel(const int64_t *ridx, int64_t n, uint8_t *b)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        b[i] = (i == 0) || (ridx[i] != ridx[i - 1]);
}

__global__ void fill_background_kernel(int64_t n_rays, float bg, float *rgb, float *alpha, float *depth)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rays; i += (int64_t)gridDim.x * blockDim.x) {
        rgb[i * 3 + 0] = bg;
        rgb[i * 3 + 1] = bg;
        rgb[i * 3 + 2] = bg;
        alpha[i] = 0.0f;
        depth[i] = 0.0f;
    }
}

// The per-ray arithmetic of derive_properties with every rounding fixed, so that the chunked kernel, its long-ray tail
// and the tile kernel give the same bits whatever the optimiser does around them: tau = sigma * delta;
// w = exp(-cum) * (1 - exp(-tau)); cum += tau; the sums accumulate with one fma each.  (`#pragma clang fp contract(off)`
// inside the bodies: HIP's __fmul_rn / __fadd_rn are plain operators, which the default -ffp-contract=fast may still
// fuse with their neighbours after inlining; the fmas that are wanted are written as __builtin_fmaf.)
struct RayAccum {
    float cum = 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f, cd = 0.0f, ca = 0.0f;
};

__device__ __forceinline__ float sample_tau(float sigma, float delta)
{
#pragma clang fp contract(off)
    return sigma * delta;
}

__device__ __forceinline__ float sample_alpha(float tau)
{
#pragma clang fp contract(off)
    return 1.0f - expf(-tau);
}

__device__ __forceinline__ float composite_step(RayAccum &a, float tau, float alpha, float r, float g, float b, float dep)
{
#pragma clang fp contract(off)
    const float w = expf(-a.cum) * alpha;
    a.cum = a.cum + tau;
    a.cr = __builtin_fmaf(w, r, a.cr);
    a.cg = __builtin_fmaf(w, g, a.cg);
    a.cb = __builtin_fmaf(w, b, a.cb);
    a.cd = __builtin_fmaf(w, dep, a.cd);
    a.ca = a.ca + w;
    return w;
}

__device__ __forceinline__ void composite_blend(const RayAccum &a, int bg_mode, const float *bkgd, float out[3])
{
#pragma clang fp contract(off)
    const float c[3] = {a.cr, a.cg, a.cb};
    const float rest = 1.0f - a.ca;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (bg_mode == QF_BG_WHITE) out[i] = __builtin_fmaf(a.ca, c[i], rest);      // (1 - a) + a * sum(w c): the double-alpha quirk (B-1)
        else if (bg_mode == QF_BG_BLACK) out[i] = a.ca * c[i];
        else if (bg_mode == QF_BG_NONE) out[i] = c[i];     // the plain sums (nerfacc's accumulate_along_rays): no blend, no quirk
        else {
            const float back = rest * bkgd[i];
            out[i] = __builtin_fmaf(a.ca, c[i], back);
        }
    }
}

// derive_properties: a block owns DP_CHUNK consecutive samples.  It stages them (plus a halo, so that rays starting
// near the end of the chunk finish without leaving LDS) with coalesced loads; the lane that sits on a ray's first
// sample then integrates the whole ray sequentially out of LDS -- same summation order as a plain per-ray loop, so
// the result does not depend on the chunking -- and the weights go back coalesced.  Rays longer than the halo
// (occupancy-grid marching) finish from global memory.
constexpr int DP_THREADS = 256;
constexpr int DP_CHUNK = 1024;
constexpr int DP_HALO = 64;
constexpr int DP_STAGE = DP_CHUNK + DP_HALO;

__global__ __launch_bounds__(DP_THREADS) void derive_properties_kernel(
    const float *rgb_s, const float *sigma, const float *depth_s, const float *deltas, float delta_const,
    const int64_t *index_ray, int64_t n, int64_t n_rays, int bg_mode, const float *bkgd, const int32_t *sample_index,
    float *out_rgb, float *out_alpha, float *out_depth, float *weights)
{
    __shared__ float s_tau[DP_STAGE];
    __shared__ float s_alpha[DP_STAGE];              // 1 - exp(-tau), computed by all lanes before the serial part
    __shared__ float s_rgb[3 * DP_STAGE];
    __shared__ float s_dep[DP_STAGE];
    __shared__ float s_w[DP_STAGE];
    __shared__ int64_t s_ray[DP_STAGE + 1];          // [0]: the sample just before the chunk
    __shared__ uint8_t s_mine[DP_STAGE];             // weight computed by this block
    __shared__ int s_heads[DP_CHUNK];                // first samples of the rays this block owns, compacted
    __shared__ int s_nheads;
    const int64_t n_chunks = (n + DP_CHUNK - 1) / DP_CHUNK;
    for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const int64_t b0 = chunk * DP_CHUNK;
        const int staged = (int)((n - b0 < DP_STAGE) ? (n - b0) : DP_STAGE);
        const int own = (int)((n - b0 < DP_CHUNK) ? (n - b0) : DP_CHUNK);
        if (threadIdx.x == 0) {
            s_ray[0] = b0 > 0 ? index_ray[b0 - 1] : 0;
            s_nheads = 0;
        }
        for (int k = threadIdx.x; k < staged; k += DP_THREADS) {
            // sample_index: colour and density live at another position (the field kernel's processing order)
            const int64_t src = sample_index ? (int64_t)sample_index[b0 + k] : b0 + k;
            const float tau = sample_tau(sigma[src], deltas ? deltas[b0 + k] : delta_const);
            s_tau[k] = tau;
            s_alpha[k] = sample_alpha(tau);
            s_dep[k] = depth_s[b0 + k];
            s_ray[k + 1] = index_ray[b0 + k];
            s_mine[k] = 0;
            if (sample_index) {
                s_rgb[3 * k + 0] = rgb_s[src * 3 + 0];
                s_rgb[3 * k + 1] = rgb_s[src * 3 + 1];
                s_rgb[3 * k + 2] = rgb_s[src * 3 + 2];
            }
        }
        if (!sample_index)
            for (int k = threadIdx.x; k < 3 * staged; k += DP_THREADS) s_rgb[k] = rgb_s[b0 * 3 + k];
        __syncthreads();
        for (int k = threadIdx.x; k < own; k += DP_THREADS)
            if (b0 + k == 0 || s_ray[k] != s_ray[k + 1]) s_heads[atomicAdd(&s_nheads, 1)] = k;
        __syncthreads();
        const int n_heads = s_nheads;
        for (int h = threadIdx.x; h < n_heads; h += DP_THREADS) {      // consecutive lanes = different rays: no idle lanes
            const int k = s_heads[h];
            const int64_t ray = s_ray[k + 1];
            RayAccum acc;
            int j = k;
            for (; j < staged && s_ray[j + 1] == ray; ++j) {
                s_w[j] = composite_step(acc, s_tau[j], s_alpha[j], s_rgb[j * 3 + 0], s_rgb[j * 3 + 1], s_rgb[j * 3 + 2], s_dep[j]);
                s_mine[j] = 1;
            }
            if (j == staged) {                       // the ray runs past the staged window
                for (int64_t g = b0 + staged; g < n && index_ray[g] == ray; ++g) {
                    const int64_t src = sample_index ? (int64_t)sample_index[g] : g;
                    const float tau = sample_tau(sigma[src], deltas ? deltas[g] : delta_const);
                    weights[g] = composite_step(acc, tau, sample_alpha(tau), rgb_s[src * 3 + 0], rgb_s[src * 3 + 1],
                                                rgb_s[src * 3 + 2], depth_s[g]);
                }
            }
            float px[3];
            composite_blend(acc, bg_mode, bkgd, px);
            const float r = px[0], g = px[1], b = px[2], ca = acc.ca, cd = acc.cd;
            if (ray >= 0 && ray < n_rays) {         // a ray id outside the image never touches memory (weights are still written)
                out_rgb[ray * 3 + 0] = r;
                out_rgb[ray * 3 + 1] = g;
                out_rgb[ray * 3 + 2] = b;
                out_alpha[ray] = ca;
                out_depth[ray] = cd;
            }
        }
        __syncthreads();
        for (int k = threadIdx.x; k < staged; k += DP_THREADS)
            if (s_mine[k]) weights[b0 + k] = s_w[k];
        __syncthreads();
    }
}

__global__ void exp_integration_kernel(const float *feats, int c, const float *tau, const int64_t *seg_start,
                                       int64_t n_seg, int64_t n, int exclusive, float *out, float *weights)
{
    const int64_t total = n_seg * c;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = e / c;
        const int ch = (int)(e - s * c);
        const int64_t b = seg_start[s], end = (s + 1 < n_seg) ? seg_start[s + 1] : n;
        float cum = 0.0f, acc = 0.0f;
        for (int64_t j = b; j < end; ++j) {
            const float t = tau[j];
            if (!exclusive) cum += t;
            const float w = expf(-cum) * (1.0f - expf(-t));
            if (exclusive) cum += t;
            if (ch == 0) weights[j] = w;
            acc += w * feats[j * c + ch];
        }
        out[e] = acc;
    }
}

__global__ void sum_reduce_kernel(const float *feats, int c, const int64_t *seg_start, int64_t n_seg, int64_t n,
                                  float *out)
{
    const int64_t total = n_seg * c;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = e / c;
        const int ch = (int)(e - s * c);
        const int64_t b = seg_start[s], end = (s + 1 < n_seg) ? seg_start[s + 1] : n;
        float acc = 0.0f;
        for (int64_t j = b; j < end; ++j) acc += feats[j * c + ch];
        out[e] = acc;
    }
}

// (start, count) per ray by binary search in the sorted ray ids.
__global__ void pack_info_kernel(const int64_t *ridx, int64_t n, int64_t n_rays, int64_t *packed)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rays; r += (int64_t)gridDim.x * blockDim.x) {
        int64_t lo = 0, hi = n;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (ridx[mid] < r) lo = mid + 1; else hi = mid; }
        const int64_t start = lo;
        hi = n;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (ridx[mid] <= r) lo = mid + 1; else hi = mid; }
        packed[2 * r] = start;
        packed[2 * r + 1] = lo - start;
    }
}

__global__ void exclusive_scan_kernel(const float *x, const int64_t *packed, int64_t n_rays, int mode, float *out)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rays; r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = packed[2 * r], cnt = packed[2 * r + 1];
        float acc = mode ? 1.0f : 0.0f;
        for (int64_t j = b; j < b + cnt; ++j) {
            out[j] = acc;
            acc = mode ? acc * x[j] : acc + x[j];
        }
    }
}

__global__ void accumulate_kernel(const float *w, const float *values, int c, const int64_t *packed, int64_t n_rays,
                                  float *out)
{
    const int64_t total = n_rays * c;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / c;
        const int ch = (int)(e - r * c);
        const int64_t b = packed[2 * r], cnt = packed[2 * r + 1];
        float acc = 0.0f;
        for (int64_t j = b; j < b + cnt; ++j) acc += values ? w[j] * values[j * c + ch] : w[j];
        out[e] = acc;
    }
}

__global__ void render_from_density_kernel(const float *ts, const float *te, const float *sigmas, const float *rgbs,
                                           const int64_t *packed, int64_t n_rays, const float *bkgd, float *weights, float *trans, float *alphas, float *colors,
                                           float *opac, float *depths)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rays; r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = packed[2 * r], cnt = packed[2 * r + 1];
        float cum = 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f, cd = 0.0f, ca = 0.0f;
        for (int64_t j = b; j < b + cnt; ++j) {
            const float sdt = sigmas[j] * (te[j] - ts[j]);
            const float al = 1.0f - expf(-sdt);
            const float T = expf(-cum);
            const float w = T * al;
            cum += sdt;
            weights[j] = w;
            trans[j] = T;
            alphas[j] = al;
            cr += w * rgbs[j * 3 + 0];
            cg += w * rgbs[j * 3 + 1];
            cb += w * rgbs[j * 3 + 2];
            cd += w * ((ts[j] + te[j]) / 2.0f);
            ca += w;
        }
        const float eps = 1.1920928955078125e-07f;   // torch.finfo(float32).eps
        const float d = cd / fmaxf(ca, eps);
        if (bkgd) {
            cr += bkgd[0] * (1.0f - ca);
            cg += bkgd[1] * (1.0f - ca);
            cb += bkgd[2] * (1.0f - ca);
        }
        colors[r * 3 + 0] = cr;
        colors[r * 3 + 1] = cg;
        colors[r * 3 + 2] = cb;
        opac[r] = ca;
        depths[r] = d;
    }
}

// The displacement of one sample along its ray (utils.py:566-571), every operation rounded on its own as the
// reference's tensor ops are (no contraction, see composite_step): v = tanh(f) * scaling; dd = (v dx + v dy) + v dz;
// p += dd * d; t += dd.
__device__ __forceinline__ void deform_sample(float f, float scaling, float dx, float dy, float dz, float &x, float &y,
                                              float &z, float &t, float *dh = nullptr)
{
#pragma clang fp contract(off)
    const float v = tanhf(f) * scaling;
    const float vx = v * dx, vy = v * dy, vz = v * dz;
    const float dd = (vx + vy) + vz;
    const float mx = dd * dx, my = dd * dy, mz = dd * dz;
    x = x + mx;
    y = y + my;
    z = z + mz;
    t = t + dd;
    if (dh) { dh[0] = mx; dh[1] = my; dh[2] = mz; }     // the reference's ``dh = del_delta * dirs`` (utils.py:570)
}

__global__ void apply_deformation_kernel(const float *f, float scaling, const float *dirs, const float *xyz,
                                         const float *ts, int64_t n, float *xyz_out, float *ts_out, float *dh_out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float x = xyz[i * 3], y = xyz[i * 3 + 1], z = xyz[i * 3 + 2], t = ts[i];
        float dh[3];
        deform_sample(f[i], scaling, dirs[i * 3], dirs[i * 3 + 1], dirs[i * 3 + 2], x, y, z, t, dh);
        xyz_out[i * 3 + 0] = x;
        xyz_out[i * 3 + 1] = y;
        xyz_out[i * 3 + 2] = z;
        ts_out[i] = t;
        if (dh_out) { dh_out[i * 3] = dh[0]; dh_out[i * 3 + 1] = dh[1]; dh_out[i * 3 + 2] = dh[2]; }
    }
}

}  // namespace

#define QF_SIMPLE_LAUNCH(kernel, count, ...)                                                            \
    hipLaunchKernelGGL(kernel, dim3(qf_grid_1d((count), 256)), dim3(256), 0, qf_stream(stream), __VA_ARGS__); \
    QF_LAUNCH_CHECK();

extern "C" int qf_mark_pack_boundaries(const int64_t *ridx, int64_t n, uint8_t *boundary, void *stream)
{
    if (n < 0) return QF_ERR_INVALID_ARGUMENT;
    if (n == 0) return QF_OK;
    if (!ridx || !boundary) return QF_ERR_INVALID_ARGUMENT;
    QF_SIMPLE_LAUNCH(mark_boundaries_kernel, n, ridx, n, boundary);
    return QF_OK;
}

extern "C" int qf_exponential_integration(const float *feats, int32_t c, const float *tau, const int64_t *seg_start,
                                          int64_t n_seg, int64_t n, int32_t exclusive, float *out, float *weights,
                                          void *stream)
{
    if (n < 0 || n_seg < 0 || c < 1) return QF_ERR_INVALID_ARGUMENT;
    if (n_seg == 0) return QF_OK;
    if (!feats || !tau || !seg_start || !out || !weights) return QF_ERR_INVALID_ARGUMENT;
    QF_SIMPLE_LAUNCH(exp_integration_kernel, n_seg * c, feats, (int)c, tau, seg_start, n_seg, n, (int)exclusive, out,
                     weights);
    return QF_OK;
}

extern "C" int qf_sum_reduce(const float *feats, int32_t c, const int64_t *seg_start, int64_t n_seg, int64_t n,
                             float *out, void *stream)
{
    if (n < 0 || n_seg < 0 || c < 1) return QF_ERR_INVALID_ARGUMENT;
    if (n_seg == 0) return QF_OK;
    if (!feats || !seg_start || !out) return QF_ERR_INVALID_ARGUMENT;
    QF_SIMPLE_LAUNCH(sum_reduce_kernel, n_seg * c, feats, (int)c, seg_start, n_seg, n, out);
    return QF_OK;
}

extern "C" int qf_derive_properties(const float *rgb_s, const float *sigma, const float *depth, const float *deltas,
                                    float delta_const, const int64_t *index_ray, int64_t n, int64_t n_rays,
                                    int32_t bg_mode, const float *bkgd, const int32_t *sample_index, float *out_rgb,
                                    float *out_alpha, float *out_depth, float *weights, void *stream)
{
    if (n < 0 || n_rays < 0 || bg_mode < 0 || bg_mode > 3) return QF_ERR_INVALID_ARGUMENT;
    if (n_rays > 0 && (!out_rgb || !out_alpha || !out_depth)) return QF_ERR_INVALID_ARGUMENT;
    if (bg_mode == QF_BG_CUSTOM && !bkgd) return QF_ERR_INVALID_ARGUMENT;
    if (n_rays > 0) {
        QF_SIMPLE_LAUNCH(fill_background_kernel, n_rays, n_rays, (bg_mode == QF_BG_BLACK || bg_mode == QF_BG_NONE) ? 0.0f : 1.0f, out_rgb,
                         out_alpha, out_depth);
    }
    if (n == 0) return QF_OK;
    if (n_rays == 0) return QF_ERR_INVALID_ARGUMENT;             // samples without an image to composite them into
    if (!rgb_s || !sigma || !depth || !index_ray || !weights) return QF_ERR_INVALID_ARGUMENT;
    const int64_t n_chunks = (n + DP_CHUNK - 1) / DP_CHUNK;
    hipLaunchKernelGGL(derive_properties_kernel, dim3((unsigned)(n_chunks < 65536 ? n_chunks : 65536)), dim3(DP_THREADS), 0,
                       qf_stream(stream), rgb_s, sigma, depth, deltas, delta_const, index_ray, n, n_rays, (int)bg_mode, bkgd,
                       sample_index, out_rgb, out_alpha, out_depth, weights);
    QF_LAUNCH_CHECK();
    return QF_OK;
}

extern "C" int qf_pack_info(const int64_t *ray_indices, int64_t n, int64_t n_rays, int64_t *packed_info, void *stream)
{
    if (n < 0 || n_rays < 0) return QF_ERR_INVALID_ARGUMENT;
    if (n_rays == 0) return QF_OK;
    if ((n > 0 && !ray_indices) || !packed_info) return QF_ERR_INVALID_ARGUMENT;
    QF_SIMPLE_LAUNCH(pack_info_kernel, n_rays, ray_indices, n, n_rays, packed_info);
    return QF_OK;
}

extern "C" int qf_exclusive_scan(const float *x, const int64_t *packed_info, int64_t n_rays, int64_t n, int32_t mode,
                                 float *out, void *stream)
{
    if (n < 0 || n_rays < 0 || (mode != 0 && mode != 1)) return QF_ERR_INVALID_ARGUMENT;
    if (n == 0 || n_rays == 0) return QF_OK;
    if (!x || !packed_info || !out) return QF_ERR_INVALID_ARGUMENT;
    QF_SIMPLE_LAUNCH(exclusive_scan_kernel, n_rays, x, packed_info, n_rays, (int)mode, out);
    return QF_OK;
}

extern "C" int qf_accumulate_along_rays(const float *weights, const float *values, int32_t c,
                                        const int64_t *packed_info, int64_t n_rays, int64_t n, float *out, void *stream)
{
    if (n < 0 || n_rays < 0 || c < 1) return QF_ERR_INVALID_ARGUMENT;
    if (n_rays == 0) return QF_OK;
    if ((n > 0 && !weights) || !packed_info || !out) return QF_ERR_INVALID_ARGUMENT;
    QF_SIMPLE_LAUNCH(accumulate_kernel, n_rays * c, weights, values, (int)c, packed_info, n_rays, out);
    return QF_OK;
}

extern "C" int qf_render_from_density(const float *t_starts, const float *t_ends, const float *sigmas,
                                      const float *rgbs, const int64_t *packed_info, int64_t n_rays, int64_t n,
                                      const float *bkgd, float *weights, float *trans, float *alphas, float *colors,
                                      float *opacities, float *depths, void *stream)
{
    if (n < 0 || n_rays < 0) return QF_ERR_INVALID_ARGUMENT;
    if (n_rays == 0) return QF_OK;
    if (!packed_info || !colors || !opacities || !depths) return QF_ERR_INVALID_ARGUMENT;
    if (n > 0 && (!t_starts || !t_ends || !sigmas || !rgbs || !weights || !trans || !alphas)) return QF_ERR_INVALID_ARGUMENT;
    QF_SIMPLE_LAUNCH(render_from_density_kernel, n_rays, t_starts, t_ends, sigmas, rgbs, packed_info, n_rays,
                     bkgd, weights, trans, alphas, colors, opacities, depths);
    return QF_OK;
}

extern "C" int qf_apply_deformation(const float *f, float scaling, const float *dirs, const float *xyz, const float *ts,
                                    int64_t n, float *xyz_out, float *ts_out, float *dh_out, void *stream)
{
    if (n < 0) return QF_ERR_INVALID_ARGUMENT;
    if (n == 0) return QF_OK;
    if (!f || !dirs || !xyz || !ts || !xyz_out || !ts_out) return QF_ERR_INVALID_ARGUMENT;
    QF_SIMPLE_LAUNCH(apply_deformation_kernel, n, f, scaling, dirs, xyz, ts, n, xyz_out, ts_out, dh_out);
    return QF_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Spatially coherent processing order for the field kernel.  Samples are stored ray-major (the reference's
// layout), so 16 consecutive samples are 1-2 rays' worth of hits strung along the ray: no two of them share a
// hash-grid cell.  Re-ordering the PROCESSING order as (8x8 pixel tile, hit rank, pixel in tile) puts the
// rank-k hits of neighbouring pixels -- points on the same surface patch, a pixel footprint apart -- into the
// same wave pass, so their gathers coalesce / hit L1 on every level whose cells are larger than a pixel.
// One wave per tile, lane = pixel; ballots give each (pixel, rank) its slot.
namespace {

__device__ __forceinline__ int tile_lane_ray(int tile, int lane, int w, int h, int tiles_x, int64_t *ray)
{
    return qf_tile_lane_ray(tile, lane, w, h, tiles_x, 0, ray);
}

__global__ __launch_bounds__(64) void tile_totals_kernel(const int32_t *hit_count, int w, int h, int tiles_x,
                                                         int64_t *tile_total)
{
    const int tile = blockIdx.x, lane = threadIdx.x;
    int64_t ray = 0;
    int cnt = 0;
    if (tile_lane_ray(tile, lane, w, h, tiles_x, &ray)) cnt = hit_count[ray];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (lane == 0) tile_total[tile] = cnt;
}

// order[pos] = sample (ray-major index) for every position of the coherent sequence; optionally the inverse map too.
__global__ __launch_bounds__(64) void coherent_order_kernel(const int32_t *hit_count, const int64_t *ray_offset,
                                                            const int64_t *tile_base, int w, int h, int tiles_x,
                                                            int32_t *order, int32_t *inverse, int band_rows)
{
    const int tile = blockIdx.x, lane = threadIdx.x;
    int64_t ray = 0;
    int cnt = 0;
    int64_t first = 0;
    if (qf_tile_lane_ray(tile, lane, w, h, tiles_x, band_rows, &ray)) { cnt = hit_count[ray]; first = ray_offset[ray]; }
    int64_t base = tile_base[tile];
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int k = 0;; ++k) {
        const unsigned long long mask = __ballot(cnt > k);
        if (mask == 0ull) break;                      // wave-uniform exit
        if (cnt > k) {
            const int64_t pos = base + __popcll(mask & below), smp = first + k;
            if (order) order[pos] = (int32_t)smp;
            if (inverse) inverse[smp] = (int32_t)pos;
        }
        base += __popcll(mask);
    }
}

// derive_properties on a frame whose colours / densities / depths are in the coherent order above (what the field
// kernel streams): one wave per 8x8 tile, lane = pixel, step k composites the rank-k samples of the tile's pixels --
// the same ballots that defined the order give every lane its position, so all loads are contiguous runs and neither
// the inverse map nor index_ray is read.  The per-ray arithmetic is derive_properties_kernel's (composite_step /
// composite_blend, rank order = depth order), bit for bit.  Pixels without samples write their own background: one
// launch instead of fill + chunked kernel.  The loads of step k+1 are issued before the arithmetic of step k.
__global__ __launch_bounds__(64) void composite_tiles_kernel(
    const float *__restrict__ rgb_c, const float *__restrict__ sigma_c, const float *__restrict__ depth_c,
    float delta_const, const int32_t *__restrict__ hit_count, int max_hits, const int64_t *__restrict__ tile_base, int w,
    int h, int tiles_x, int bg_mode, const float *__restrict__ bkgd, float *__restrict__ out_rgb, float *__restrict__ out_alpha,
    float *__restrict__ out_depth, float *__restrict__ weights_c, float *__restrict__ out_packed)
{
    const int tile = blockIdx.x, lane = threadIdx.x;
    int64_t ray = 0;
    int cnt = 0;
    const int inside = tile_lane_ray(tile, lane, w, h, tiles_x, &ray);
    if (inside) cnt = hit_count[ray] < max_hits ? hit_count[ray] : max_hits;
    int64_t base = tile_base[tile];
    const unsigned long long below = (1ull << lane) - 1ull;
    RayAccum acc;
    unsigned long long mask = __ballot(cnt > 0);
    int64_t pos = base + __popcll(mask & below);
    float sg = 0.0f, r = 0.0f, g = 0.0f, b = 0.0f, dep = 0.0f;
    if (cnt > 0) { sg = sigma_c[pos]; r = rgb_c[pos * 3]; g = rgb_c[pos * 3 + 1]; b = rgb_c[pos * 3 + 2]; dep = depth_c[pos]; }
    for (int k = 0; mask != 0ull; ++k) {              // wave-uniform
        base += __popcll(mask);
        const unsigned long long next = __ballot(cnt > k + 1);
        const int64_t npos = base + __popcll(next & below);
        float nsg = 0.0f, nr = 0.0f, ng = 0.0f, nb = 0.0f, ndep = 0.0f;
        if (cnt > k + 1) { nsg = sigma_c[npos]; nr = rgb_c[npos * 3]; ng = rgb_c[npos * 3 + 1]; nb = rgb_c[npos * 3 + 2]; ndep = depth_c[npos]; }
        if (cnt > k) {
            const float tau = sample_tau(sg, delta_const);
            const float wt = composite_step(acc, tau, sample_alpha(tau), r, g, b, dep);
            if (weights_c) weights_c[pos] = wt;
        }
        mask = next; pos = npos; sg = nsg; r = nr; g = ng; b = nb; dep = ndep;
    }
    if (!inside) return;
    float px[3];
    if (cnt > 0) {
        composite_blend(acc, bg_mode, bkgd, px);
    } else {                                           // fill_background_kernel's values
        px[0] = px[1] = px[2] = (bg_mode == QF_BG_BLACK || bg_mode == QF_BG_NONE) ? 0.0f : 1.0f;
    }
    if (out_packed) {                  // [n, 5] = rgb | alpha | depth per pixel: what the band gather sends, no concat launch
        float *o = out_packed + ray * 5;
        o[0] = px[0]; o[1] = px[1]; o[2] = px[2]; o[3] = acc.ca; o[4] = acc.cd;
        return;
    }
    out_rgb[ray * 3 + 0] = px[0];
    out_rgb[ray * 3 + 1] = px[1];
    out_rgb[ray * 3 + 2] = px[2];
    out_alpha[ray] = acc.ca;
    out_depth[ray] = acc.cd;
}

// quadrature points per pixel row, sum_x min(hit_count, K): what parallel.ShardedFrameRenderer balances its bands with
// (one wave per row; replaces a clamp, a reshape and a row sum in torch)
__global__ __launch_bounds__(64) void row_sample_counts_kernel(const int32_t *__restrict__ hit_count, int max_hits, int w, int h,
                                                               float *__restrict__ out_rows)
{
    const int row = blockIdx.x, lane = threadIdx.x;
    int sum = 0;
    for (int x = lane; x < w; x += 64) {
        const int c = hit_count[(int64_t)row * w + x];
        sum += c < 0 ? 0 : (c < max_hits ? c : max_hits);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane == 0) out_rows[row] = (float)sum;
}

// The "before" evaluation of a frame in the coherent order (train_finetune.py:696 -> utils.py:555-572 and the re-sort of
// mesh_utils.py:389-403): displace every sample along its ray by the deformation field's output, then put each ray's
// samples back in depth order -- a stable sort by the new fp32 depth, which is what np.lexsort((depth, ray)) does to a
// ray's run.  One wave per 8x8 tile, lane = pixel: a ray's samples sit at the slots the ballots give its ranks, the
// lane computes its <= K new depths into an LDS column, sorts (source rank, depth) there -- almost always already in
// order -- and writes rank k's slot from the source it drew.  Out of place (xyz_out / depth_out); directions do not
// change.  Slots past a tile's kept samples (re-origin gaps) are copied through.
__global__ __launch_bounds__(64) void deform_resort_tiles_kernel(
    const float *__restrict__ f_c, float scaling, const float *__restrict__ xyz_c, const float *__restrict__ dirs_c,
    const float *__restrict__ depth_c, const int32_t *__restrict__ hit_count, int max_hits,
    const int64_t *__restrict__ tile_base, int64_t total, const int64_t *__restrict__ total_dev, int w, int h, int tiles_x,
    int n_tiles, float *__restrict__ xyz_out, float *__restrict__ depth_out)
{
    extern __shared__ float dr_lds[];
    if (total_dev) { const int64_t td = *total_dev; total = td < total ? (td > 0 ? td : 0) : total; }   // device-side slot count
    const int K = max_hits;
    float *col_t = dr_lds + threadIdx.x;                                  // [K][64] new depth of source rank k
    int *col_s = reinterpret_cast<int *>(dr_lds + (size_t)K * 64) + threadIdx.x;      // [K][64] source rank at sorted place k
    const int tile = blockIdx.x, lane = threadIdx.x;
    int64_t ray = 0;
    int cnt = 0;
    if (tile_lane_ray(tile, lane, w, h, tiles_x, &ray)) cnt = hit_count[ray] < K ? hit_count[ray] : K;
    const int64_t base0 = tile_base[tile];
    const unsigned long long below = (1ull << lane) - 1ull;
    // slot of rank k = base0 + (slots of the ranks before) + (pixels before this one that have a rank k): the per-rank
    // offsets are wave-uniform, keep them in SGPR-like registers by recomputing the ballots in both passes
    int64_t base = base0;
    for (int k = 0;; ++k) {
        const unsigned long long mask = __ballot(cnt > k);
        if (mask == 0ull) break;
        if (cnt > k) {
            const int64_t c = base + __popcll(mask & below);
            float x = xyz_c[c * 3], y = xyz_c[c * 3 + 1], z = xyz_c[c * 3 + 2], t = depth_c[c];
            deform_sample(f_c[c], scaling, dirs_c[c * 3], dirs_c[c * 3 + 1], dirs_c[c * 3 + 2], x, y, z, t);
            col_t[k * 64] = t;
            col_s[k * 64] = k;
        }
        base += __popcll(mask);
    }
    const int64_t written_end = base;
    for (int i = 1; i < cnt; ++i) {                           // stable insertion by the new depth
        const float t = col_t[i * 64];
        const int src = col_s[i * 64];
        int j = i - 1;
        while (j >= 0 && col_t[j * 64] > t) { col_t[(j + 1) * 64] = col_t[j * 64]; col_s[(j + 1) * 64] = col_s[j * 64]; --j; }
        col_t[(j + 1) * 64] = t;
        col_s[(j + 1) * 64] = src;
    }
    // second pass over the ranks: sorted place k lands in rank k's slot.  The depths go out first; their column then
    // holds the ranks' slots (relative to the tile), which the sources are looked up through.
    int *col_slot = reinterpret_cast<int *>(col_t);
    base = base0;
    for (int k = 0;; ++k) {
        const unsigned long long mask = __ballot(cnt > k);
        if (mask == 0ull) break;
        if (cnt > k) {
            const int64_t c = base + __popcll(mask & below);
            depth_out[c] = col_t[k * 64];
            col_slot[k * 64] = (int)(c - base0);
        }
        base += __popcll(mask);
    }
    for (int k = 0; k < cnt; ++k) {
        const int64_t dst = base0 + col_slot[k * 64], src = base0 + col_slot[col_s[k * 64] * 64];
        float x = xyz_c[src * 3], y = xyz_c[src * 3 + 1], z = xyz_c[src * 3 + 2], t = depth_c[src];
        deform_sample(f_c[src], scaling, dirs_c[src * 3], dirs_c[src * 3 + 1], dirs_c[src * 3 + 2], x, y, z, t);
        xyz_out[dst * 3 + 0] = x;
        xyz_out[dst * 3 + 1] = y;
        xyz_out[dst * 3 + 2] = z;
    }
    const int64_t end = tile + 1 < n_tiles ? tile_base[tile + 1] : total;
    for (int64_t c = written_end + lane; c < end; c += 64) {  // re-origin gaps: finite points nobody composites
        xyz_out[c * 3 + 0] = xyz_c[c * 3 + 0];
        xyz_out[c * 3 + 1] = xyz_c[c * 3 + 1];
        xyz_out[c * 3 + 2] = xyz_c[c * 3 + 2];
        depth_out[c] = depth_c[c];
    }
}

}  // namespace

extern "C" int qf_deform_resort_tiles(const float *f_c, float scaling, const float *xyz_c, const float *dirs_c,
                                      const float *depth_c, const int32_t *hit_count, int32_t max_hits,
                                      const int64_t *tile_base, int64_t total, const int64_t *total_device, int32_t width,
                                      int32_t height, float *xyz_out, float *depth_out, void *stream)
{
    if (width < 1 || height < 1 || max_hits < 1 || max_hits > 64) return QF_ERR_INVALID_ARGUMENT;
    if (!f_c || !xyz_c || !dirs_c || !depth_c || !hit_count || !tile_base || total < 0 || !xyz_out || !depth_out)
        return QF_ERR_INVALID_ARGUMENT;
    if (xyz_out == xyz_c || depth_out == depth_c) return QF_ERR_INVALID_ARGUMENT;          // out of place
    const int tiles_x = (width + 7) / 8, tiles_y = (height + 7) / 8;
    hipLaunchKernelGGL(deform_resort_tiles_kernel, dim3(tiles_x * tiles_y), dim3(64), (size_t)max_hits * 64 * 8, qf_stream(stream),
                       f_c, scaling, xyz_c, dirs_c, depth_c, hit_count, (int)max_hits, tile_base, total, total_device, (int)width,
                       (int)height, tiles_x, tiles_x * tiles_y, xyz_out, depth_out);
    QF_LAUNCH_CHECK();
    return QF_OK;
}

extern "C" int qf_composite_tiles(const float *rgb_c, const float *sigma_c, const float *depth_c, float delta_const,
                                  const int32_t *hit_count, int32_t max_hits, const int64_t *tile_base, int32_t width,
                                  int32_t height, int32_t bg_mode, const float *bkgd, float *out_rgb, float *out_alpha,
                                  float *out_depth, float *weights_c, float *out_packed, void *stream)
{
    if (width < 1 || height < 1 || max_hits < 1 || bg_mode < 0 || bg_mode > 3) return QF_ERR_INVALID_ARGUMENT;
    if (!rgb_c || !sigma_c || !depth_c || !hit_count || !tile_base) return QF_ERR_INVALID_ARGUMENT;
    if (!out_packed && (!out_rgb || !out_alpha || !out_depth)) return QF_ERR_INVALID_ARGUMENT;
    if (bg_mode == QF_BG_CUSTOM && !bkgd) return QF_ERR_INVALID_ARGUMENT;
    const int tiles_x = (width + 7) / 8, tiles_y = (height + 7) / 8;
    hipLaunchKernelGGL(composite_tiles_kernel, dim3(tiles_x * tiles_y), dim3(64), 0, qf_stream(stream), rgb_c, sigma_c,
                       depth_c, delta_const, hit_count, (int)max_hits, tile_base, (int)width, (int)height, tiles_x, (int)bg_mode, bkgd,
                       out_rgb, out_alpha, out_depth, weights_c, out_packed);
    QF_LAUNCH_CHECK();
    return QF_OK;
}

extern "C" int qf_row_sample_counts(const int32_t *hit_count, int32_t max_hits, int32_t width, int32_t height,
                                    float *out_rows, void *stream)
{
    if (width < 1 || height < 1 || max_hits < 1 || !hit_count || !out_rows) return QF_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(row_sample_counts_kernel, dim3(height), dim3(64), 0, qf_stream(stream), hit_count, (int)max_hits,
                       (int)width, (int)height, out_rows);
    QF_LAUNCH_CHECK();
    return QF_OK;
}

extern "C" int qf_tile_totals(const int32_t *hit_count, int32_t width, int32_t height, int64_t *tile_total, void *stream)
{
    if (width < 1 || height < 1 || !hit_count || !tile_total) return QF_ERR_INVALID_ARGUMENT;
    const int tiles_x = (width + 7) / 8, tiles_y = (height + 7) / 8;
    hipLaunchKernelGGL(tile_totals_kernel, dim3(tiles_x * tiles_y), dim3(64), 0, qf_stream(stream), hit_count, (int)width,
                       (int)height, tiles_x, tile_total);
    QF_LAUNCH_CHECK();
    return QF_OK;
}

extern "C" int qf_coherent_order(const int32_t *hit_count, const int64_t *ray_offset, const int64_t *tile_base,
                                 int32_t width, int32_t height, int32_t *order, void *stream)
{
    if (width < 1 || height < 1 || !hit_count || !ray_offset || !tile_base || !order) return QF_ERR_INVALID_ARGUMENT;
    const int tiles_x = (width + 7) / 8, tiles_y = (height + 7) / 8;
    hipLaunchKernelGGL(coherent_order_kernel, dim3(tiles_x * tiles_y), dim3(64), 0, qf_stream(stream), hit_count,
                       ray_offset, tile_base, (int)width, (int)height, tiles_x, order, (int32_t *)nullptr, 0);
    QF_LAUNCH_CHECK();
    return QF_OK;
}

extern "C" int qf_coherent_layout(const int32_t *hit_count, const int64_t *ray_offset, const int64_t *tile_base,
                                  int32_t width, int32_t height, int32_t *order, int32_t *inverse, int32_t band_rows,
                                  void *stream)
{
    if (width < 1 || height < 1 || band_rows < 0 || !hit_count || !ray_offset || !tile_base || !inverse)    // order may be NULL
        return QF_ERR_INVALID_ARGUMENT;
    const int tiles_x = (width + 7) / 8;
    hipLaunchKernelGGL(coherent_order_kernel, dim3((unsigned)qf_banded_tiles(width, height, band_rows)), dim3(64), 0,
                       qf_stream(stream), hit_count, ray_offset, tile_base, (int)width, (int)height, tiles_x, order, inverse,
                       (int)band_rows);
    QF_LAUNCH_CHECK();
    return QF_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Callers on either side of the path.
namespace {

// Full-image rays, SubjectLoader.fetch_data (datasets/nerf_synthetic.py:341-373): pixel centres, OpenGL axes
// (-y, -z), directions = sum_j camera_dir[j] * c2w[i][j], viewdirs = directions / |directions|.
__global__ void generate_rays_kernel(qf_camera cam, int opengl, float *origins, float *viewdirs)
{
#pragma clang fp contract(off)   // individually rounded products and sums, as torch evaluates them
    const int64_t n = (int64_t)cam.width * cam.height;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % cam.width), y = (int)(i / cam.width);
        const float sgn = opengl ? -1.0f : 1.0f;
        const float cdx = ((float)x - cam.cx + 0.5f) / cam.fx;
        const float cdy = ((float)y - cam.cy + 0.5f) / cam.fy * sgn;
        const float cdz = sgn;
        float d[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) d[k] = (cdx * cam.c2w[4 * k + 0] + cdy * cam.c2w[4 * k + 1]) + cdz * cam.c2w[4 * k + 2];
        const float nrm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            origins[i * 3 + k] = cam.c2w[4 * k + 3];
            viewdirs[i * 3 + k] = d[k] / nrm;
        }
    }
}

// out[index[i]] = max(out[index[i]], values[i]) -- torch_scatter.scatter_max as used for triangle pruning
// (prune_mesh_after_finetuning.py:355-357).  Float max through integer atomics on the IEEE bit pattern.
__global__ void scatter_max_kernel(const float *values, const int64_t *index, int64_t n, int64_t n_out, float *out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = index[i];
        if (j < 0 || j >= n_out) continue;
        const float v = values[i];
        if (v != v) continue;
        if (v >= 0.0f) atomicMax(reinterpret_cast<int *>(out + j), __float_as_int(v));
        else atomicMin(reinterpret_cast<unsigned int *>(out + j), __float_as_uint(v));
    }
}

}  // namespace

extern "C" int qf_generate_rays(const qf_camera *cam, int32_t opengl, float *origins, float *viewdirs, void *stream)
{
    if (!cam || cam->width < 1 || cam->height < 1 || !(cam->fx > 0.0f) || !(cam->fy > 0.0f) || !origins || !viewdirs)
        return QF_ERR_INVALID_ARGUMENT;
    const int64_t n = (int64_t)cam->width * cam->height;
    hipLaunchKernelGGL(generate_rays_kernel, dim3(qf_grid_1d(n, 256)), dim3(256), 0, qf_stream(stream), *cam,
                       (int)opengl, origins, viewdirs);
    QF_LAUNCH_CHECK();
    return QF_OK;
}

extern "C" int qf_scatter_max(const float *values, const int64_t *index, int64_t n, int64_t n_out, float *out,
                              void *stream)
{
    if (n < 0 || n_out < 0) return QF_ERR_INVALID_ARGUMENT;
    if (n == 0) return QF_OK;
    if (!values || !index || !out) return QF_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(scatter_max_kernel, dim3(qf_grid_1d(n, 256)), dim3(256), 0, qf_stream(stream), values, index, n,
                       n_out, out);
    QF_LAUNCH_CHECK();
    return QF_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Coherent processing order of a SPLIT (the 160 000-ray windows of generate_splits, train_finetune.py:419-439) from
// nothing but its sorted ray ids: the reference-shaped entry point render_image_finetune_with_occgrid receives the six
// sample tensors of a window of a frame and no processing order; without one the field kernel runs ray-major at
// 0.58 ms per 10^6 points instead of 0.34.  Four launches, no host round trip:
//   1. split_rays_kernel: per ray r of the frame, ray_offset[r] = lower_bound(index_ray, r) (binary search in the
//      sorted ids) and hit_count[r] = the run length; the same launch verifies that the ids ARE sorted and in range
//      and raises *invalid otherwise;
//   2.+3. the tile totals and their exclusive scan (frame_partials_kernel / frame_scan_kernel via qf_tile_offsets);
//   4. coherent_order_kernel.  When *invalid is set it writes the identity instead -- an unsorted batch renders in its
//      own order rather than through a layout derived from garbage counts (whose positions could exceed n).
namespace {

__global__ void split_rays_kernel(const int64_t *__restrict__ index_ray, int64_t n, int64_t n_rays,
                                  int32_t *__restrict__ hit_count, int64_t *__restrict__ ray_offset, int32_t *invalid)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int bad = 0;
    for (int64_t i = t0; i < n; i += stride) {
        const int64_t r = index_ray[i];
        bad |= (r < 0) | (r >= n_rays) | (i + 1 < n && index_ray[i + 1] < r);
    }
    if (bad) atomicOr(invalid, 1);
    // a window of a frame holds a quarter of its rays: everything before the first id / behind the last is empty
    const int64_t r_first = n > 0 ? index_ray[0] : n_rays, r_last = n > 0 ? index_ray[n - 1] : -1;
    for (int64_t r = t0; r <= n_rays; r += stride) {
        if (r < r_first || r > r_last) {             // (ids out of order are caught above; the layout is then the identity)
            ray_offset[r] = r < r_first ? 0 : n;
            if (r < n_rays) hit_count[r] = 0;
            continue;
        }
        int64_t lo = 0, hi = n;                      // first i with index_ray[i] >= r
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (index_ray[mid] < r) lo = mid + 1; else hi = mid;
        }
        ray_offset[r] = lo;
        if (r < n_rays) {
            int64_t e = lo, he = n;                  // first i with index_ray[i] > r: runs are short, gallop from lo
            int64_t step = 1;
            while (e + step < he && index_ray[e + step] <= r) { e += step; step <<= 1; }
            he = e + step < he ? e + step : he;
            while (e < he) {
                const int64_t mid = (e + he) >> 1;
                if (index_ray[mid] <= r) e = mid + 1; else he = mid;
            }
            hit_count[r] = (int32_t)(e - lo);
        }
    }
}

__global__ __launch_bounds__(64) void split_order_kernel(const int32_t *hit_count, const int64_t *ray_offset,
                                                         const int64_t *tile_base, int w, int h, int tiles_x, int64_t n,
                                                         const int32_t *invalid, int32_t *order, int32_t *inverse)
{
    const int tile = blockIdx.x, lane = threadIdx.x;
    if (*invalid) {                                   // identity, the tiles sharing [0, n)
        const int64_t per = (n + gridDim.x - 1) / gridDim.x, i0 = per * tile, i1 = i0 + per < n ? i0 + per : n;
        for (int64_t i = i0 + lane; i < i1; i += 64) {
            if (order) order[i] = (int32_t)i;
            inverse[i] = (int32_t)i;
        }
        return;
    }
    int64_t ray = 0;
    int cnt = 0;
    int64_t first = 0;
    if (tile_lane_ray(tile, lane, w, h, tiles_x, &ray)) { cnt = hit_count[ray]; first = ray_offset[ray]; }
    int64_t base = tile_base[tile];
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int k = 0;; ++k) {
        const unsigned long long mask = __ballot(cnt > k);
        if (mask == 0ull) break;
        if (cnt > k) {
            const int64_t pos = base + __popcll(mask & below), smp = first + k;
            if (order) order[pos] = (int32_t)smp;
            inverse[smp] = (int32_t)pos;
        }
        base += __popcll(mask);
    }
}

// MeshFinetune.update_d (mesh_utils.py:126-131): cache_d[tri] += d * w, cache_w[tri] += w, one launch (the reference
// issues two torch index_add_).  d == NULL: the displacement is identically zero (scaling 0) and only cache_w moves.
// cache [n_faces, 4] = (sum d w | sum w) per triangle, one 16-byte row: FOUR lanes serve a sample, lane q adds component
// q, so the four atomics of a sample are one instruction on one 16-byte piece of one line -- the memory pipeline carries
// them as ONE request (the trick of grid_backward_table_kernel).  The atomic rate is per request (~1.9e10/s chip-wide),
// so 10^6 samples cost ~50 us instead of the ~200 us of four separate atomics per lane.
// A triangle id outside [0, n_faces) cannot be accumulated (torch's index_add_ / torch_scatter raise on it): the sample
// is skipped and, when the caller passes a counter, COUNTED -- MeshFinetune raises on a non-zero count the next time
// it synchronises anyway (update_faces), so corrupt ids (a stale sample set after a mesh swap) do not vanish silently.
__global__ void mesh_update_d_kernel(const float *__restrict__ d, const float *__restrict__ w,
                                     const int64_t *__restrict__ index_tri, int64_t n, int64_t n_faces, float *cache,
                                     int32_t *__restrict__ skipped)
{
#pragma clang fp contract(off)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (!d) {                                         // zero displacement: only the weight column moves
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
            const int64_t t = index_tri[i];
            if (t >= 0 && t < n_faces) unsafeAtomicAdd(cache + t * 4 + 3, w[i]);
            else if (skipped) atomicAdd(skipped, 1);
        }
        return;
    }
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < 4 * n; e += stride) {
        const int64_t i = e >> 2;
        const int c = (int)(e & 3);
        const int64_t t = index_tri[i];
        if (t < 0 || t >= n_faces) {
            if (skipped && c == 0) atomicAdd(skipped, 1);
            continue;
        }
        const float wi = w[i];
        unsafeAtomicAdd(cache + t * 4 + c, c < 3 ? d[i * 3 + c] * wi : wi);
    }
}

// The same accumulation with the duplicates of a chunk merged first (round 4).  Samples arrive sorted by (ray, depth)
// and a triangle of the quadrature mesh covers several pixels, so the x-neighbours of a ray meet the same triangles a few
// samples later: of 1024 consecutive samples of the bench frame only about a third carry a triangle nobody else in the
// chunk carries.  A workgroup merges its chunk in an LDS hash table (open addressing, 2048 slots for 1024 samples:
// ds_cmpst on the key, ds_add_f32 on the sums) and then issues ONE global atomic request per distinct triangle --
// the global atomic rate (~1.9e10 requests/s chip-wide) is what a frame's 3.9 M samples cost 0.2 ms at.
// The order of the fp32 additions differs from the unmerged kernel's (which is itself unordered: atomics).
constexpr int kUdBlock = 256, kUdChunk = 1024, kUdSlots = 2048;

template <bool kHasD>
__global__ __launch_bounds__(kUdBlock) void mesh_update_d_merged_kernel(const float *__restrict__ d, const float *__restrict__ w,
                                                                        const int64_t *__restrict__ index_tri, int64_t n,
                                                                        int64_t n_faces, float *cache,
                                                                        int32_t *__restrict__ skipped)
{
#pragma clang fp contract(off)
    __shared__ int s_key[kUdSlots];
    __shared__ float s_val[kUdSlots * (kHasD ? 4 : 1)];
    const int64_t n_chunks = (n + kUdChunk - 1) / kUdChunk;
    for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        for (int s_ = threadIdx.x; s_ < kUdSlots; s_ += kUdBlock) {
            s_key[s_] = -1;
            if (kHasD) { s_val[4 * s_] = 0.0f; s_val[4 * s_ + 1] = 0.0f; s_val[4 * s_ + 2] = 0.0f; s_val[4 * s_ + 3] = 0.0f; }
            else s_val[s_] = 0.0f;
        }
        __syncthreads();
        const int64_t base = chunk * kUdChunk;
#pragma unroll
        for (int k = 0; k < kUdChunk / kUdBlock; ++k) {
            const int64_t i = base + k * kUdBlock + threadIdx.x;
            if (i >= n) continue;
            const int64_t t = index_tri[i];
            if (t < 0 || t >= n_faces) {
                if (skipped) atomicAdd(skipped, 1);
                continue;
            }
            const float wi = w[i];
            const int key = (int)t;                                   // n_faces < 2^31: the cache is indexed by it
            unsigned h = ((unsigned)key * 2654435761u) >> (32 - 11);  // 11 bits = kUdSlots
            for (;;) {
                const int prev = atomicCAS(&s_key[h], -1, key);
                if (prev == -1 || prev == key) break;
                h = (h + 1) & (kUdSlots - 1);
            }
            if (kHasD) {
                atomicAdd(&s_val[4 * h + 0], d[i * 3 + 0] * wi);
                atomicAdd(&s_val[4 * h + 1], d[i * 3 + 1] * wi);
                atomicAdd(&s_val[4 * h + 2], d[i * 3 + 2] * wi);
                atomicAdd(&s_val[4 * h + 3], wi);
            } else {
                atomicAdd(&s_val[h], wi);
            }
        }
        __syncthreads();
        if (kHasD) {
            // four lanes per slot: the row's four atomics are one 16-byte request, as in the unmerged kernel
            for (int e = threadIdx.x; e < kUdSlots * 4; e += kUdBlock) {
                const int key = s_key[e >> 2];
                if (key >= 0) unsafeAtomicAdd(cache + (int64_t)key * 4 + (e & 3), s_val[e]);
            }
        } else {
            for (int s_ = threadIdx.x; s_ < kUdSlots; s_ += kUdBlock) {
                const int key = s_key[s_];
                if (key >= 0) unsafeAtomicAdd(cache + (int64_t)key * 4 + 3, s_val[s_]);
            }
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int qf_split_layout(const int64_t *index_ray, int64_t n, int32_t width, int32_t height, int32_t *hit_count,
                               int64_t *ray_offset, int64_t *tile_base, int32_t *invalid, int32_t *order,
                               int32_t *inverse, void *stream)
{
    if (n < 0 || n >= 0x7fffffff || width < 1 || height < 1) return QF_ERR_INVALID_ARGUMENT;
    const int64_t n_rays = (int64_t)width * height;
    if (n_rays >= 0x7fffffff) return QF_ERR_INVALID_ARGUMENT;
    if (!hit_count || !ray_offset || !tile_base || !invalid || !inverse || (n > 0 && !index_ray))
        return QF_ERR_INVALID_ARGUMENT;
    hipStream_t st = qf_stream(stream);
    QF_HIP_TRY(hipMemsetAsync(invalid, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(split_rays_kernel, dim3(qf_grid_1d(n_rays + 1, 256)), dim3(256), 0, st, index_ray, n, n_rays,
                       hit_count, ray_offset, invalid);
    QF_LAUNCH_CHECK();
    // tile bases: exclusive scan of the 8x8-tile totals (the grand total lands in ray_offset[n_rays], where it already is)
    int rc = qf_tile_offsets(hit_count, 0x7fffffff, width, height, tile_base, ray_offset + n_rays, nullptr, nullptr, nullptr,
                             nullptr, stream);
    if (rc != QF_OK) return rc;
    const int tiles_x = (width + 7) / 8, tiles_y = (height + 7) / 8;
    hipLaunchKernelGGL(split_order_kernel, dim3(tiles_x * tiles_y), dim3(64), 0, st, hit_count, ray_offset, tile_base,
                       (int)width, (int)height, tiles_x, n, invalid, order, inverse);
    QF_LAUNCH_CHECK();
    return QF_OK;
}

extern "C" int qf_mesh_update_d(const float *d, const float *w, const int64_t *index_tri, int64_t n, int64_t n_faces,
                                float *cache, int32_t *skipped, void *stream)
{
    if (n < 0 || n_faces < 0) return QF_ERR_INVALID_ARGUMENT;
    if (n == 0) return QF_OK;
    if (!w || !index_tri || !cache) return QF_ERR_INVALID_ARGUMENT;
    if (n >= 8 * kUdChunk && n_faces < 0x7fffffff) {     // enough for the merge to pay (a training batch, a frame's window)
        const int64_t chunks = qf_div_up(n, kUdChunk);
        const int64_t cap = (int64_t)qf_cu_count_cached() * 8;
        const unsigned grid = (unsigned)(chunks < cap ? chunks : cap);
        if (d)
            hipLaunchKernelGGL(mesh_update_d_merged_kernel<true>, dim3(grid), dim3(kUdBlock), 0, qf_stream(stream), d, w,
                               index_tri, n, n_faces, cache, skipped);
        else
            hipLaunchKernelGGL(mesh_update_d_merged_kernel<false>, dim3(grid), dim3(kUdBlock), 0, qf_stream(stream), d, w,
                               index_tri, n, n_faces, cache, skipped);
        QF_LAUNCH_CHECK();
        return QF_OK;
    }
    hipLaunchKernelGGL(mesh_update_d_kernel, dim3(qf_grid_1d(d ? 4 * n : n, 256)), dim3(256), 0, qf_stream(stream), d, w,
                       index_tri, n, n_faces, cache, skipped);
    QF_LAUNCH_CHECK();
    return QF_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Backward of derive_properties (training side): gradients of the per-ray outputs
// (rgb, alpha, depth) w.r.t. the per-sample colour, density and depth.  With C = sum w c, A = sum w, D = sum w t:
//   white: rgb = (1-A) + A C ; black: rgb = A C ; custom: rgb = A C + (1-A) bg
//   gC = A g_rgb ; gA = g_alpha + sum_ch g_rgb (C - {1, 0, bg}) ; gw_i = gC.c_i + gA + gD t_i
//   w_i = T_i (1 - exp(-tau_i)), dw_i/dtau_i = T_i exp(-tau_i), dw_k/dtau_i = -w_k for k > i
//   => gtau_i = gw_i T_i exp(-tau_i) - sum_{k>i} gw_k w_k ; gsigma_i = gtau_i delta_i ; gc_i = w_i gC ; gt_i = w_i gD
namespace {

__global__ void derive_properties_backward_kernel(const float *rgb_s, const float *sigma, const float *depth_s,
                                                  const float *deltas, float delta_const, const int64_t *index_ray,
                                                  int64_t n, int64_t n_rays, int bg_mode, const float *bkgd, const float *g_rgb,
                                                  const float *g_alpha, const float *g_depth, float *grad_rgb_s,
                                                  float *grad_sigma, float *grad_depth_s)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t ray = index_ray[i];
        if (i != 0 && index_ray[i - 1] == ray) continue;
        // forward pass over the ray: totals and the index just past its last sample
        float cum = 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f, ca = 0.0f;
        int64_t end = i;
        for (; end < n && index_ray[end] == ray; ++end) {
            const float tau = sigma[end] * (deltas ? deltas[end] : delta_const);
            const float w = expf(-cum) * (1.0f - expf(-tau));
            cum += tau;
            cr += w * rgb_s[end * 3 + 0];
            cg += w * rgb_s[end * 3 + 1];
            cb += w * rgb_s[end * 3 + 2];
            ca += w;
        }
        const bool in_image = ray >= 0 && ray < n_rays;      // a ray id outside the image has no output: zero gradient
        const float gr = in_image ? g_rgb[ray * 3 + 0] : 0.0f, gg = in_image ? g_rgb[ray * 3 + 1] : 0.0f;
        const float gb = in_image ? g_rgb[ray * 3 + 2] : 0.0f;
        const float gD = (g_depth && in_image) ? g_depth[ray] : 0.0f;
        const bool plain = bg_mode == QF_BG_NONE;
        const float b0 = bg_mode == QF_BG_WHITE ? 1.0f : (bg_mode == QF_BG_CUSTOM ? bkgd[0] : 0.0f);
        const float b1 = bg_mode == QF_BG_WHITE ? 1.0f : (bg_mode == QF_BG_CUSTOM ? bkgd[1] : 0.0f);
        const float b2 = bg_mode == QF_BG_WHITE ? 1.0f : (bg_mode == QF_BG_CUSTOM ? bkgd[2] : 0.0f);
        const float gA = ((g_alpha && in_image) ? g_alpha[ray] : 0.0f) +
                         (plain ? 0.0f : gr * (cr - b0) + gg * (cg - b1) + gb * (cb - b2));
        const float gCr = plain ? gr : ca * gr, gCg = plain ? gg : ca * gg, gCb = plain ? gb : ca * gb;
        // backward sweep: cum holds the total optical depth; peel samples off the far end
        float suffix = 0.0f;
        for (int64_t j = end - 1; j >= i; --j) {
            const float dl = deltas ? deltas[j] : delta_const;
            const float tau = sigma[j] * dl;
            cum -= tau;                                   // exclusive optical depth of sample j (up to rounding)
            const float T = expf(-cum), e = expf(-tau);
            const float w = T * (1.0f - e);
            const float gw = gCr * rgb_s[j * 3 + 0] + gCg * rgb_s[j * 3 + 1] + gCb * rgb_s[j * 3 + 2] + gA + gD * depth_s[j];
            grad_sigma[j] = (gw * T * e - suffix) * dl;
            suffix += gw * w;
            grad_rgb_s[j * 3 + 0] = w * gCr;
            grad_rgb_s[j * 3 + 1] = w * gCg;
            grad_rgb_s[j * 3 + 2] = w * gCb;
            if (grad_depth_s) grad_depth_s[j] = w * gD;
        }
    }
}

}  // namespace

extern "C" int qf_derive_properties_backward(const float *rgb_s, const float *sigma, const float *depth,
                                             const float *deltas, float delta_const, const int64_t *index_ray,
                                             int64_t n, int64_t n_rays, int32_t bg_mode, const float *bkgd, const float *g_rgb,
                                             const float *g_alpha, const float *g_depth, float *grad_rgb_s,
                                             float *grad_sigma, float *grad_depth, void *stream)
{
    if (n < 0 || n_rays < 0 || bg_mode < 0 || bg_mode > 3 || (bg_mode == QF_BG_CUSTOM && !bkgd)) return QF_ERR_INVALID_ARGUMENT;
    if (n == 0) return QF_OK;
    if (n_rays == 0) return QF_ERR_INVALID_ARGUMENT;
    if (!rgb_s || !sigma || !depth || !index_ray || !g_rgb || !grad_rgb_s || !grad_sigma) return QF_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(derive_properties_backward_kernel, dim3(qf_grid_1d(n, 256)), dim3(256), 0, qf_stream(stream),
                       rgb_s, sigma, depth, deltas, delta_const, index_ray, n, n_rays, (int)bg_mode, bkgd, g_rgb, g_alpha,
                       g_depth, grad_rgb_s, grad_sigma, grad_depth);
    QF_LAUNCH_CHECK();
    return QF_OK;
}
