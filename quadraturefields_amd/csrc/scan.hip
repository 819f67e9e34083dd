// Sample offsets of the packed quadrature points: ray_offset[r] = sum_{q<r} min(hit_count[q], max_hits), with the
// grand total in ray_offset[n_rays].  The total lands in device memory, so the caller can read it back while the pack
// kernel is already running.  Hand-written three-launch scan (below); round 1's rocPRIM/hipCUB lookback scan behind
// qf_sample_offsets is gone -- the library is no longer linked on any product path.
#include "qf_common.h"

// ---------------------------------------------------------------------------------------------------------------------
// One frame's offsets in three small launches: the per-ray sample offsets (as qf_sample_offsets) AND, for an image-shaped
// batch, the exclusive scan of the 8x8-tile totals that qf_coherent_layout needs -- round 1 issued a library scan (two
// kernels), qf_tile_totals, torch.cumsum (two kernels) and a subtraction for the same numbers.
//   1. frame_partials_kernel: workgroup b < n_blocks sums min(count, K) over its 1024 rays; the workgroups after those
//      compute the tile totals (one wave per tile, four tiles per workgroup);
//   2. frame_scan_kernel: ONE workgroup turns both arrays into exclusive prefix sums in place (a few thousand
//      elements) and writes the grand total to ray_offset[n_rays];
//   3. frame_ray_offsets_kernel: every workgroup scans its 1024 rays from its base.
namespace {

constexpr int kFoRays = 1024;          // rays per workgroup in steps 1 and 3
constexpr int kFoThreads = 256;

__device__ __forceinline__ int64_t clamped(const int32_t *hit_count, int64_t r, int64_t n, int32_t cap)
{
    if (r >= n) return 0;
    const int32_t c = hit_count[r];
    return c < 0 ? 0 : (c < cap ? c : cap);
}

// block-wide inclusive scan of one int64 per thread (kFoThreads threads); returns the inclusive value, *total = block sum
__device__ __forceinline__ int64_t block_inclusive_scan(int64_t v, int64_t *s_wave, int64_t *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int64_t o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    if (lane == 63) s_wave[wave] = v;
    __syncthreads();
    int64_t base = 0, sum = 0;
    for (int w = 0; w < kFoThreads / 64; ++w) {
        if (w < wave) base += s_wave[w];
        sum += s_wave[w];
    }
    __syncthreads();
    *total = sum;
    return v + base;
}

__global__ __launch_bounds__(kFoThreads) void frame_partials_kernel(const int32_t *__restrict__ hit_count, int64_t n_rays,
                                                                    int32_t cap, int n_blocks, int w, int h, int tiles_x,
                                                                    int n_tiles, int64_t *__restrict__ partial,
                                                                    int64_t *__restrict__ tile_total, int band_rows)
{
    __shared__ int64_t s_wave[kFoThreads / 64];
    if ((int)blockIdx.x < n_blocks) {
        const int64_t r0 = (int64_t)blockIdx.x * kFoRays;
        int64_t v = 0;
#pragma unroll
        for (int k = 0; k < kFoRays / kFoThreads; ++k) v += clamped(hit_count, r0 + threadIdx.x + (int64_t)k * kFoThreads, n_rays, cap);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            int64_t s = 0;
            for (int q = 0; q < kFoThreads / 64; ++q) s += s_wave[q];
            partial[blockIdx.x] = s;
        }
        return;
    }
    const int tile = ((int)blockIdx.x - n_blocks) * (kFoThreads / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (tile >= n_tiles) return;
    int64_t ray = 0;
    int64_t cnt = qf_tile_lane_ray(tile, lane, w, h, tiles_x, band_rows, &ray) ? clamped(hit_count, ray, n_rays, cap) : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (lane == 0) tile_total[tile] = cnt;
}

// one workgroup; every thread owns a CONTIGUOUS run of each array, so an array is scanned with one block scan
__device__ __forceinline__ void frame_scan_body(int64_t *partial, int n_blocks, int64_t *tile_total, int n_tiles,
                                                int64_t *grand_total, const int32_t *overflow_in, int64_t *host_out,
                                                int64_t *s_wave, const int32_t *ray_flag_in)
{
    for (int pass = 0; pass < 2; ++pass) {
        int64_t *a = pass == 0 ? partial : tile_total;
        const int m = pass == 0 ? n_blocks : n_tiles;
        if (!a) continue;
        const int per = (m + kFoThreads - 1) / kFoThreads;
        const int i0 = threadIdx.x * per, i1 = i0 + per < m ? i0 + per : m;
        int64_t sum = 0;
#pragma unroll 8
        for (int i = i0; i < i1; ++i) sum += a[i];
        int64_t total;
        int64_t run = block_inclusive_scan(sum, s_wave, &total) - sum;
#pragma unroll 8
        for (int i = i0; i < i1; ++i) {
            const int64_t v = a[i];
            a[i] = run;
            run += v;
        }
        // the grand total comes from the per-ray partials when they exist, from the tile totals otherwise
        if (threadIdx.x == 0 && (pass == 0 || !partial)) {
            *grand_total = total;
            if (host_out) {                 // pinned host memory: the frame's 16-byte readback without a copy kernel
                host_out[0] = total;
                host_out[1] = overflow_in ? (int64_t)*overflow_in : 0;
                // the camera-coherent route's verdict on the rays (camera_rays_check): policy only, like the overflow count
                host_out[3] = ray_flag_in ? (int64_t)*ray_flag_in : 0;
            }
        }
    }
}

// zero_word (or NULL): a device int32 this launch zeroes for a later kernel of the frame -- the dropped-hit counter of
// qf_pack_tiles, which then needs neither a memset launch nor a publishing launch of its own.
// (A "last workgroup does the tail" fusion of the two launches of qf_tile_offsets, and of the tile pack with its
// publishing launch, was measured and dropped: one same-address ticket atomic per workgroup costs ~37 ns, i.e. 0.47 ms
// for the 12 500 workgroups of an 800x800 frame.)
__global__ __launch_bounds__(kFoThreads) void frame_scan_kernel(int64_t *partial, int n_blocks, int64_t *tile_total, int n_tiles,
                                                                int64_t *grand_total, const int32_t *overflow_in,
                                                                int64_t *host_out, int32_t *zero_word,
                                                                const int32_t *ray_flag_in)
{
    __shared__ int64_t s_wave[kFoThreads / 64];
    if (zero_word && threadIdx.x == 0) *zero_word = 0;
    frame_scan_body(partial, n_blocks, tile_total, n_tiles, grand_total, overflow_in, host_out, s_wave, ray_flag_in);
}

__global__ __launch_bounds__(kFoThreads) void frame_ray_offsets_kernel(const int32_t *__restrict__ hit_count, int64_t n_rays,
                                                                       int32_t cap, const int64_t *__restrict__ partial,
                                                                       int64_t *__restrict__ ray_offset)
{
    __shared__ int64_t s_wave[kFoThreads / 64];
    const int64_t r0 = (int64_t)blockIdx.x * kFoRays + (int64_t)threadIdx.x * (kFoRays / kFoThreads);
    int64_t c[kFoRays / kFoThreads], v = 0;
#pragma unroll
    for (int k = 0; k < kFoRays / kFoThreads; ++k) { c[k] = clamped(hit_count, r0 + k, n_rays, cap); v += c[k]; }
    int64_t total;
    int64_t run = partial[blockIdx.x] + block_inclusive_scan(v, s_wave, &total) - v;
#pragma unroll
    for (int k = 0; k < kFoRays / kFoThreads; ++k) {
        if (r0 + k < n_rays) ray_offset[r0 + k] = run;
        run += c[k];
    }
}

}  // namespace

extern "C" int64_t qf_banded_tile_count(int32_t width, int32_t height, int32_t band_rows)
{
    if (width < 1 || height < 1 || band_rows < 0) return -1;
    return qf_banded_tiles(width, height, band_rows);
}

extern "C" int64_t qf_frame_offsets_temp_bytes(int64_t n_rays)
{
    if (n_rays < 0) return -1;
    return (qf_div_up(n_rays, kFoRays) + 1) * (int64_t)sizeof(int64_t);
}

extern "C" int qf_frame_offsets(const int32_t *hit_count, int64_t n_rays, int32_t max_hits, int32_t width, int32_t height,
                                int64_t *ray_offset, int64_t *tile_base, void *temp, int64_t temp_bytes,
                                const int32_t *overflow_in, const int32_t *ray_flag_in, int64_t *host_out, int32_t band_rows,
                                void *stream)
{
    if (n_rays < 0 || n_rays >= 0x7fffffff || max_hits < 1 || width < 0 || height < 0 || band_rows < 0)
        return QF_ERR_INVALID_ARGUMENT;
    if (!ray_offset || !temp || (n_rays > 0 && !hit_count)) return QF_ERR_INVALID_ARGUMENT;
    if (temp_bytes < qf_frame_offsets_temp_bytes(n_rays)) return QF_ERR_INVALID_ARGUMENT;
    const bool tiles = tile_base != nullptr;
    if (tiles && (width < 1 || height < 1 || (int64_t)width * height != n_rays)) return QF_ERR_INVALID_ARGUMENT;
    hipStream_t st = qf_stream(stream);
    const int n_blocks = (int)qf_div_up(n_rays, kFoRays);
    const int tiles_x = tiles ? (width + 7) / 8 : 0;
    const int n_tiles = tiles ? (int)qf_banded_tiles(width, height, band_rows) : 0;
    int64_t *partial = reinterpret_cast<int64_t *>(temp);
    const int tile_blocks = (int)qf_div_up(n_tiles, kFoThreads / 64);
    if (n_blocks + tile_blocks > 0) {
        hipLaunchKernelGGL(frame_partials_kernel, dim3((unsigned)(n_blocks + tile_blocks)), dim3(kFoThreads), 0, st, hit_count,
                           n_rays, max_hits, n_blocks, (int)width, (int)height, tiles_x, n_tiles, partial, tile_base,
                           (int)band_rows);
        QF_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(frame_scan_kernel, dim3(1), dim3(kFoThreads), 0, st, partial, n_blocks, tile_base, n_tiles,
                       ray_offset + n_rays, overflow_in, host_out, (int32_t *)nullptr, ray_flag_in);
    QF_LAUNCH_CHECK();
    if (n_blocks > 0) {
        hipLaunchKernelGGL(frame_ray_offsets_kernel, dim3((unsigned)n_blocks), dim3(kFoThreads), 0, st, hit_count, n_rays,
                           max_hits, partial, ray_offset);
        QF_LAUNCH_CHECK();
    }
    return QF_OK;
}

// The render-only frame's version: only the tile bases and the total (what qf_pack_tiles and qf_composite_tiles take),
// no per-ray offsets -- the tile part of step 1 and a scan over the tile totals alone.
extern "C" int qf_tile_offsets(const int32_t *hit_count, int32_t max_hits, int32_t width, int32_t height,
                               int64_t *tile_base, int64_t *total, const int32_t *overflow_in, const int32_t *ray_flag_in,
                               int64_t *host_out, int32_t *zero_word, void *stream)
{
    if (max_hits < 1 || width < 1 || height < 1 || !hit_count || !tile_base || !total) return QF_ERR_INVALID_ARGUMENT;
    const int64_t n_rays = (int64_t)width * height;
    if (n_rays >= 0x7fffffff) return QF_ERR_INVALID_ARGUMENT;
    hipStream_t st = qf_stream(stream);
    const int tiles_x = (width + 7) / 8;
    const int n_tiles = tiles_x * ((height + 7) / 8);
    const int tile_blocks = (int)qf_div_up(n_tiles, kFoThreads / 64);
    hipLaunchKernelGGL(frame_partials_kernel, dim3((unsigned)tile_blocks), dim3(kFoThreads), 0, st, hit_count, n_rays, max_hits,
                       0, (int)width, (int)height, tiles_x, n_tiles, (int64_t *)nullptr, tile_base, 0);
    hipLaunchKernelGGL(frame_scan_kernel, dim3(1), dim3(kFoThreads), 0, st, (int64_t *)nullptr, 0, tile_base, n_tiles, total,
                       overflow_in, host_out, zero_word, ray_flag_in);
    QF_LAUNCH_CHECK();
    return QF_OK;
}

// qf_sample_offsets = the per-ray half of qf_frame_offsets (no tiles, no host block).
extern "C" int64_t qf_sample_offsets_temp_bytes(int64_t n_rays)
{
    const int64_t b = qf_frame_offsets_temp_bytes(n_rays);
    return b < 0 ? b : (b < 16 ? 16 : b);
}

extern "C" int qf_sample_offsets(const int32_t *hit_count, int64_t n_rays, int32_t max_hits, int64_t *ray_offset,
                                 void *temp, int64_t temp_bytes, void *stream)
{
    if (temp_bytes < qf_sample_offsets_temp_bytes(n_rays)) return QF_ERR_INVALID_ARGUMENT;
    return qf_frame_offsets(hit_count, n_rays, max_hits, 0, 0, ray_offset, nullptr, temp, temp_bytes, nullptr, nullptr,
                            nullptr, 0, stream);
}
