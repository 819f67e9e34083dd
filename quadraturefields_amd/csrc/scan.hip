// Sample offsets of the packed quadrature points: ray_offset[r] = sum_{q<r} min(hit_count[q], max_hits), with the
// grand total in ray_offset[n_rays].  One device scan (rocPRIM through hipCUB: a plain library primitive) replaces the
// cast / cumsum / subtract / concatenate chain the host side used to issue (six ~5 us launches per frame), and because
// the total lands in device memory the caller can read it back while the pack kernel is already running.
#include <hipcub/hipcub.hpp>

#include "qf_common.h"

namespace {

struct ClampedCount {
    const int32_t *count;
    int64_t n;
    int32_t cap;
    __host__ __device__ __forceinline__ int64_t operator()(const int64_t &i) const
    {
        if (i >= n) return 0;                        // the extra element: its exclusive sum is the grand total
        const int32_t c = count[i];
        return c < 0 ? 0 : (c < cap ? c : cap);
    }
};

using CountIter = hipcub::TransformInputIterator<int64_t, ClampedCount, hipcub::CountingInputIterator<int64_t>>;

inline CountIter make_iter(const int32_t *hit_count, int64_t n_rays, int32_t max_hits)
{
    return CountIter(hipcub::CountingInputIterator<int64_t>(0), ClampedCount{hit_count, n_rays, max_hits});
}

}  // namespace

extern "C" int64_t qf_sample_offsets_temp_bytes(int64_t n_rays)
{
    if (n_rays < 0) return -1;
    size_t bytes = 0;
    int64_t *out = nullptr;
    if (hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, make_iter(nullptr, n_rays, 1), out, (int)(n_rays + 1), nullptr) !=
        hipSuccess)
        return -1;
    return (int64_t)(bytes < 16 ? 16 : bytes);
}

extern "C" int qf_sample_offsets(const int32_t *hit_count, int64_t n_rays, int32_t max_hits, int64_t *ray_offset,
                                 void *temp, int64_t temp_bytes, void *stream)
{
    if (n_rays < 0 || n_rays >= 0x7fffffff || max_hits < 1) return QF_ERR_INVALID_ARGUMENT;
    if (!ray_offset || !temp || (n_rays > 0 && !hit_count)) return QF_ERR_INVALID_ARGUMENT;
    if (temp_bytes < qf_sample_offsets_temp_bytes(n_rays)) return QF_ERR_INVALID_ARGUMENT;
    size_t bytes = (size_t)temp_bytes;
    QF_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(temp, bytes, make_iter(hit_count, n_rays, max_hits), ray_offset,
                                               (int)(n_rays + 1), qf_stream(stream)));
    return QF_OK;
}
