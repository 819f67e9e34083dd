// Host-only entry points: status text, device query, hash-grid level table.
#include <atomic>
#include <cmath>
#include <cstdio>

#include "qf_common.h"

namespace {
std::atomic<int> g_last_hip_error{0};
std::atomic<int> g_cu_count{0};
thread_local char g_msg[160];
}  // namespace

extern "C" void qf_set_last_hip_error(int code) { g_last_hip_error.store(code); }

extern "C" const char *qf_status_string(int status)
{
    switch (status) {
    case QF_OK: return "ok";
    case QF_ERR_INVALID_ARGUMENT: return "invalid argument";
    case QF_ERR_UNSUPPORTED: return "unsupported configuration";
    case QF_ERR_NO_DEVICE: return "no HIP device";
    case QF_ERR_HIP: {
        const int e = g_last_hip_error.load();
        std::snprintf(g_msg, sizeof(g_msg), "HIP error %d: %s", e, hipGetErrorString((hipError_t)e));
        return g_msg;
    }
    default: return "unknown status";
    }
}

// An experiment build (tools/cell_record_experiment.sh) adds an offset, so that the loader refuses it as the product.
#ifndef QF_ABI_VERSION_OFFSET
#define QF_ABI_VERSION_OFFSET 0
#endif
extern "C" int qf_abi_version(void) { return QF_ABI_VERSION + QF_ABI_VERSION_OFFSET; }

int qf_cu_count_cached()
{
    int v = g_cu_count.load();
    if (v > 0) return v;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    v = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    g_cu_count.store(v);
    return v;
}

extern "C" int qf_device_cu_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return QF_ERR_NO_DEVICE;
    return qf_cu_count_cached();
}

// tiny-cuda-nn level rule, evaluated once on the host in fp32 and handed to the
// kernels as a table, so host and device can never disagree on scale / resolution / offsets.
extern "C" int qf_grid_desc_init(qf_grid_desc *d, uint32_t n_levels, uint32_t log2_hashmap_size,
                                 uint32_t base_resolution, double per_level_scale)
{
    if (!d || n_levels < 1 || n_levels > QF_MAX_LEVELS || log2_hashmap_size < 1 || log2_hashmap_size > 30 ||
        base_resolution < 1 || !(per_level_scale > 0.0))
        return QF_ERR_INVALID_ARGUMENT;
    d->n_levels = n_levels;
    d->n_features = 2;
    d->log2_hashmap_size = log2_hashmap_size;
    d->base_resolution = base_resolution;
    d->per_level_scale = (float)per_level_scale;
    d->hashed_mask = 0;
    // each fp32 step of tcnn's chain exp2f(l * log2f(b)) * N - 1 is evaluated in double and rounded once,
    // i.e. correctly rounded and therefore reproducible across libm implementations
    const float log2_b = (float)std::log2((double)(float)per_level_scale);
    uint32_t offset = 0;
    for (uint32_t l = 0; l < QF_MAX_LEVELS; ++l) {
        if (l >= n_levels) { d->offset[l + 1] = offset; d->resolution[l] = 0; d->scale[l] = 0.f; continue; }
        const float arg = (float)l * log2_b;
        const float scale = (float)std::exp2((double)arg) * (float)base_resolution - 1.0f;
        const uint32_t res = (uint32_t)std::ceil(scale) + 1u;
        const uint32_t max_params = 0xFFFFFFFFu / 2u;
        const double cube = (double)res * res * res;
        uint32_t rows = cube > (double)max_params ? max_params : (uint32_t)cube;
        rows = (rows + 7u) / 8u * 8u;
        const uint32_t cap = 1u << log2_hashmap_size;
        if (rows > cap) rows = cap;
        uint32_t stride = 1;
        for (int k = 0; k < 3; ++k)
            if (stride <= rows) stride *= res;   // uint32 wrap, as tcnn
        if (rows < stride) d->hashed_mask |= 1u << l;
        d->offset[l] = offset;
        d->resolution[l] = res;
        d->scale[l] = scale;
        if ((uint64_t)offset + rows > 0xFFFFFFFFull) return QF_ERR_UNSUPPORTED;
        offset += rows;
        d->offset[l + 1] = offset;
    }
    return QF_OK;
}
