// Shared host/device helpers for libqf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "qf_hip.h"

#define QF_HIP_TRY(expr)                       \
    do {                                       \
        hipError_t _e = (expr);                \
        if (_e != hipSuccess) {                \
            qf_set_last_hip_error((int)_e);    \
            return QF_ERR_HIP;                 \
        }                                      \
    } while (0)

#define QF_LAUNCH_CHECK()                      \
    do {                                       \
        hipError_t _e = hipGetLastError();     \
        if (_e != hipSuccess) {                \
            qf_set_last_hip_error((int)_e);    \
            return QF_ERR_HIP;                 \
        }                                      \
    } while (0)

extern "C" void qf_set_last_hip_error(int code);

static inline hipStream_t qf_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Number of CUs of the current device, cached per process (host).
int qf_cu_count_cached();

static inline int64_t qf_div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Grid size for a grid-stride elementwise kernel: enough blocks to fill the chip, capped.
static inline int qf_grid_1d(int64_t n, int block, int blocks_per_cu = 8)
{
    int64_t want = qf_div_up(n, block);
    int64_t cap = (int64_t)qf_cu_count_cached() * blocks_per_cu;
    if (want < 1) want = 1;
    return (int)(want < cap ? want : cap);
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) for a kernel that wants more than 48 KB of dynamic LDS: the attribute
// is PER DEVICE, so one flag per device ordinal (atomic: the entry points may be called from several host threads); a
// process that launches on a second GPU sets it there too instead of failing the launch.
struct QfLdsAttr {
    std::atomic<bool> set[64];
};
static inline hipError_t qf_ensure_dynamic_lds(QfLdsAttr &a, const void *fn, size_t bytes)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const bool tracked = dev >= 0 && dev < 64;
    if (tracked && a.set[dev].load(std::memory_order_acquire)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && tracked) a.set[dev].store(true, std::memory_order_release);
    return e;
}

// The 8x8-pixel tiles of the coherent processing order (qf_frame_offsets / qf_coherent_layout), optionally cut into ROW
// BANDS of band_rows rows (0 = the whole image is one band): the tile grid restarts at the first row of every band, so
// that a band's samples are ONE contiguous run of the order -- the band is the 160 000-ray window of the reference's eval
// loop (generate_splits, train_finetune.py:419-439), whatever its height modulo 8.  Tile ids ascend band by band; a
// band's last tile row is partial when band_rows is not a multiple of 8.
static inline int64_t qf_banded_tiles(int32_t width, int32_t height, int32_t band_rows)
{
    const int64_t tiles_x = (width + 7) / 8;
    if (band_rows <= 0 || band_rows >= height) return tiles_x * ((height + 7) / 8);
    const int64_t full = height / band_rows, rest = height - full * band_rows;
    return tiles_x * (full * ((band_rows + 7) / 8) + (rest + 7) / 8);
}

// lane (0..63) of `tile` -> its ray; 0 when the pixel lies outside the image or beyond its band's last row
__device__ __forceinline__ int qf_tile_lane_ray(int tile, int lane, int w, int h, int tiles_x, int band_rows, int64_t *ray)
{
    const int px = (tile % tiles_x) * 8 + (lane & 7);
    int py;
    if (band_rows > 0 && band_rows < h) {
        const int per_band = tiles_x * ((band_rows + 7) / 8);
        const int band = tile / per_band, t = tile - band * per_band;
        const int local = (t / tiles_x) * 8 + (lane >> 3);
        if (local >= band_rows) return 0;
        py = band * band_rows + local;
    } else {
        py = (tile / tiles_x) * 8 + (lane >> 3);
    }
    if (px >= w || py >= h) return 0;
    *ray = (int64_t)py * w + px;
    return 1;
}
