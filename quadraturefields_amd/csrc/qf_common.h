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
