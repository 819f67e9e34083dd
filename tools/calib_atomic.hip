// Returning integer atomics (slot allocation of the camera-coherent intersector: slot = atomicAdd(&hit_count[ray], 1)):
// agent scope (performed at the memory side, coherent across the 8 XCDs) against workgroup scope (performed in the
// XCD's own L2), on a 2.5 MB counter array where every workgroup only touches counters of its own XCD's stripe set
// (rows of 8 interleaved by HW_REG_XCC_ID), as raster_xcd_kernel does.
//   hipcc --offload-arch=gfx950 -O3 tools/calib_atomic.hip -o /tmp/calib_atomic && /tmp/calib_atomic
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15; }

template <int SCOPE>
__global__ void bump(int *cnt, int n_cnt, long n_ops, unsigned long long mul, int *out, int *xcc_mask)
{
    const int x = xcc_id();
    if (threadIdx.x == 0) atomicOr(xcc_mask, 1 << x);
    int acc = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_ops; i += (long)gridDim.x * blockDim.x) {
        // a pseudo-random counter of this XCD's share: row = 8 * (8 * k + x) + r  (w = 800)
        const unsigned long long h = (unsigned long long)i * mul;
        const int k = (int)((h >> 20) % 12), r = (int)((h >> 40) & 7), col = (int)((h >> 8) % 800);
        const int row = 8 * (8 * k + x) + r;
        const int idx = (row * 800 + col) % n_cnt;
        acc += __hip_atomic_fetch_add(cnt + idx, 1, __ATOMIC_RELAXED, SCOPE);
    }
    if (acc == -12345) out[0] = acc;
}

int main()
{
    const int n_cnt = 640000;
    const long n_ops = 4000000;
    int *cnt, *out, *mask;
    hipMalloc(&cnt, n_cnt * 4); hipMalloc(&out, 4); hipMalloc(&mask, 4);
    hipMemset(mask, 0, 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int scope = 0; scope < 2; ++scope) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipMemset(cnt, 0, n_cnt * 4);
            hipEventRecord(a);
            if (scope == 0) hipLaunchKernelGGL(bump<__HIP_MEMORY_SCOPE_AGENT>, dim3(2048), dim3(256), 0, 0, cnt, n_cnt, n_ops, 2654435761ull, out, mask);
            else hipLaunchKernelGGL(bump<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(2048), dim3(256), 0, 0, cnt, n_cnt, n_ops, 2654435761ull, out, mask);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        // verify the total
        int *h = (int *)malloc(n_cnt * 4);
        hipMemcpy(h, cnt, n_cnt * 4, hipMemcpyDeviceToHost);
        long sum = 0; for (int i = 0; i < n_cnt; ++i) sum += h[i];
        int m; hipMemcpy(&m, mask, 4, hipMemcpyDeviceToHost);
        printf("%s scope: %.1f us for %ld returning atomics = %.3e /s; sum of counters %ld (expected %ld); xcc mask 0x%x\n",
               scope == 0 ? "agent" : "workgroup", best * 1e3, n_ops, n_ops / (best * 1e-3), sum, n_ops, m);
        free(h);
    }
    return 0;
}
