// FETCH_SIZE calibration for the field kernel's access pattern (MI355X_MICROARCH.md "HBM": calibrate on a known byte
// count in your own access pattern).  Each lane reads ONE float2 (8 B) from a distinct 128-B-aligned line of a table
// far larger than the Infinity Cache, lines visited in a pseudo-random order.  Known: n_loads distinct lines.
//   hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o /tmp/calib_fetch && rocprofv3 --pmc FETCH_SIZE ... -- /tmp/calib_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void gather8(const float2 *table, size_t n_lines, size_t n_loads, unsigned long long mul, int both_halves, float *out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_loads; i += (size_t)gridDim.x * blockDim.x) {
        if (both_halves == 2) {      // lane pairs: lanes 2k / 2k+1 read the two 64-B halves of ONE line in the SAME instruction
            const size_t line2 = ((i >> 1) * mul) % n_lines;
            const float2 v2 = table[line2 * 16 + (i & 1) * 8];
            acc += v2.x + v2.y;
            continue;
        }
        const size_t line = (i * mul) % n_lines;          // mul odd and coprime with n_lines (power of two): a permutation
        const float2 v = table[line * 16];                // 16 float2 = 128 B per line
        acc += v.x + v.y;
        if (both_halves == 1) { const float2 w = table[line * 16 + 8]; acc += w.x + w.y; }   // +64 B: other half of the line
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main(int argc, char **argv)
{
    const int log2_lines = argc > 2 ? atoi(argv[2]) : 24;  // 24: 2 GiB table (HBM); 19: 64 MiB (Infinity-Cache resident)
    const int sweeps = argc > 3 ? atoi(argv[3]) : 1;
    const size_t n_lines = (size_t)1 << log2_lines;
    const int both = argc > 1 ? atoi(argv[1]) : 0;
    const size_t n_loads = n_lines * sweeps * (both == 2 ? 2 : 1);   // every line exactly once per sweep (mode 2: two lanes per line)
    float2 *table; float *out;
    hipMalloc(&table, n_lines * 128);
    hipMalloc(&out, 4);
    hipMemset(table, 0, n_lines * 128);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(gather8, dim3(256 * 7 + 3), dim3(256), 0, 0, table, n_lines, n_loads, 2654435761ull, both, out);
        hipDeviceSynchronize();
    }
    printf("n_loads=%zu distinct 128-B lines=%zu bytes_if_64B_requests=%zu bytes_if_128B=%zu\n", n_loads, n_lines,
           n_loads * 64, n_loads * 128);
    return 0;
}
