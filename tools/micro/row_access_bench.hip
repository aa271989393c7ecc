// Micro-benchmark (diagnostic, not part of the library): what does the MFMA-layout access pattern of the row kernels cost against
// memory-order access?  A wave reads (and writes) 32 rows x 80 floats either as k_rows_linear_x3 does -- lane (g, j) takes 32 bytes
// of row j at channel 32 s + 8 g: per instruction 16 rows x 64-byte segments at a 320-byte stride -- or flat (1 KB runs).
//   hipcc --offload-arch=gfx950 -O3 row_access_bench.hip -o row_access_bench && ./row_access_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));

// MODE 0: MFMA layout, MODE 1: flat.  Each wave handles 32 rows of C = 80 floats: 2560 floats = 640 v4f = 10 per lane.
template <int MODE>
__global__ void __launch_bounds__(256) k_copy(const float* __restrict__ x, float* __restrict__ y, long long rows) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const long long r0 = ((long long)blockIdx.x * 4 + wave) * 32;
    if (r0 >= rows) return;
    v4f v[12];
    if (MODE == 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int c = 32 * s + 8 * g;
                const float* p = x + ((r0 + 16 * t + j) * 80 + (c < 80 ? c : 0));
                v[(t * 3 + s) * 2] = *(const v4f*)p;
                v[(t * 3 + s) * 2 + 1] = *(const v4f*)(p + 4);
            }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int c = 32 * s + 8 * g;
                if (c < 80) {
                    float* p = y + ((r0 + 16 * t + j) * 80 + c);
                    *(v4f*)p = v[(t * 3 + s) * 2] * 2.f;
                    *(v4f*)(p + 4) = v[(t * 3 + s) * 2 + 1] * 2.f;
                }
            }
    } else {
#pragma unroll
        for (int u = 0; u < 10; ++u) v[u] = *(const v4f*)(x + r0 * 80 + (u * 64 + lane) * 4);
#pragma unroll
        for (int u = 0; u < 10; ++u) *(v4f*)(y + r0 * 80 + (u * 64 + lane) * 4) = v[u] * 2.f;
    }
}

int main() {
    const long long rows = 160000 * 8;          // 410 MB in, 410 MB out: beyond the 256 MB memory-side cache
    float *x, *y;
    hipMalloc(&x, rows * 80 * 4); hipMalloc(&y, rows * 80 * 4);
    hipMemset(x, 0, rows * 80 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            const int grid = (int)((rows / 32 + 3) / 4);
            hipEventRecord(a);
            for (int i = 0; i < 10; ++i) {
                if (mode == 0) hipLaunchKernelGGL(k_copy<0>, dim3(grid), dim3(256), 0, 0, x, y, rows);
                else hipLaunchKernelGGL(k_copy<1>, dim3(grid), dim3(256), 0, 0, x, y, rows);
            }
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("mode %d (%s): %.1f us per launch, %.2f TB/s read + write\n", mode, mode ? "flat" : "MFMA layout", ms * 100.0, 2.0 * rows * 80 * 4 / (ms / 10 * 1e-3) / 1e12);
        }
    // the library's size: 160 000 rows (51 MB in, 51 MB out: cache resident)
    for (int mode = 0; mode < 2; ++mode) {
        const long long r2 = 160000; const int grid = (int)((r2 / 32 + 3) / 4);
        hipEventRecord(a);
        for (int i = 0; i < 20; ++i) {
            if (mode == 0) hipLaunchKernelGGL(k_copy<0>, dim3(grid), dim3(256), 0, 0, x, y, r2);
            else hipLaunchKernelGGL(k_copy<1>, dim3(grid), dim3(256), 0, 0, x, y, r2);
        }
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("160000 rows, mode %d: %.1f us per launch, %.2f TB/s\n", mode, ms * 50.0, 2.0 * r2 * 80 * 4 / (ms / 20 * 1e-3) / 1e12);
    }
    return 0;
}
