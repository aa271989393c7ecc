// Micro-benchmark (diagnostic, not part of the library): which zero-fill kernel shape reaches the
// speed of torch's fill on MI355X?  hipcc --offload-arch=gfx950 -O3 fill_bench.hip -o fill_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int ST> __device__ __forceinline__ void st4(float* p, v4f v) {
    if constexpr (ST == 1) __builtin_nontemporal_store(v, (v4f*)p);
    else if constexpr (ST == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" : : "v"(p), "v"(v) : "memory");
    else *(v4f*)p = v;
}
// one-shot: each thread stores U consecutive-by-wave float4 (block covers U*NT*16 bytes contiguous)
template <int NT, int U, int ST>
__global__ void __launch_bounds__(NT) k_fill_oneshot(float* out, long long n4) {
    v4f z = {0.f, 0.f, 0.f, 0.f};
    long long base = (long long)blockIdx.x * (NT * U) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; ++u) { long long i = base + (long long)u * NT; if (i < n4) st4<ST>(out + 4 * i, z); }
}
// grid-stride persistent
template <int NT, int ST>
__global__ void __launch_bounds__(NT) k_fill_stride(float* out, long long n4) {
    v4f z = {0.f, 0.f, 0.f, 0.f};
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long long)gridDim.x * NT) st4<ST>(out + 4 * i, z);
}
// one-shot with a dependent scalar load first (like reading tile metadata) and an LDS allocation
template <int NT, int U, int ST>
__global__ void __launch_bounds__(NT) k_fill_meta(float* out, long long n4, const int* meta) {
    extern __shared__ float lds[];
    v4f z = {0.f, 0.f, 0.f, 0.f};
    if (meta[2 * blockIdx.x] == 12345) { lds[threadIdx.x] = 1.f; return; }
    long long base = (long long)blockIdx.x * (NT * U) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; ++u) { long long i = base + (long long)u * NT; if (i < n4) st4<ST>(out + 4 * i, z); }
}
// wave-contiguous: each wave owns a contiguous U KiB span (its U stores are adjacent 1-KiB pieces)
template <int NT, int U, int ST>
__global__ void __launch_bounds__(NT) k_fill_wavecontig(float* out, long long n4) {
    v4f z = {0.f, 0.f, 0.f, 0.f};
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    long long base = (long long)blockIdx.x * (NT * U) + (long long)wave * (64 * U) + lane;
#pragma unroll
    for (int u = 0; u < U; ++u) { long long i = base + u * 64; if (i < n4) st4<ST>(out + 4 * i, z); }
}
// thread-contiguous: each thread owns U consecutive float4 (64*U contiguous bytes per lane)
template <int NT, int U, int ST>
__global__ void __launch_bounds__(NT) k_fill_threadcontig(float* out, long long n4) {
    v4f z = {0.f, 0.f, 0.f, 0.f};
    long long base = ((long long)blockIdx.x * NT + threadIdx.x) * U;
#pragma unroll
    for (int u = 0; u < U; ++u) { long long i = base + u; if (i < n4) st4<ST>(out + 4 * i, z); }
}
// u stores per thread, but separated by a barrier (waves release their stores in lock-step rounds)
template <int NT, int U, int ST>
__global__ void __launch_bounds__(NT) k_fill_rounds(float* out, long long n4) {
    v4f z = {0.f, 0.f, 0.f, 0.f};
    long long base = (long long)blockIdx.x * (NT * U) + threadIdx.x;
    for (int u = 0; u < U; ++u) { long long i = base + (long long)u * NT; if (i < n4) st4<ST>(out + 4 * i, z); __syncthreads(); }
}
// XCD/stack affinity test: block b (assumed on XCD b%8) writes U 4-KiB chunks whose index is == (b + shift) mod 8
template <int U, int ST>
__global__ void __launch_bounds__(256) k_fill_affine(float* out, long long n4, int shift, int modulus) {
    v4f z = {0.f, 0.f, 0.f, 0.f};
    const long long b = blockIdx.x;
    const long long x = (b + shift) % modulus, j = b / modulus;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long long chunk = (j * U + u) * modulus + x;        // chunk % modulus == x
        const long long i = chunk * 256 + threadIdx.x;           // 256 float4 = 4 KiB per chunk
        if (i < n4) st4<ST>(out + 4 * i, z);
    }
}
template <class F> float timeit(F f, int iters = 10) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> ts;
    for (int i = 0; i < iters + 2; ++i) {
        hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (i >= 2) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end()); return ts[ts.size() / 2];
}
#include <algorithm>
int main() {
    const long long bytes = 16ll * 80 * 16 * 200 * 200 * 4;   // BL2 B=16 output
    const long long n4 = bytes / 16;
    float* out; hipMalloc(&out, bytes);
    int* meta; hipMalloc(&meta, 16 * 1000000); hipMemset(meta, 0, 16 * 1000000);
    auto rep = [&](const char* name, float ms) { printf("{\"variant\": \"%s\", \"ms\": %.4f, \"GBps\": %.0f}\n", name, ms, bytes / ms / 1e6); fflush(stdout); };
    rep("hipMemsetAsync", timeit([&] { hipMemsetAsync(out, 0, bytes, 0); }));
#define ONESHOT(NT, U, ST) rep("oneshot_nt" #NT "_u" #U "_st" #ST, timeit([&] { hipLaunchKernelGGL((k_fill_oneshot<NT, U, ST>), dim3((n4 + NT * U - 1) / (NT * U)), dim3(NT), 0, 0, out, n4); }));
    ONESHOT(256, 1, 0) ONESHOT(256, 4, 0) ONESHOT(256, 5, 0) ONESHOT(256, 8, 0) ONESHOT(256, 16, 0)
    ONESHOT(128, 4, 0) ONESHOT(512, 4, 0) ONESHOT(1024, 4, 0) ONESHOT(64, 8, 0)
    ONESHOT(256, 4, 1) ONESHOT(256, 4, 4) ONESHOT(256, 16, 4) ONESHOT(512, 4, 4)
#define STRIDE(NT, ST, G) rep("stride_nt" #NT "_st" #ST "_g" #G, timeit([&] { hipLaunchKernelGGL((k_fill_stride<NT, ST>), dim3(G), dim3(NT), 0, 0, out, n4); }));
    STRIDE(256, 0, 2048) STRIDE(256, 0, 4096) STRIDE(256, 0, 8192) STRIDE(512, 0, 2048) STRIDE(1024, 0, 1024) STRIDE(256, 4, 4096) STRIDE(256, 1, 4096)
#define META(NT, U, ST, L) rep("meta_nt" #NT "_u" #U "_st" #ST "_lds" #L, timeit([&] { hipLaunchKernelGGL((k_fill_meta<NT, U, ST>), dim3((n4 + NT * U - 1) / (NT * U)), dim3(NT), L, 0, out, n4, meta); }));
    META(256, 5, 0, 0) META(256, 5, 0, 24000) META(256, 10, 0, 44000) META(256, 5, 4, 24000)
#define WC(NT, U, ST) rep("wavecontig_nt" #NT "_u" #U "_st" #ST, timeit([&] { hipLaunchKernelGGL((k_fill_wavecontig<NT, U, ST>), dim3((n4 + NT * U - 1) / (NT * U)), dim3(NT), 0, 0, out, n4); }));
    WC(256, 2, 0) WC(256, 4, 0) WC(256, 8, 0) WC(64, 4, 0) WC(64, 16, 0) WC(256, 4, 4)
#define TC(NT, U, ST) rep("threadcontig_nt" #NT "_u" #U "_st" #ST, timeit([&] { hipLaunchKernelGGL((k_fill_threadcontig<NT, U, ST>), dim3((n4 + NT * U - 1) / (NT * U)), dim3(NT), 0, 0, out, n4); }));
    TC(256, 2, 0) TC(256, 4, 0)
#define RD(NT, U, ST) rep("rounds_nt" #NT "_u" #U "_st" #ST, timeit([&] { hipLaunchKernelGGL((k_fill_rounds<NT, U, ST>), dim3((n4 + NT * U - 1) / (NT * U)), dim3(NT), 0, 0, out, n4); }));
    RD(256, 4, 0) RD(256, 8, 0)
    ONESHOT(256, 2, 0) ONESHOT(256, 3, 0) ONESHOT(64, 1, 0) ONESHOT(128, 1, 0) ONESHOT(512, 1, 0) ONESHOT(1024, 1, 0) ONESHOT(256, 1, 1) ONESHOT(256, 1, 4)
    META(256, 1, 0, 0) META(256, 1, 0, 24000) META(256, 1, 4, 0)
#define AFF(U, ST, SH, MOD) rep("affine_u" #U "_st" #ST "_shift" #SH "_mod" #MOD, timeit([&] { hipLaunchKernelGGL((k_fill_affine<U, ST>), dim3((n4 / 256 + U - 1) / U), dim3(256), 0, 0, out, n4, SH, MOD); }));
    AFF(1, 4, 0, 8) AFF(4, 4, 0, 8) AFF(4, 4, 4, 8) AFF(4, 4, 1, 8) AFF(8, 4, 0, 8) AFF(16, 4, 0, 8) AFF(4, 0, 0, 8) AFF(4, 4, 0, 16) AFF(4, 4, 0, 4) AFF(4, 4, 0, 32) AFF(5, 4, 0, 8) AFF(10, 4, 0, 8)
    return 0;
}
