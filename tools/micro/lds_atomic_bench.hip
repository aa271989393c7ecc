// LDS atomic throughput on gfx950: wave-instructions per CU-cycle for ds_add_f32 / ds_add_u32 / ds_add_u64 / plain
// read-add-write, by address pattern (distinct per lane, 4 lanes per address, 16 lanes per address, one address).
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/micro/lds_atomic_bench.hip -o tools/micro/lds_atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP, int NT>
__global__ void __launch_bounds__(NT) k_bench(int iters, int dup, int stride, long long* cycles, float* sink) {
    extern __shared__ unsigned char raw[];
    float* f = reinterpret_cast<float*>(raw);
    unsigned int* u = reinterpret_cast<unsigned int*>(raw);
    unsigned long long* q = reinterpret_cast<unsigned long long*>(raw);
    for (int i = threadIdx.x; i < 8192; i += NT) f[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int addr = ((lane / dup) * stride + wave * 1031) & 4095;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int a = (addr + k * 67) & 4095;
            if (OP == 0) __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float*)(f + a), 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP, false);
            else if (OP == 1) atomicAdd(u + a, 1u);
            else if (OP == 2) atomicAdd(q + (a & 2047), 1ull);
            else if (OP == 3) { f[a] = f[a] + 1.0f; }
            else if (OP == 4) atomicMax(u + a, (unsigned)i);
        }
        addr = (addr + 129) & 4095;
    }
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (threadIdx.x == 0) sink[blockIdx.x] = f[addr];
}

template <int OP, int NT>
void run(const char* name, int dup, int stride, int wgs_per_cu) {
    const int iters = 2000, wgs = 256 * wgs_per_cu;
    long long* cyc; float* sink;
    hipMalloc(&cyc, wgs * sizeof(long long)); hipMalloc(&sink, wgs * sizeof(float));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_bench<OP, NT>), wgs, NT, 32768, 0, 10, dup, stride, cyc, sink);
    hipEventRecord(a);
    hipLaunchKernelGGL((k_bench<OP, NT>), wgs, NT, 32768, 0, iters, dup, stride, cyc, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<long long> h(wgs); hipMemcpy(h.data(), cyc, wgs * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : h) avg += c; avg /= wgs;
    const double insts_per_wave = iters * 8.0, waves_per_cu = wgs_per_cu * (NT / 64.0);
    // clock64 counts at a fixed 100 MHz-class rate on this part: report time-based numbers
    printf("{\"op\": \"%s\", \"threads\": %d, \"wgs_per_cu\": %d, \"lanes_per_address\": %d, \"stride\": %d, \"ms\": %.4f, "
           "\"ns_per_wave_instr_per_cu\": %.2f, \"lane_ops_per_ns_per_cu\": %.2f}\n", name, NT, wgs_per_cu, dup, stride, ms,
           ms * 1e6 / (insts_per_wave * waves_per_cu), insts_per_wave * waves_per_cu * 64 / (ms * 1e6));
    hipFree(cyc); hipFree(sink);
}

int main() {
    for (int dup : {1, 4, 16, 64}) {
        run<0, 256>("ds_add_f32", dup, 1, 4);
        run<1, 256>("ds_add_u32", dup, 1, 4);
        run<2, 256>("ds_add_u64", dup, 1, 4);
        run<4, 256>("ds_max_u32", dup, 1, 4);
    }
    run<3, 256>("plain_rmw_f32", 1, 1, 4);
    run<0, 256>("ds_add_f32", 1, 12, 4);      // the kernel's token pitch (12 floats)
    run<0, 256>("ds_add_f32", 4, 12, 4);
    run<0, 64>("ds_add_f32", 1, 1, 1);        // one wave per CU: latency of the instruction stream
    run<1, 64>("ds_add_u32", 1, 1, 1);
    return 0;
}
