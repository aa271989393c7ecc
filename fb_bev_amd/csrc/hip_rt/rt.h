// rt.h -- device runtime glue for the gfx950 build (HIP).  The kernels and the C-ABI launchers
// include "rt.h" and nothing else from HIP; tests/emu/rt.h offers the same names for the
// CPU kernel-logic emulator used by the `not gpu` tests (test infrastructure, never shipped).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

typedef hipStream_t fbbev_rt_stream;

#define FBBEV_LAUNCH(kern, grid, block, lds_bytes, stream, ...) \
    hipLaunchKernelGGL(kern, dim3((unsigned)(grid)), dim3((unsigned)(block)), (size_t)(lds_bytes), stream, __VA_ARGS__)

static inline int fbbev_rt_last_error() { return (int)hipGetLastError(); }

static inline int fbbev_rt_memset_async(void* p, int byte, size_t n, fbbev_rt_stream s) {
    return (int)hipMemsetAsync(p, byte, n, s);
}

// a second stream of the calling thread's device and the events that order work across the two (the pipelined history step)
typedef hipEvent_t fbbev_rt_event;
static inline int fbbev_rt_device() { int d = 0; return hipGetDevice(&d) == hipSuccess ? d : -1; }
// high = 1: the device's highest stream priority (its workgroups are dispatched ahead of the caller's stream's when both wait for a CU)
static inline int fbbev_rt_stream_create(fbbev_rt_stream* s, int high) {
    int least = 0, greatest = 0;
    if (high && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least)
        return (int)hipStreamCreateWithPriority(s, hipStreamNonBlocking, greatest);
    return (int)hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}
static inline int fbbev_rt_event_create(fbbev_rt_event* e) { return (int)hipEventCreateWithFlags(e, hipEventDisableTiming); }
static inline int fbbev_rt_event_record(fbbev_rt_event e, fbbev_rt_stream s) { return (int)hipEventRecord(e, s); }
static inline int fbbev_rt_stream_wait(fbbev_rt_stream s, fbbev_rt_event e) { return (int)hipStreamWaitEvent(s, e, 0); }

static inline int fbbev_rt_allow_dyn_lds(const void* kern, size_t bytes) {
    return (int)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// dynamic LDS carve-out; base is 16-byte aligned (no static __shared__ precedes it in any kernel
// that uses it -- cdna_hip_programming.md Guideline 17)
extern __shared__ __attribute__((aligned(16))) unsigned char fbbev_dyn_lds_raw[];
__device__ __forceinline__ float* fbbev_dyn_lds_f32() { return reinterpret_cast<float*>(fbbev_dyn_lds_raw); }

// 16-byte store with a selectable cache policy for the write-once streaming output.
//   0 plain | 1 nt (clang nontemporal builtin) | 2 sc1 | 3 sc0 sc1 | 4 sc1 nt | 5 sc0 nt | 6 sc0 sc1 nt | 7 sc0
// (gfx950 cache-control bits; MI355X_MICROARCH.md "stores of each flavour").  Stores have no
// return value, so the hand-written forms need no extra s_waitcnt bookkeeping.
typedef float fbbev_v4f __attribute__((ext_vector_type(4)));
typedef float fbbev_v2f __attribute__((ext_vector_type(2)));
typedef unsigned int fbbev_v4u __attribute__((ext_vector_type(4)));
typedef int fbbev_v4i __attribute__((ext_vector_type(4)));
template <int ST>
__device__ __forceinline__ void fbbev_store4(float* p, fbbev_v4f v) {
    if constexpr (ST == 1) {
        __builtin_nontemporal_store(v, reinterpret_cast<fbbev_v4f*>(p));
    } else if constexpr (ST == 2) {
        asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
    } else if constexpr (ST == 3) {
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
    } else if constexpr (ST == 4) {
        asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" : : "v"(p), "v"(v) : "memory");
    } else if constexpr (ST == 5) {
        asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" : : "v"(p), "v"(v) : "memory");
    } else if constexpr (ST == 6) {
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" : : "v"(p), "v"(v) : "memory");
    } else if constexpr (ST == 7) {
        asm volatile("global_store_dwordx4 %0, %1, off sc0" : : "v"(p), "v"(v) : "memory");
    } else {
        *reinterpret_cast<fbbev_v4f*>(p) = v;
    }
}

// Stores of the backward projection's INTERMEDIATE tensors (token rows, head planes, attention outputs, the refined BEV).  Round 6
// (tools/dbg_store_policy.py, profiles/r06_exp_store_policy.md): a kernel that writes with plain stores parks its lines dirty in the 256 MB
// memory-side cache, and the NEXT streaming kernel pays for their write-back (the dense pooling kernel: 166 us behind 512 MB of `nt`
// stores, 287 us behind the same bytes stored plain).  FBBEV_STREAM_STORES=1 (build flag) makes these stores non-temporal: the
// latency-bound attention kernels write through while HBM is idle.  Values are unaffected.
#ifndef FBBEV_STREAM_STORES
#define FBBEV_STREAM_STORES 0
#endif
template <typename T>
__device__ __forceinline__ void fbbev_st(T* p, T v) {
#if FBBEV_STREAM_STORES
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

__device__ __forceinline__ void fbbev_atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }
// 64-bit integer add on an LDS word pair (ds_add_u64, no return value): 22 lane-adds per ns and CU on gfx950 where
// ds_add_f32 manages 0.8 (profiles/r02_micro_lds_atomics.jsonl)
__device__ __forceinline__ void fbbev_lds_atomic_add_i64(long long* p, long long v) {
    __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// float -> int32, floor(x + 0.5) in ONE instruction (v_cvt_rpi_i32_f32; __float2int_rn is v_rndne_f32 + v_cvt_i32_f32 and differs
// from it only on exact .5 ties).  The fixed-point LDS planes convert 40 addends per sample with it.
__device__ __forceinline__ int fbbev_cvt_rpi(float x) {
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
// fp32 add on an LDS word (ds_add_f32, no return value)
__device__ __forceinline__ void fbbev_lds_atomic_add_f32(float* p, float v) {
    __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP, false);
}

// IEEE binary16 bits -> binary32: v_cvt_f32_f16, exact for every half (subnormals included)
__device__ __forceinline__ float fbbev_f16_bits_to_f32(unsigned int h) {
    const unsigned short b = (unsigned short)h;
    _Float16 x;
    __builtin_memcpy(&x, &b, 2);
    return (float)x;
}

// two floats -> one packed pair of 16-bit elements (lo in bits 0..15), round to nearest even, by the conversion
// instructions (ET 1: bf16, v_cvt_pk_bf16_f32; ET 2: f16, v_cvt_f16_f32 -- half subnormals kept, overflow -> inf).  The same
// bits as the integer-only fbbev_pack2 (pool_kernels.h) for every non-NaN input (tested on the GPU against it); NaNs come
// out quiet with the instruction's payload.
template <int ET>
__device__ __forceinline__ unsigned int fbbev_cvt_pk16(float lo, float hi) {
    unsigned int u;
    if constexpr (ET == 1) {
        typedef __bf16 pair __attribute__((ext_vector_type(2)));
        const pair r = {(__bf16)lo, (__bf16)hi};
        __builtin_memcpy(&u, &r, 4);
    } else {
        typedef _Float16 pair __attribute__((ext_vector_type(2)));
        const pair r = {(_Float16)lo, (_Float16)hi};
        __builtin_memcpy(&u, &r, 4);
    }
    return u;
}

// v_mfma_f32_16x16x4_f32: D(16x16) = A(16x4) . B(4x16) + C, exact fp32 (a k-ordered fmaf chain per element).
// A: lane holds A[lane%16][lane/16]; B: lane holds B[lane/16][lane%16]; register r of C/D: row 4*(lane/16)+r, col lane%16.
__device__ __forceinline__ fbbev_v4f fbbev_mfma_f32_16x16x4(float a, float b, fbbev_v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// v_mfma_f32_16x16x32_bf16: D(16x16) = A(16x32) . B(32x16) + C with bf16 operands and fp32 accumulation.  A lane passes 8
// bf16 of row lane%16 of A and 8 bf16 of column lane%16 of B; which k each (lane/16, element) slot stands for is the SAME
// function for A and B, so a kernel that fills both operands with the same slot -> channel rule needs no further
// knowledge of it.  C/D layout as the f32 form (dtype independent on gfx950): register r = row 4*(lane/16)+r, col lane%16.
typedef __bf16 fbbev_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ fbbev_bf16x8 fbbev_cvt_bf16x8(fbbev_v4f lo, fbbev_v4f hi) {     // round to nearest even
    fbbev_bf16x8 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) { r[e] = (__bf16)lo[e]; r[4 + e] = (__bf16)hi[e]; }
    return r;
}
__device__ __forceinline__ fbbev_bf16x8 fbbev_ld_bf16x8(const void* p) { return *reinterpret_cast<const fbbev_bf16x8*>(p); }
__device__ __forceinline__ fbbev_v4f fbbev_mfma_f32_16x16x32_bf16(fbbev_bf16x8 a, fbbev_bf16x8 b, fbbev_v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// v_mfma_f32_16x16x32_f16: the same shape and slot rule with IEEE binary16 operands (8 halves = 16 raw bytes per lane), fp32
// accumulation: a product of two halves is exact in fp32 (11 x 11 mantissa bits).
__device__ __forceinline__ fbbev_v4f fbbev_mfma_f32_16x16x32_f16(fbbev_v4u a, fbbev_v4u b, fbbev_v4f c) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    h8 ha, hb;
    __builtin_memcpy(&ha, &a, 16);
    __builtin_memcpy(&hb, &b, 16);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c, 0, 0, 0);
}

// fmaf(half, w, acc) with the half taken straight from a packed pair (v_fma_mix_f32: the widening is part of the instruction -- the
// same value as v_cvt_f32_f16 + v_fma_f32, one issue slot instead of two).  HI = 0: bits 0..15, HI = 1: bits 16..31.
template <int HI>
__device__ __forceinline__ float fbbev_fma_f16(unsigned int pair, float w, float acc) {
    float d;
    if constexpr (HI == 0) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(pair), "v"(w), "v"(acc));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(pair), "v"(w), "v"(acc));
    return d;
}

// 16 bytes per lane straight from global memory into LDS (global_load_lds_dwordx4): lane l's bytes land at lds_wave_base + 16 l
// (the base is wave-uniform).  Counted by vmcnt like any load; fbbev_wait_loads() before the barrier that publishes the data.
__device__ __forceinline__ void fbbev_lds_dma16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void fbbev_wait_loads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// ... for everything but the N youngest loads / stores of the wave (vmcnt retires in issue order)
template <int N> __device__ __forceinline__ void fbbev_wait_loads_but() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

// wave-level ordering point for a wave-PRIVATE LDS region: the 64 lanes run in lockstep and the LDS queue of a wave is
// in order, so only the compiler has to be kept from moving LDS accesses across it (no s_barrier, no other wave waits)
// instruction-scheduling fence: nothing is moved across it (keeps prefetch loads ahead of the MFMA block they overlap)
__device__ __forceinline__ void fbbev_sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// instruction-order hint inside a scheduling region: the next N instructions of a kind (MFMA / LDS read) form a group, groups are
// issued in the order they are declared -- spells out "five fragment reads ahead of the MFMA that consumes them" where the
// compiler on its own issues each read right in front of its MFMA and waits for it (history_fused_x3_kernels.h)
#define FBBEV_SCHED_MFMA(N) __builtin_amdgcn_sched_group_barrier(0x008, N, 0)
#define FBBEV_SCHED_LDS_READ(N) __builtin_amdgcn_sched_group_barrier(0x100, N, 0)

// ONE correctly rounded fp32 operation that is never fused with a neighbour.  HIP's __fmul_rn / __fadd_rn are plain `x * y` /
// `x + y` (no OCML_BASIC_ROUNDED_OPERATIONS) and inherit -ffp-contract=fast: after inlining, a product feeding a sum may
// still become an fma -- or not, depending on the surrounding kernel.  Here the operations are built without the `contract`
// flag (the pragma is per instruction and survives inlining), so kernels that must agree bit for bit can share a formula.
__device__ __forceinline__ float fbbev_mul(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float fbbev_add(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float fbbev_sub(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}
__device__ __forceinline__ float fbbev_div(float a, float b) { return __fdiv_rn(a, b); }

// value barrier: the compiler must materialise x here and may not look through it (used where an LDS load followed by a
// conditional global override of the same variable was if-converted into ONE flat load of a selected pointer)
__device__ __forceinline__ void fbbev_opaque_u32(unsigned int x) { asm volatile("" : : "v"(x)); }
__device__ __forceinline__ void fbbev_opaque(int& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void fbbev_opaque(float& x) { asm volatile("" : "+v"(x)); }
// ordering point for a software pipeline: x must be COMPUTED here, and no memory operation moves across (the "memory"
// clobber).  __builtin_amdgcn_sched_barrier alone does not do this: it binds the machine scheduler, but pure arithmetic is
// not chained to it when the selection DAG is linearised -- the blend of sample i floated below the loads of samples
// i+2 and i+3 (three samples of registers live, 285 VGPRs) until its results were pinned like this.
__device__ __forceinline__ void fbbev_pin(fbbev_v2f& x) { asm volatile("" : "+v"(x) : : "memory"); }
// read of a float that is KNOWN to live in LDS, through an explicit local-address-space pointer: always a ds_read, never
// merged with a global load of the other arm of a condition
__device__ __forceinline__ float fbbev_lds_ld_f32(const float* p) {
    return *(const __attribute__((address_space(3))) float*)p;
}
// a 16-byte load through a pointer the compiler cannot trace to a kernel argument (a nullable member of a by-value struct): global, not flat
__device__ __forceinline__ fbbev_v4f fbbev_gld_v4f(const float* p) { return *(const __attribute__((address_space(1))) fbbev_v4f*)p; }
__device__ __forceinline__ int fbbev_lds_ld_i32(const int* p) { return *(const __attribute__((address_space(3))) int*)p; }
// 16 bytes at an 8-byte aligned LDS address (two ds_read_b64 / one ds_read2_b64: head-plane tokens of 10 floats are 8-byte aligned)
// high 32 bits of a 32 x 32-bit product (division by a run-time constant through its reciprocal)
__device__ __forceinline__ unsigned int fbbev_umulhi(unsigned int a, unsigned int b) { return __umulhi(a, b); }
// a * b + c with a, b < 2^24 as ONE full-rate v_mad_u32_u24 (the compiler turns the mul24 builtins back into the quarter-rate
// v_mad_u64_u32 / v_mul_lo_u32 when it cannot prove the ranges).  _vsv: b wave-uniform (SGPR); _vks: b a compile-time inline
// constant (<= 64), c wave-uniform
__device__ __forceinline__ unsigned int fbbev_mad_u24_vsv(unsigned int a, unsigned int b_uniform, unsigned int c) {
    unsigned int r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c));
    return r;
}
template <unsigned int K>
__device__ __forceinline__ unsigned int fbbev_mad_u24_vks(unsigned int a, unsigned int c_uniform) {
    static_assert(K <= 64, "inline constant");
    unsigned int r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "n"(K), "s"(c_uniform));
    return r;
}
__device__ __forceinline__ fbbev_v4f fbbev_lds_ld_v4f_a8(const float* p) {
    const __attribute__((address_space(3))) fbbev_v2f* q = (const __attribute__((address_space(3))) fbbev_v2f*)p;
    const fbbev_v2f a = q[0], b = q[1];
    return fbbev_v4f{a[0], a[1], b[0], b[1]};
}

__device__ __forceinline__ void fbbev_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
