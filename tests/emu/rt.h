// tests/emu/rt.h -- CPU emulation of the device runtime names used by fb_bev_amd/csrc/*.h.
//
// TEST INFRASTRUCTURE ONLY.  It lets the `not gpu` test-suite execute the *same kernel source and
// the same C-ABI launch code* (capi.hip) on the CPU at tiny sizes, to catch indexing / scan /
// tiling logic errors without a GPU.  It is never built into, nor loadable by, the product
// package (fb_bev_amd/_capi.py only ever opens libfbbev_hip.so and rejects CPU tensors).
//
// Model: blocks run one after another; the threads of a block are ucontext fibers scheduled
// round-robin on ONE OS thread, so __syncthreads() and the wave64 shuffles are exact rendezvous
// points and execution is deterministic.  Wave intrinsics must be called wave-uniformly.
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <algorithm>
#include <functional>
#include <vector>

#define FBBEV_TEST_OVERRIDES 1   /* emulator build: per-call test overrides of launcher plans (capi.hip) */
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 { unsigned x, y, z; };
struct float4 { float x, y, z, w; } __attribute__((aligned(16)));
struct int2 { int x, y; } __attribute__((aligned(8)));
inline int2 make_int2(int x, int y) { int2 r; r.x = x; r.y = y; return r; }
struct uint2 { unsigned int x, y; } __attribute__((aligned(8)));
inline uint2 make_uint2(unsigned int x, unsigned int y) { uint2 r; r.x = x; r.y = y; return r; }

namespace emu {
struct Fiber { ucontext_t ctx; bool done; };
struct State {
    dim3 tid{0, 0, 0}, bid{0, 0, 0}, bdim{1, 1, 1}, gdim{1, 1, 1};
    std::vector<Fiber> fibers;
    std::vector<char> stacks;
    ucontext_t sched;
    int cur = 0;
    int live = 0, arrived = 0; unsigned gen = 0;                 // block barrier
    int wlive[16], warrived[16]; unsigned wgen[16];              // wave rendezvous (<=1024 threads)
    uint64_t wslot[16][64];
    std::function<void()> body;
    std::vector<unsigned char> lds;
    int last_error = 0;
};
inline State& S() { static State s; return s; }
inline void yield() { State& s = S(); swapcontext(&s.fibers[s.cur].ctx, &s.sched); }
inline void trampoline() {
    State& s = S();
    s.body();
    const int me = s.cur, w = me >> 6;
    s.fibers[me].done = true;
    s.live--; s.wlive[w]--;
    if (s.live > 0 && s.arrived == s.live) { s.arrived = 0; s.gen++; }
    if (s.wlive[w] > 0 && s.warrived[w] == s.wlive[w]) { s.warrived[w] = 0; s.wgen[w]++; }
    swapcontext(&s.fibers[me].ctx, &s.sched);
}
inline void block_barrier() {
    State& s = S();
    const unsigned g = s.gen;
    if (++s.arrived == s.live) { s.arrived = 0; s.gen++; return; }
    while (s.gen == g) yield();
}
inline void wave_barrier() {
    State& s = S();
    const int w = s.cur >> 6;
    const unsigned g = s.wgen[w];
    if (++s.warrived[w] == s.wlive[w]) { s.warrived[w] = 0; s.wgen[w]++; return; }
    while (s.wgen[w] == g) yield();
}
inline void launch(unsigned grid, unsigned block, size_t lds_bytes, std::function<void()> body) {
    State& s = S();
    const size_t STK = 256 * 1024;
    if (s.fibers.size() < block) { s.fibers.resize(block); s.stacks.resize((size_t)block * STK); }
    if (s.lds.size() < lds_bytes + 16) s.lds.resize(lds_bytes + 16);
    s.body = body;
    s.bdim = dim3{block, 1, 1};
    s.gdim = dim3{grid, 1, 1};
    for (unsigned b = 0; b < grid; ++b) {
        s.bid = dim3{b, 0, 0};
        s.live = (int)block; s.arrived = 0;
        for (int w = 0; w < 16; ++w) { s.wlive[w] = 0; s.warrived[w] = 0; }
        for (unsigned t = 0; t < block; ++t) {
            s.wlive[t >> 6]++;
            Fiber& f = s.fibers[t];
            f.done = false;
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = s.stacks.data() + (size_t)t * STK;
            f.ctx.uc_stack.ss_size = STK;
            f.ctx.uc_link = &s.sched;
            makecontext(&f.ctx, (void (*)())trampoline, 0);
        }
        int remaining = (int)block;
        while (remaining > 0) {
            remaining = 0;
            for (unsigned t = 0; t < block; ++t) {
                if (s.fibers[t].done) continue;
                s.cur = (int)t;
                s.tid = dim3{t, 0, 0};
                swapcontext(&s.sched, &s.fibers[t].ctx);
                if (!s.fibers[t].done) remaining++;
            }
        }
    }
}
template <class T> inline uint64_t to_bits(T v) { uint64_t b = 0; memcpy(&b, &v, sizeof(T)); return b; }
template <class T> inline T from_bits(uint64_t b) { T v; memcpy(&v, &b, sizeof(T)); return v; }
template <class T> inline T shfl_src(T v, int src_lane_or_neg) {
    State& s = S();
    const int w = s.cur >> 6, lane = s.cur & 63;
    s.wslot[w][lane] = to_bits(v);
    wave_barrier();
    T r = (src_lane_or_neg < 0 || src_lane_or_neg > 63) ? v : from_bits<T>(s.wslot[w][src_lane_or_neg]);
    wave_barrier();
    return r;
}
}  // namespace emu

#define threadIdx (emu::S().tid)
#define blockIdx (emu::S().bid)
#define blockDim (emu::S().bdim)
#define gridDim (emu::S().gdim)

inline void __syncthreads() { emu::block_barrier(); }
template <class T> inline T __shfl_xor(T v, int mask, int = 64) { return emu::shfl_src(v, (emu::S().cur & 63) ^ mask); }
template <class T> inline T __shfl_up(T v, int d, int = 64) { return emu::shfl_src(v, (emu::S().cur & 63) - d); }
template <class T> inline T __shfl_down(T v, int d, int = 64) { return emu::shfl_src(v, (emu::S().cur & 63) + d); }
template <class T> inline T __shfl(T v, int src, int = 64) { return emu::shfl_src(v, src & 63); }
inline unsigned long long __ballot(int pred) {
    emu::State& s = emu::S();
    const int w = s.cur >> 6, lane = s.cur & 63;
    s.wslot[w][lane] = pred ? 1 : 0;
    emu::wave_barrier();
    unsigned long long m = 0;
    const int base = w << 6;
    for (int l = 0; l < 64; ++l)
        if (base + l < (int)s.bdim.x && !s.fibers[base + l].done && s.wslot[w][l]) m |= (1ull << l);
    emu::wave_barrier();
    return m;
}
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __float2int_rn(float x) { return (int)__builtin_rintf(x); }      // nearest even, as v_cvt_i32 after v_rndne
inline int __ffsll(long long x) { return __builtin_ffsll(x); }
inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
inline unsigned int atomicMax(unsigned int* p, unsigned int v) { unsigned int o = *p; if (v > o) *p = v; return o; }
inline unsigned int atomicOr(unsigned int* p, unsigned int v) { unsigned int o = *p; *p = o | v; return o; }

// single correctly-rounded fp32 ops (volatile defeats re-association / contraction)
inline float fbbev_mul(float a, float b) { volatile float r = a * b; return r; }
inline float fbbev_add(float a, float b) { volatile float r = a + b; return r; }
inline float fbbev_sub(float a, float b) { volatile float r = a - b; return r; }
inline float fbbev_div(float a, float b) { volatile float r = a / b; return r; }
inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
inline float __expf(float x) { return expf(x); }   // device fast exp (v_exp_f32 based): the emulator uses libm's

typedef void* fbbev_rt_stream;
#define FBBEV_LAUNCH(kern, grid, block, lds_bytes, stream, ...) \
    emu::launch((unsigned)(grid), (unsigned)(block), (size_t)(lds_bytes), [=]() { kern(__VA_ARGS__); })
static inline int fbbev_rt_last_error() { return 0; }
typedef int fbbev_rt_event;                                  // launches are synchronous here: streams and events are no-ops
static inline int fbbev_rt_device() { return 0; }
static inline int fbbev_rt_stream_create(fbbev_rt_stream* s, int) { *s = nullptr; return 0; }
static inline int fbbev_rt_event_create(fbbev_rt_event* e) { *e = 0; return 0; }
static inline int fbbev_rt_event_record(fbbev_rt_event, fbbev_rt_stream) { return 0; }
static inline int fbbev_rt_stream_wait(fbbev_rt_stream, fbbev_rt_event) { return 0; }
static inline int fbbev_rt_allow_dyn_lds(const void*, size_t) { return 0; }
static inline int fbbev_rt_memset_async(void* p, int byte, size_t n, fbbev_rt_stream) { memset(p, byte, n); return 0; }
inline float* fbbev_dyn_lds_f32() {
    uintptr_t p = reinterpret_cast<uintptr_t>(emu::S().lds.data());
    return reinterpret_cast<float*>((p + 15) & ~uintptr_t(15));
}
typedef float fbbev_v4f __attribute__((vector_size(16)));
typedef float fbbev_v2f __attribute__((vector_size(8)));
typedef unsigned int fbbev_v4u __attribute__((vector_size(16)));
typedef int fbbev_v4i __attribute__((vector_size(16)));
template <int ST> inline void fbbev_store4(float* p, fbbev_v4f v) { memcpy(p, &v, 16); }
template <typename T> inline void fbbev_st(T* p, T v) { *p = v; }
inline void fbbev_atomic_add_f32(float* p, float v) { *p += v; }
inline void fbbev_lds_atomic_add_f32(float* p, float v) { *p += v; }
inline void fbbev_lds_atomic_add_i64(long long* p, long long v) { *p += v; }
inline int fbbev_cvt_rpi(float x) {                       // v_cvt_rpi_i32_f32: floor(x + 0.5), saturating
    const double r = __builtin_floor((double)x + 0.5);
    return r >= 2147483647.0 ? 2147483647 : (r <= -2147483648.0 ? (int)(-2147483647 - 1) : (int)r);
}
// emulation of v_mfma_f32_16x16x4_f32 with the documented fragment layouts (see csrc/hip_rt/rt.h); every lane of the
// wave must call it (wave-uniform control flow, as on the hardware)
inline fbbev_v4f fbbev_mfma_f32_16x16x4(float a, float b, fbbev_v4f c) {
    emu::State& s = emu::S();
    const int w = s.cur >> 6, lane = s.cur & 63;
    static thread_local float A[16][64], B[16][64];
    A[w][lane] = a; B[w][lane] = b;
    emu::wave_barrier();
    const int g = lane >> 4, j = lane & 15;
    fbbev_v4f d = c;
    for (int r = 0; r < 4; ++r) {
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(A[w][k * 16 + 4 * g + r], B[w][k * 16 + j], acc);
        d[r] = acc;
    }
    emu::wave_barrier();
    return d;
}

// bf16 (round to nearest even, NaN kept quiet) and the emulation of v_mfma_f32_16x16x32_bf16: slot (lane/16, e) of A and B
// stands for k = 8*(lane/16) + e; products are exact in fp32, accumulation is a k-ordered fp32 chain (the hardware's
// internal order is unspecified: tests compare with a tolerance)
struct fbbev_bf16x8 { uint16_t v[8]; };
inline uint16_t fbbev_emu_bf16(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float fbbev_emu_bf16_to_f32(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
inline unsigned int fbbev_f32_to_f16(float f);          // pool_kernels.h: integer-only round to nearest even
template <int ET> inline unsigned int fbbev_cvt_pk16(float lo, float hi) {
    if constexpr (ET == 1) return (unsigned int)fbbev_emu_bf16(lo) | ((unsigned int)fbbev_emu_bf16(hi) << 16);
    else return fbbev_f32_to_f16(lo) | (fbbev_f32_to_f16(hi) << 16);
}
inline fbbev_bf16x8 fbbev_cvt_bf16x8(fbbev_v4f lo, fbbev_v4f hi) {
    fbbev_bf16x8 r;
    for (int e = 0; e < 4; ++e) { r.v[e] = fbbev_emu_bf16(lo[e]); r.v[4 + e] = fbbev_emu_bf16(hi[e]); }
    return r;
}
inline fbbev_bf16x8 fbbev_ld_bf16x8(const void* p) { fbbev_bf16x8 r; memcpy(&r, p, 16); return r; }
inline fbbev_v4f fbbev_mfma_f32_16x16x32_bf16(fbbev_bf16x8 a, fbbev_bf16x8 b, fbbev_v4f c) {
    emu::State& s = emu::S();
    const int w = s.cur >> 6, lane = s.cur & 63;
    static thread_local fbbev_bf16x8 A[16][64], B[16][64];
    A[w][lane] = a; B[w][lane] = b;
    emu::wave_barrier();
    const int g = lane >> 4, j = lane & 15;
    fbbev_v4f d = c;
    for (int r = 0; r < 4; ++r) {
        float acc = c[r];
        for (int gg = 0; gg < 4; ++gg)
            for (int e = 0; e < 8; ++e)
                acc += fbbev_emu_bf16_to_f32(A[w][gg * 16 + 4 * g + r].v[e]) * fbbev_emu_bf16_to_f32(B[w][gg * 16 + j].v[e]);
        d[r] = acc;
    }
    emu::wave_barrier();
    return d;
}

// IEEE binary16 bits -> binary32, exact (the GPU build uses v_cvt_f32_f16)
inline float fbbev_f16_bits_to_f32(unsigned int h) {
    const unsigned int sign = (h & 0x8000u) << 16;
    unsigned int e = (h >> 10) & 0x1fu, m = h & 0x3ffu, u;
    if (e == 0) {
        if (m == 0) u = sign;
        else {                                              // subnormal half: normalise
            int sh = 0;
            while (!(m & 0x400u)) { m <<= 1; ++sh; }
            u = sign | ((unsigned int)(113 - sh) << 23) | ((m & 0x3ffu) << 13);
        }
    } else if (e == 31) u = sign | 0x7f800000u | (m << 13);
    else u = sign | ((e + 112u) << 23) | (m << 13);
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
// emulation of v_mfma_f32_16x16x32_f16 (raw halves; same slot rule and k-ordered fp32 chain as the bf16 form above)
inline fbbev_v4f fbbev_mfma_f32_16x16x32_f16(fbbev_v4u a, fbbev_v4u b, fbbev_v4f c) {
    emu::State& s = emu::S();
    const int w = s.cur >> 6, lane = s.cur & 63;
    static thread_local fbbev_v4u A[16][64], B[16][64];
    A[w][lane] = a; B[w][lane] = b;
    emu::wave_barrier();
    const int g = lane >> 4, j = lane & 15;
    fbbev_v4f d = c;
    auto half = [](const fbbev_v4u& v, int e) { return fbbev_f16_bits_to_f32((v[e >> 1] >> (16 * (e & 1))) & 0xffffu); };
    for (int r = 0; r < 4; ++r) {
        float acc = c[r];
        for (int gg = 0; gg < 4; ++gg)
            for (int e = 0; e < 8; ++e) acc += half(A[w][gg * 16 + 4 * g + r], e) * half(B[w][gg * 16 + j], e);
        d[r] = acc;
    }
    emu::wave_barrier();
    return d;
}
inline void fbbev_lds_dma16(const void* gsrc, void* lds_wave_base) {
    memcpy(static_cast<char*>(lds_wave_base) + 16 * (emu::S().cur & 63), gsrc, 16);
}
inline void fbbev_wait_loads() {}
template <int N> inline void fbbev_wait_loads_but() {}
template <int HI> inline float fbbev_fma_f16(unsigned int pair, float w, float acc) {
    return __builtin_fmaf(fbbev_f16_bits_to_f32((pair >> (16 * HI)) & 0xffffu), w, acc);
}
inline void fbbev_wave_sync() { emu::wave_barrier(); }
inline void fbbev_sched_fence() {}
#define FBBEV_SCHED_MFMA(N) ((void)0)
#define FBBEV_SCHED_LDS_READ(N) ((void)0)
inline void fbbev_pin(fbbev_v2f&) {}
inline void fbbev_opaque(int&) {}
inline void fbbev_opaque(float&) {}
inline float fbbev_lds_ld_f32(const float* p) { return *p; }
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }    // wave-uniform by construction where the kernels use it
inline int fbbev_lds_ld_i32(const int* p) { return *p; }
inline unsigned int fbbev_umulhi(unsigned int a, unsigned int b) { return (unsigned int)(((unsigned long long)a * b) >> 32); }
inline unsigned int fbbev_mad_u24_vsv(unsigned int a, unsigned int b, unsigned int c) { return (unsigned int)((unsigned long long)(a & 0xffffffu) * (b & 0xffffffu)) + c; }
template <unsigned int K>
inline unsigned int fbbev_mad_u24_vks(unsigned int a, unsigned int c) { return (unsigned int)((unsigned long long)(a & 0xffffffu) * K) + c; }
inline fbbev_v4f fbbev_gld_v4f(const float* p) { return *reinterpret_cast<const fbbev_v4f*>(p); }
inline fbbev_v4f fbbev_lds_ld_v4f_a8(const float* p) { return fbbev_v4f{p[0], p[1], p[2], p[3]}; }
inline void fbbev_opaque_u32(unsigned int x) { volatile unsigned int sink = x; (void)sink; }
