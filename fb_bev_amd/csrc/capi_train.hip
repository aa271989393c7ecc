// capi_train.hip -- extern "C" entry points of the training path added in round 6 (include/fbbev.h): their own translation unit of
// libfbbev_hip.so, so that an edit of these kernels recompiles in seconds; the CPU emulator build includes this file from capi.hip.
#include <stdio.h>
#include <stdlib.h>
#include "rt.h"
#include "capi_common.h"
#include "wgrad_kernels.h"
#include "../../include/fbbev.h"

// ------------------------------------------------------------------------------ weight / bias gradient of a row-wise linear layer
struct wgrad_plan { int nti, n_oc, n_ic, ksteps, kps, n_split, amt; size_t part_w, part_b, total; };

static bool wgrad_plan_make(long long rows, int I, int O, wgrad_plan* p) {
    if (rows <= 0 || I <= 0 || O <= 0 || I % 4 != 0 || O % 4 != 0) return false;
    const long long ks = (rows + 31) / 32;
    if (ks >= (1ll << 30)) return false;
    p->nti = I <= 80 ? 5 : 8;
    p->n_oc = (O + 127) / 128;
    p->n_ic = (I + 16 * p->nti - 1) / (16 * p->nti);
    p->ksteps = (int)ks;
    // enough workgroups to fill the chip twice, few enough partial results that the fixed-order reduction stays small
    p->amt = O >= 128 ? 8 : (O + 15) / 16;
    // (768 workgroups for the narrow layers -- three per CU at their 41 KB of LDS -- measured slower than 256: 48.8 -> 54.8 us at 80 x 80)
    long long want = 512 / ((long long)p->n_oc * p->n_ic);
    static const int env_split = [] { const char* e = getenv("FBBEV_WGRAD_SPLITS"); return e ? atoi(e) : 0; }();   // tuning knob
    if (env_split > 0) want = env_split;
    if (want < 8) want = 8;
    if (want > 256) want = 256;
    if (want > ks) want = ks;
    p->kps = (int)((ks + want - 1) / want);
    p->n_split = (int)((ks + p->kps - 1) / p->kps);
    p->part_w = 0;
    p->part_b = align_up((size_t)p->n_split * O * I * sizeof(float), 256);
    p->total = p->part_b + align_up((size_t)p->n_split * O * sizeof(float), 256);
    return true;
}

extern "C" size_t fbbev_rows_wgrad_x3_ws_bytes(long long rows, int in_features, int out_features) {
    wgrad_plan p;
    return wgrad_plan_make(rows, in_features, out_features, &p) ? p.total : 0;
}

extern "C" int fbbev_rows_wgrad_x3(const float* grad_out, long long ld_grad, const float* x, long long ldx, const float* x_addend,
                                   long long addend_row_stride, long long addend_period, long long rows, int in_features,
                                   int out_features, float* grad_weight, float* grad_bias, void* workspace, size_t workspace_bytes,
                                   fbbev_stream_t stream_) {
    const int I = in_features, O = out_features;
    if (rows < 0 || I <= 0 || O <= 0) return FBBEV_E_BADARG;
    if (!grad_weight) return FBBEV_E_BADARG;
    fbbev_rt_stream stream = (fbbev_rt_stream)stream_;
    if (rows == 0) {
        int e = fbbev_rt_memset_async(grad_weight, 0, (size_t)O * I * sizeof(float), stream);
        if (!e && grad_bias) e = fbbev_rt_memset_async(grad_bias, 0, (size_t)O * sizeof(float), stream);
        return e;
    }
    if (!grad_out || !x) return FBBEV_E_BADARG;
    if (ld_grad == 0) ld_grad = O;
    if (ldx == 0) ldx = I;
    if (ld_grad < O || ldx < I) return FBBEV_E_BADARG;
    if (x_addend) {
        if (addend_period <= 0 || addend_period >= (1ll << 31)) return FBBEV_E_BADARG;
        if (addend_row_stride == 0) addend_row_stride = I;
        if (addend_row_stride < I) return FBBEV_E_BADARG;
        if (addend_row_stride % 4 != 0 || !aligned16(x_addend)) return FBBEV_E_UNSUPPORTED;
    } else {
        addend_period = 1;
    }
    wgrad_plan p;
    if (!wgrad_plan_make(rows, I, O, &p) || ld_grad % 4 != 0 || ldx % 4 != 0 || !aligned16(grad_out) || !aligned16(x))
        return FBBEV_E_UNSUPPORTED;
    if (!workspace || !aligned16(workspace) || workspace_bytes < p.total) return FBBEV_E_WORKSPACE;
    float* part_w = reinterpret_cast<float*>(static_cast<char*>(workspace) + p.part_w);
    float* part_b = grad_bias ? reinterpret_cast<float*>(static_cast<char*>(workspace) + p.part_b) : nullptr;
    const long long wgs = (long long)p.n_split * p.n_oc * p.n_ic;
    if (wgs >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    const size_t lds = (size_t)2 * (p.amt + p.nti) * FBBEV_WG_TILE_DW * 4;
#define FBBEV_WGRAD_LAUNCH(NTI_, ADD_)                                                                                            \
    do {                                                                                                                          \
        if (lds > 64 * 1024) {                                                                                                    \
            const int e_ = fbbev_rt_allow_dyn_lds((const void*)k_rows_wgrad_x3<NTI_, ADD_>, lds);                                  \
            if (e_) return e_;                                                                                                    \
        }                                                                                                                         \
        FBBEV_LAUNCH((k_rows_wgrad_x3<NTI_, ADD_>), wgs, 256, lds, stream, grad_out, ld_grad, x, ldx, x_addend, addend_row_stride, \
                     (int)addend_period, rows, O, I, p.n_oc, p.n_ic, p.ksteps, p.kps, p.amt, part_w, part_b);                             \
    } while (0)
    if (p.nti == 5) { if (x_addend) FBBEV_WGRAD_LAUNCH(5, true); else FBBEV_WGRAD_LAUNCH(5, false); }
    else { if (x_addend) FBBEV_WGRAD_LAUNCH(8, true); else FBBEV_WGRAD_LAUNCH(8, false); }
#undef FBBEV_WGRAD_LAUNCH
    FBBEV_CHECK_LAUNCH();
    const long long OI = (long long)O * I, total = OI + (grad_bias ? O : 0);
    FBBEV_LAUNCH(k_rows_wgrad_reduce<8>, (total + 31) / 32, 256, 0, stream, (const float*)part_w, (const float*)part_b, p.n_split, OI,
                 O, grad_weight, grad_bias);
    FBBEV_CHECK_LAUNCH();
    return 0;
}

// out (N) = sum over the leading dimension of x (B, N) [+ x2 (B, N)], ascending b
extern "C" int fbbev_sum_leading(const float* x, const float* x2, int B, long long N, float* out, fbbev_stream_t stream_) {
    if (B <= 0 || N < 0) return FBBEV_E_BADARG;
    if (N == 0) return 0;
    if (!x || !out) return FBBEV_E_BADARG;
    if (N % 4 != 0 || !aligned16(x) || !aligned16(out) || (x2 && !aligned16(x2))) return FBBEV_E_UNSUPPORTED;
    const long long n4 = N / 4;
    if ((n4 + 255) / 256 >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    FBBEV_LAUNCH(k_sum_leading<0>, (n4 + 255) / 256, 256, 0, (fbbev_rt_stream)stream_, x, x2, B, n4, reinterpret_cast<fbbev_v4f*>(out));
    FBBEV_CHECK_LAUNCH();
    return 0;
}

// out (len) = sum over the n rows of part (n, len), fixed association (32 lanes each adding every 32nd row in ascending order, lane sums
// added in lane order): the per-workgroup partial parameter gradients of fbbev_layernorm_bwd -- ATen's reduction over the leading
// dimension of a (2048, 160) tensor is one latency chain per column (79 us per LayerNorm at 160 000 rows)
extern "C" int fbbev_sum_partials(const float* part, int n, long long len, float* out, fbbev_stream_t stream_) {
    if (n <= 0 || len < 0) return FBBEV_E_BADARG;
    if (len == 0) return 0;
    if (!part || !out) return FBBEV_E_BADARG;
    FBBEV_LAUNCH(k_rows_wgrad_reduce<32>, (len + 31) / 32, 1024, 0, (fbbev_rt_stream)stream_, part, (const float*)nullptr, n, len, 0, out,
                 (float*)nullptr);
    FBBEV_CHECK_LAUNCH();
    return 0;
}

// diagnostics (tools/dbg_store_policy.py): fill n floats with 16-byte stores of cache policy `policy` (fbbev_store4: 0 plain, 1 nt, 2 sc1,
// 3 sc0 sc1, 4 sc1 nt, 5 sc0 nt, 6 sc0 sc1 nt, 7 sc0) -- what a kernel's stores leave behind for the NEXT kernel's store stream
template <int ST>
__global__ void __launch_bounds__(256)
k_diag_fill(float* __restrict__ p, long long n4, float v) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
        fbbev_store4<ST>(p + 4 * i, fbbev_v4f{v, v, v, v});
}
extern "C" int fbbev_diag_fill(float* p, long long n, int policy, fbbev_stream_t stream_) {
    if (!p || n < 0 || n % 4 != 0 || policy < 0 || policy > 7 || !aligned16(p)) return FBBEV_E_BADARG;
    if (n == 0) return 0;
    const long long n4 = n / 4;
    long long blocks = (n4 + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    fbbev_rt_stream stream = (fbbev_rt_stream)stream_;
    switch (policy) {
    case 0: FBBEV_LAUNCH(k_diag_fill<0>, blocks, 256, 0, stream, p, n4, 1.f); break;
    case 1: FBBEV_LAUNCH(k_diag_fill<1>, blocks, 256, 0, stream, p, n4, 1.f); break;
    case 2: FBBEV_LAUNCH(k_diag_fill<2>, blocks, 256, 0, stream, p, n4, 1.f); break;
    case 3: FBBEV_LAUNCH(k_diag_fill<3>, blocks, 256, 0, stream, p, n4, 1.f); break;
    case 4: FBBEV_LAUNCH(k_diag_fill<4>, blocks, 256, 0, stream, p, n4, 1.f); break;
    case 5: FBBEV_LAUNCH(k_diag_fill<5>, blocks, 256, 0, stream, p, n4, 1.f); break;
    case 6: FBBEV_LAUNCH(k_diag_fill<6>, blocks, 256, 0, stream, p, n4, 1.f); break;
    default: FBBEV_LAUNCH(k_diag_fill<7>, blocks, 256, 0, stream, p, n4, 1.f); break;
    }
    FBBEV_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------ read-ahead of a kernel's gather sources
// The dense pooling kernel at the END of the forward-backward step gathers through index tensors, depth and feature rows that were
// written ~1 ms and ~0.7 GB of intermediate traffic earlier: they have left the 256 MB memory-side cache, and the kernel's dependent
// gather chains run at HBM latency (tools/dbg_pool_in_step.py: 260 us cold, 176 us with the ~30 MB of sources read once right before).
// fbbev_touch reads up to 8 spans with 16-byte loads and discards them: the lines are back in the memory-side cache (and the L2s) when
// the gathers start.
struct fbbev_touch_spans { const char* p[8]; unsigned long long n16[8]; };
template <int UNUSED>
__global__ void __launch_bounds__(256)
k_touch(fbbev_touch_spans sp, int n) {
    fbbev_v4u acc = {0u, 0u, 0u, 0u};
    for (int k = 0; k < n; ++k) {
        const fbbev_v4u* q = reinterpret_cast<const fbbev_v4u*>(sp.p[k]);
        for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < sp.n16[k]; i += (unsigned long long)gridDim.x * 256) {
            const fbbev_v4u v = q[i];
            acc[0] |= v[0]; acc[1] |= v[1]; acc[2] |= v[2]; acc[3] |= v[3];
        }
    }
    fbbev_opaque_u32(acc[0] | acc[1] | acc[2] | acc[3]);                                 // the loads are used; nothing is stored
}
extern "C" int fbbev_touch(const void* const* spans, const size_t* bytes, int n, fbbev_stream_t stream_) {
    if (n < 0 || n > 8 || (n > 0 && (!spans || !bytes))) return FBBEV_E_BADARG;
    fbbev_touch_spans sp;
    unsigned long long total = 0;
    int m = 0;
    for (int k = 0; k < n; ++k) {
        if (!spans[k] || bytes[k] < 16) continue;
        const uintptr_t a = (reinterpret_cast<uintptr_t>(spans[k]) + 15u) & ~(uintptr_t)15u;      // whole 16-byte pieces inside the span
        const size_t skip = a - reinterpret_cast<uintptr_t>(spans[k]);
        sp.p[m] = reinterpret_cast<const char*>(a);
        sp.n16[m] = (bytes[k] - skip) / 16;
        total += sp.n16[m];
        ++m;
    }
    if (m == 0 || total == 0) return 0;
    unsigned long long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    FBBEV_LAUNCH(k_touch<0>, blocks, 256, 0, (fbbev_rt_stream)stream_, sp, m);
    FBBEV_CHECK_LAUNCH();
    return 0;
}

// softmax over groups of `group` consecutive floats and its backward (the attention weights of the deformable attentions, training)
template <bool BWD>
static int softmax_groups_launch(const float* a, const float* b, long long n_groups, int group, float* out, fbbev_stream_t stream_) {
    if (n_groups < 0 || group <= 0) return FBBEV_E_BADARG;
    if (n_groups == 0) return 0;
    if (!a || !out || (BWD && !b)) return FBBEV_E_BADARG;
    if (!(group == 4 || group == 8 || group == 16 || group == 32) || !aligned16(a) || !aligned16(out) || (b && !aligned16(b)))
        return FBBEV_E_UNSUPPORTED;
    const long long n4 = n_groups * (group / 4), blocks = (n4 + 255) / 256;
    if (blocks >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    fbbev_rt_stream stream = (fbbev_rt_stream)stream_;
    switch (group) {
    case 4: FBBEV_LAUNCH((k_softmax_groups<4, BWD>), blocks, 256, 0, stream, a, b, n4, out); break;
    case 8: FBBEV_LAUNCH((k_softmax_groups<8, BWD>), blocks, 256, 0, stream, a, b, n4, out); break;
    case 16: FBBEV_LAUNCH((k_softmax_groups<16, BWD>), blocks, 256, 0, stream, a, b, n4, out); break;
    default: FBBEV_LAUNCH((k_softmax_groups<32, BWD>), blocks, 256, 0, stream, a, b, n4, out); break;
    }
    FBBEV_CHECK_LAUNCH();
    return 0;
}
extern "C" int fbbev_softmax_groups(const float* x, long long n_groups, int group, float* y, fbbev_stream_t stream_) {
    return softmax_groups_launch<false>(x, nullptr, n_groups, group, y, stream_);
}
extern "C" int fbbev_softmax_groups_bwd(const float* y, const float* grad_y, long long n_groups, int group, float* grad_x,
                                        fbbev_stream_t stream_) {
    return softmax_groups_launch<true>(y, grad_y, n_groups, group, grad_x, stream_);
}
