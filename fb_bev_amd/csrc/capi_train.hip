// capi_train.hip -- extern "C" entry points of the training path added in round 6 (include/fbbev.h): their own translation unit of
// libfbbev_hip.so, so that an edit of these kernels recompiles in seconds; the CPU emulator build includes this file from capi.hip.
#include <stdio.h>
#include <stdlib.h>
#include "rt.h"
#include "capi_common.h"
#include "wgrad_kernels.h"
#include "../../include/fbbev.h"

// ------------------------------------------------------------------------------ weight / bias gradient of a row-wise linear layer
struct wgrad_plan { int nti, n_oc, n_ic, ksteps, kps, n_split; size_t part_w, part_b, total; };

static bool wgrad_plan_make(long long rows, int I, int O, wgrad_plan* p) {
    if (rows <= 0 || I <= 0 || O <= 0 || I % 4 != 0 || O % 4 != 0) return false;
    const long long ks = (rows + 31) / 32;
    if (ks >= (1ll << 30)) return false;
    p->nti = I <= 80 ? 5 : 8;
    p->n_oc = (O + 127) / 128;
    p->n_ic = (I + 16 * p->nti - 1) / (16 * p->nti);
    p->ksteps = (int)ks;
    // enough workgroups to fill the chip twice, few enough partial results that the fixed-order reduction stays small
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

extern "C" int fbbev_rows_wgrad_x3(const float* grad_out, long long ld_grad, const float* x, long long ldx, long long rows,
                                   int in_features, int out_features, float* grad_weight, float* grad_bias, void* workspace,
                                   size_t workspace_bytes, fbbev_stream_t stream_) {
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
    wgrad_plan p;
    if (!wgrad_plan_make(rows, I, O, &p) || ld_grad % 4 != 0 || ldx % 4 != 0 || !aligned16(grad_out) || !aligned16(x))
        return FBBEV_E_UNSUPPORTED;
    if (!workspace || !aligned16(workspace) || workspace_bytes < p.total) return FBBEV_E_WORKSPACE;
    float* part_w = reinterpret_cast<float*>(static_cast<char*>(workspace) + p.part_w);
    float* part_b = grad_bias ? reinterpret_cast<float*>(static_cast<char*>(workspace) + p.part_b) : nullptr;
    const long long wgs = (long long)p.n_split * p.n_oc * p.n_ic;
    if (wgs >= (1ll << 31)) return FBBEV_E_UNSUPPORTED;
    const size_t lds = (size_t)2 * (8 + p.nti) * FBBEV_WG_TILE_DW * 4;
    if (p.nti == 5) {
        FBBEV_LAUNCH(k_rows_wgrad_x3<5>, wgs, 256, lds, stream, grad_out, ld_grad, x, ldx, rows, O, I, p.n_oc, p.n_ic, p.ksteps,
                     p.kps, part_w, part_b);
    } else {
        if (lds > 64 * 1024) {
            const int e = fbbev_rt_allow_dyn_lds((const void*)k_rows_wgrad_x3<8>, lds);
            if (e) return e;
        }
        FBBEV_LAUNCH(k_rows_wgrad_x3<8>, wgs, 256, lds, stream, grad_out, ld_grad, x, ldx, rows, O, I, p.n_oc, p.n_ic, p.ksteps,
                     p.kps, part_w, part_b);
    }
    FBBEV_CHECK_LAUNCH();
    const long long OI = (long long)O * I, total = OI + (grad_bias ? O : 0);
    FBBEV_LAUNCH(k_rows_wgrad_reduce<0>, (total + 31) / 32, 256, 0, stream, (const float*)part_w, (const float*)part_b, p.n_split, OI,
                 O, grad_weight, grad_bias);
    FBBEV_CHECK_LAUNCH();
    return 0;
}
