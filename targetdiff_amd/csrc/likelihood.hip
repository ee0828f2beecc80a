// Kernels of the other `forward` consumers of the reference (gfx950):
//   * perturb_kernel           forward-process sample x_t, v_t   (models/molopt_score_model.py:577-588, q_v_sample :394-398)
//   * likelihood_terms_kernel  per-graph KL / decoder-NLL terms  (:594-613, compute_pos_Lt :463-474, compute_v_Lt :476-483)
//   * likelihood_prior_kernel  prior KL terms                    (:569-576, kl_pos_prior :430-438, kl_v_prior :410-416)
//   * embed_ligand_kernel / v_inference on free-standing rows    (return_all: :360-367)
// One wave per graph for the reductions (scatter_mean over a graph's ligand atoms): deterministic, no atomics.
#include "td_device.h"
#include "td_internal.h"

namespace {

__device__ __forceinline__ float lk_log_add_exp(float a, float b) {      // molopt_score_model.py:173-175
    const float m = fmaxf(a, b);
    return m + logf(expf(a - m) + expf(b - m));
}

__device__ __forceinline__ int lk_find_graph(const int32_t *ptr, int B, int at) {
    int lo = 0, hi = B;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (ptr[mid] <= at) lo = mid; else hi = mid;
    }
    return lo;
}

// q(v_{t-1} | v_t, v_0) in log space (:401-409) for one atom; log_v0 / out: [C] registers
__device__ __forceinline__ void lk_q_v_posterior(const TdSchedules &sc, int t, int C, const float (&log_v0)[TD_MAXC],
                                                 int vt, float (&out)[TD_MAXC]) {
    const int tm1 = t - 1 < 0 ? 0 : t - 1;
    const float lnK = logf((float)C);
    const float l_ca = sc.log_ca[tm1], l_1mca = sc.log_1mca[tm1] - lnK;
    const float l_a = sc.log_a[t], l_1ma = sc.log_1ma[t] - lnK;
    const float LOG_EPS = logf(1e-30f);
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < TD_MAXC; ++c) {
        if (c < C) {
            const float lvt = c == vt ? 0.f : LOG_EPS;
            out[c] = lk_log_add_exp(log_v0[c] + l_ca, l_1mca) + lk_log_add_exp(lvt + l_a, l_1ma);
            mx = fmaxf(mx, out[c]);
        } else {
            out[c] = -INFINITY;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < TD_MAXC; ++c) s += c < C ? expf(out[c] - mx) : 0.f;
    const float lse = mx + logf(s);
#pragma unroll
    for (int c = 0; c < TD_MAXC; ++c) out[c] -= lse;
}

__global__ void perturb_kernel(TdSchedules sc, int T, const int32_t *__restrict__ tg, const int32_t *__restrict__ lptr,
                               int64_t Nl, int B, int C, const float *__restrict__ pos, const int64_t *__restrict__ v,
                               const float *__restrict__ noise, const float *__restrict__ uni,
                               float *__restrict__ pos_t, int64_t *__restrict__ v_t) {
    const int64_t at = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (at >= Nl) return;
    const int g = lk_find_graph(lptr, B, (int)at);
    int t = tg[g];
    t = t < 0 ? 0 : (t >= T ? T - 1 : t);
    const float a = sc.abar[t];
    const float sa = sqrtf(a), sb = sqrtf(1.0f - a);
#pragma unroll
    for (int d = 0; d < 3; ++d) pos_t[at * 3 + d] = sa * pos[at * 3 + d] + sb * noise[at * 3 + d];
    const float lnK = logf((float)C);
    const float l_ca = sc.log_ca[t], l_1mca = sc.log_1mca[t] - lnK;
    const float LOG_EPS = logf(1e-30f);
    const int v0 = (int)v[at];
    int best = 0;
    float bestv = -INFINITY;
    for (int c = 0; c < C; ++c) {
        const float lq = lk_log_add_exp((c == v0 ? 0.f : LOG_EPS) + l_ca, l_1mca);
        const float gum = -logf(-logf(uni[at * C + c] + 1e-30f) + 1e-30f);
        const float s = gum + lq;
        if (s > bestv) { bestv = s; best = c; }
    }
    v_t[at] = best;
}

__global__ __launch_bounds__(64) void likelihood_terms_kernel(TdSchedules sc, int T, const int32_t *__restrict__ tg,
                                                              const int32_t *__restrict__ lptr, int C,
                                                              const float *__restrict__ x0, const float *__restrict__ xt,
                                                              const int64_t *__restrict__ v0, const int64_t *__restrict__ vt,
                                                              const float *__restrict__ pred_pos,
                                                              const float *__restrict__ pred_v,
                                                              float *__restrict__ kl_pos, float *__restrict__ kl_v) {
    const int g = blockIdx.x, lane = threadIdx.x;
    int t = tg[g];
    t = t < 0 ? 0 : (t >= T ? T - 1 : t);
    const float c0 = sc.c0[t], ct = sc.ct[t], logvar = sc.logvar[t];
    const bool decoder = t == 0;
    const int b = lptr[g], e = lptr[g + 1];
    float acc_pos = 0.f, acc_v = 0.f;
    const float LOG_EPS = logf(1e-30f);
    for (int at = b + lane; at < e; at += 64) {
        // ---- positions: normal_kl(true_mean, logvar, model_mean, logvar) / ln 2, or -log_normal(x0; model_mean) at t = 0
        float kl = 0.f, nll = 0.f;
        const float log_scales = 0.5f * logvar;
        const float var = expf(log_scales * 2.f);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float xd = xt[at * 3 + d], x0d = x0[at * 3 + d];
            const float mm = c0 * pred_pos[at * 3 + d] + ct * xd;       // q_pos_posterior(x0 = prediction)
            const float tm = c0 * x0d + ct * xd;                         // q_pos_posterior(x0 = data)
            const float df = tm - mm;
            kl += 0.5f * ((((-1.0f + logvar) - logvar) + expf(logvar - logvar)) + df * df * expf(-logvar));
            const float dv = x0d - mm;
            nll += (-(dv * dv) / (2.f * var) - log_scales) - 0.91893853320467274f;        // log(sqrt(2 pi))
        }
        kl = kl / 0.69314718055994531f;
        acc_pos += decoder ? -nll : kl;
        // ---- types
        float lg[TD_MAXC], lv0[TD_MAXC], pm[TD_MAXC], pt[TD_MAXC];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < TD_MAXC; ++c) {
            lg[c] = c < C ? pred_v[(int64_t)at * C + c] : -INFINITY;
            mx = fmaxf(mx, lg[c]);
        }
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < TD_MAXC; ++c) se += c < C ? expf(lg[c] - mx) : 0.f;
        const float lse = mx + logf(se);
        const int a0 = (int)v0[at], at_t = (int)vt[at];
#pragma unroll
        for (int c = 0; c < TD_MAXC; ++c) {
            lg[c] = lg[c] - lse;                                        // log_softmax(pred_v)
            lv0[c] = c == a0 ? 0.f : LOG_EPS;                           // index_to_log_onehot(v0)
        }
        lk_q_v_posterior(sc, t, C, lg, at_t, pm);                       // model posterior
        lk_q_v_posterior(sc, t, C, lv0, at_t, pt);                      // true posterior
        float klv = 0.f, nllv = 0.f;
#pragma unroll
        for (int c = 0; c < TD_MAXC; ++c) {
            if (c < C) {
                klv += expf(pt[c]) * (pt[c] - pm[c]);                   // categorical_kl(true, model)
                nllv += expf(lv0[c]) * pm[c];                           // log_categorical(log_v0, model)
            }
        }
        acc_v += decoder ? -nllv : klv;
    }
    acc_pos = td_sum64(acc_pos);
    acc_v = td_sum64(acc_v);
    if (lane == 0) {
        const int cnt = e - b < 1 ? 1 : e - b;
        kl_pos[g] = acc_pos / (float)cnt;
        kl_v[g] = acc_v / (float)cnt;
    }
}

__global__ __launch_bounds__(64) void likelihood_prior_kernel(TdSchedules sc, int T, const int32_t *__restrict__ lptr, int C,
                                                              const float *__restrict__ x0, const int64_t *__restrict__ vidx,
                                                              float *__restrict__ kl_pos, float *__restrict__ kl_v) {
    const int g = blockIdx.x, lane = threadIdx.x;
    const float a = sc.abar[T - 1];
    const float sa = sqrtf(a);
    const float logvar2 = logf(sqrtf(1.0f - a));
    const float lnK = logf((float)C);
    const float l_ca = sc.log_ca[T - 1], l_1mca = sc.log_1mca[T - 1] - lnK;
    const float LOG_EPS = logf(1e-30f);
    const int b = lptr[g], e = lptr[g + 1];
    float acc_pos = 0.f, acc_v = 0.f;
    for (int at = b + lane; at < e; at += 64) {
        float kl = 0.f;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float m2 = sa * x0[at * 3 + d];
            const float df = 0.f - m2;
            kl += 0.5f * ((((-1.0f + logvar2) - 0.f) + expf(0.f - logvar2)) + df * df * expf(-logvar2));
        }
        acc_pos += kl;
        const int vi = (int)vidx[at];
        float klv = 0.f;
        for (int c = 0; c < C; ++c) {
            const float lq = lk_log_add_exp((c == vi ? 0.f : LOG_EPS) + l_ca, l_1mca);
            klv += expf(lq) * (lq - (-lnK));
        }
        acc_v += klv;
    }
    acc_pos = td_sum64(acc_pos);
    acc_v = td_sum64(acc_v);
    if (lane == 0) {
        const int cnt = e - b < 1 ? 1 : e - b;
        kl_pos[g] = acc_pos / (float)cnt;
        kl_v[g] = acc_v / (float)cnt;
    }
}

// h = [Linear(one_hot(v)) ; 1]  (models/molopt_score_model.py:317,334,338)
__global__ __launch_bounds__(128) void embed_ligand_kernel(const float *__restrict__ WlT, const float *__restrict__ bl, int C,
                                                           const int64_t *__restrict__ lv, int64_t Nl, float *__restrict__ h) {
    const int n = threadIdx.x;
    const int64_t at = blockIdx.x;
    if (at >= Nl) return;
    int c = (int)lv[at];
    c = c < 0 ? 0 : (c >= C ? C - 1 : c);
    h[at * TD_H + n] = WlT[c * TD_H + n] + bl[n];       // column 127: weight 0, bias 1 (node indicator)
}

}  // namespace

int td_launch_perturb(const TdSchedules &sc, int T, const int32_t *t, const int32_t *lptr, int64_t Nl, int64_t B, int classes,
                      const float *pos, const int64_t *v, const float *noise, const float *uni, float *pos_t, int64_t *v_t,
                      hipStream_t s) {
    if (Nl == 0) return TD_OK;
    perturb_kernel<<<dim3((unsigned)((Nl + 127) / 128)), dim3(128), 0, s>>>(sc, T, t, lptr, Nl, (int)B, classes, pos, v, noise,
                                                                          uni, pos_t, v_t);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

int td_launch_likelihood_terms(const TdSchedules &sc, int T, const int32_t *t, const int32_t *lptr, int64_t B, int classes,
                               const float *x0, const float *xt, const int64_t *v0, const int64_t *vt, const float *pred_pos,
                               const float *pred_v, float *kl_pos, float *kl_v, hipStream_t s) {
    if (B == 0) return TD_OK;
    likelihood_terms_kernel<<<dim3((unsigned)B), dim3(64), 0, s>>>(sc, T, t, lptr, classes, x0, xt, v0, vt, pred_pos, pred_v,
                                                                  kl_pos, kl_v);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

int td_launch_likelihood_prior(const TdSchedules &sc, int T, const int32_t *lptr, int64_t B, int classes, const float *x0,
                               const int64_t *vidx, float *kl_pos, float *kl_v, hipStream_t s) {
    if (B == 0) return TD_OK;
    likelihood_prior_kernel<<<dim3((unsigned)B), dim3(64), 0, s>>>(sc, T, lptr, classes, x0, vidx, kl_pos, kl_v);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

int td_launch_embed_ligand(const TdEmbed &emb, int classes, const int64_t *lv, int64_t Nl, float *h, hipStream_t s) {
    if (Nl == 0) return TD_OK;
    embed_ligand_kernel<<<dim3((unsigned)Nl), dim3(128), 0, s>>>(emb.WlT, emb.bl, classes, lv, Nl, h);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}
