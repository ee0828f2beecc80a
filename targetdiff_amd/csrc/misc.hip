// Small per-ligand-atom kernels of one reverse-diffusion step (gfx950):
//   * head_kernel       v_inference = Linear -> ShiftedSoftplus -> Linear on ligand rows
//                       (models/molopt_score_model.py:307-311,351-352; models/common.py:156-162)
//   * posterior_kernel  Gaussian + categorical posterior and Gumbel-max draw
//                       (models/molopt_score_model.py:673-685 and the helpers cited in targetdiff_hip.h)
//   * center kernels    center_pos(mode='protein') (models/molopt_score_model.py:110-120)
// The reference spends ~40 tiny launches and two host syncs per step here; each is one launch.
#include "td_device.h"
#include "td_internal.h"

constexpr int HEAD_ATOMS = 8;

__global__ __launch_bounds__(128) void head_kernel(TdHead hd, const float *__restrict__ h,
                                                   const float4 *__restrict__ x4,
                                                   const int32_t *__restrict__ lig_node, int64_t Nl, int C,
                                                   float *__restrict__ pred_pos, float *__restrict__ pred_v,
                                                   float *__restrict__ lig_h) {
    __shared__ float s_h[HEAD_ATOMS][TD_H];
    __shared__ float s_y[HEAD_ATOMS][TD_H];
    const int n = threadIdx.x;
    const int64_t a0 = (int64_t)blockIdx.x * HEAD_ATOMS;
    for (int a = 0; a < HEAD_ATOMS; ++a) {
        const int64_t at = a0 + a;
        float v = 0.f;
        if (at < Nl) {
            const int64_t p = lig_node ? (int64_t)lig_node[at] : at;     // nullptr: free-standing rows (return_all)
            v = h[p * TD_H + n];
            if (lig_h) lig_h[at * TD_H + n] = v;
            if (n < 3 && x4) {
                const float4 xp = x4[p];
                pred_pos[at * 3 + n] = n == 0 ? xp.x : (n == 1 ? xp.y : xp.z);
            }
        }
        s_h[a][n] = v;
    }
    __syncthreads();
    float acc[HEAD_ATOMS];
    const float b0 = hd.b0[n];
#pragma unroll
    for (int a = 0; a < HEAD_ATOMS; ++a) acc[a] = b0;
#pragma unroll 16        // sixteen weight loads in flight
    for (int k = 0; k < TD_H; ++k) {
        const float wv = hd.W0T[k * TD_H + n];
#pragma unroll
        for (int a = 0; a < HEAD_ATOMS; ++a) acc[a] = fmaf(wv, s_h[a][k], acc[a]);
    }
#pragma unroll
    for (int a = 0; a < HEAD_ATOMS; ++a) {
        const float xv = acc[a];
        const float sp = xv > 20.f ? xv : log1pf(expf(xv));          // F.softplus (beta 1, threshold 20)
        s_y[a][n] = sp - 0.69314718055994531f;                      // ShiftedSoftplus: - log(2)
    }
    __syncthreads();
    // second Linear: 128 -> C (C <= 16).  thread (a, cc) = (n / 16, n % 16)
    const int a = n >> 4, cc = n & 15;
    if (cc < C && a0 + a < Nl) {
        float o = hd.b2[cc];
#pragma unroll 16
        for (int k = 0; k < TD_H; ++k) o = fmaf(hd.W2T[k * TD_MAXC + cc], s_y[a][k], o);
        pred_v[(a0 + a) * C + cc] = o;
    }
}

int td_launch_head(const TdHead &hd, const float *h, const float4 *x4, const int32_t *lig_node, int64_t Nl,
                   int classes, float *pred_pos, float *pred_v, float *lig_h, hipStream_t s) {
    if (Nl == 0) return TD_OK;
    head_kernel<<<dim3((unsigned)((Nl + HEAD_ATOMS - 1) / HEAD_ATOMS)), dim3(128), 0, s>>>(
        hd, h, x4, lig_node, Nl, classes, pred_pos, pred_v, lig_h);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

// ------------------------------------------------------------------------------------------ posterior
__device__ __forceinline__ float td_log_add_exp(float a, float b) {      // molopt_score_model.py:173-175
    const float m = fmaxf(a, b);
    return m + logf(expf(a - m) + expf(b - m));
}

__device__ __forceinline__ int td_find_graph_l(const int32_t *__restrict__ ptr, int B, int i) {
    int lo = 0, hi = B;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (ptr[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

// one ligand atom of the posterior update.  pos / v may alias pos_next / v_next (the in-place form of td_session_step): every input of
// the atom is read before its outputs are written, and atoms do not read each other.  pos_cur / v_cur (optional): second copies
// of x_{t-1} / v_{t-1} (the trajectory slot and the current state of td_session_step); v_frozen: pos_only, v_next = the input type.
__device__ __forceinline__ void td_posterior_atom(const TdSchedules &sc, int T, const int32_t *__restrict__ tg,
                                                  const int32_t *__restrict__ lptr, int B, int C, int64_t at,
                                                  const float *pos, const int64_t *v,
                                                  const float *__restrict__ pred_pos, const float *__restrict__ pred_v,
                                                  const float *__restrict__ noise, const float *__restrict__ uni,
                                                  float *pos_next, int64_t *v_next,
                                                  float *__restrict__ log_v0_out, float *__restrict__ log_post_out,
                                                  float *pos_cur = nullptr, int64_t *v_cur = nullptr, bool v_frozen = false,
                                                  int mean_type = 0) {
    const int g = td_find_graph_l(lptr, B, (int)at);
    int t = tg[g];
    t = t < 0 ? 0 : (t >= T ? T - 1 : t);
    // ---- positions: mean = c0[t] x0 + ct[t] x_t ; x_{t-1} = mean + [t != 0] exp(0.5 logvar[t]) eps  (:673-679)
    const float c0 = sc.c0[t], ct = sc.ct[t];
    const float sd = t == 0 ? 0.f : expf(0.5f * sc.logvar[t]);
    float xn[3];
#pragma unroll
    for (int d = 0; d < 3; ++d)       // three products, two sums, each rounded on its own -- PyTorch's eager arithmetic (:376, :679), and the
                                      // same bits in every kernel this function is inlined into (no compiler-chosen FMA contraction)
    {
        const float xt = pos[at * 3 + d];
        float x0 = pred_pos[at * 3 + d];
        // model_mean_type 'noise' (:412-416, :663-666): the network's output is x_t + eps; x0 = rc[t] x_t - rm1[t] eps
        if (mean_type == 1) x0 = td_add_rn(td_mul_rn(sc.rc[t], xt), -td_mul_rn(sc.rm1[t], td_add_rn(x0, -xt)));
        xn[d] = td_add_rn(td_add_rn(td_mul_rn(c0, x0), td_mul_rn(ct, xt)), td_mul_rn(sd, noise[at * 3 + d]));
    }
    const int vt = (int)v[at];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        pos_next[at * 3 + d] = xn[d];
        if (pos_cur) pos_cur[at * 3 + d] = xn[d];
    }
    // ---- types (:682-685)
    float lg[TD_MAXC];
    float mx = -INFINITY;
#pragma unroll
    for (int cc = 0; cc < TD_MAXC; ++cc) {
        lg[cc] = cc < C ? pred_v[at * C + cc] : -INFINITY;
        mx = fmaxf(mx, lg[cc]);
    }
    float se = 0.f;
#pragma unroll
    for (int cc = 0; cc < TD_MAXC; ++cc) se += cc < C ? expf(lg[cc] - mx) : 0.f;
    const float lse = mx + logf(se);
    const int tm1 = t - 1 < 0 ? 0 : t - 1;
    const float lnK = logf((float)C);
    const float l_ca = sc.log_ca[tm1], l_1mca = sc.log_1mca[tm1] - lnK;
    const float l_a = sc.log_a[t], l_1ma = sc.log_1ma[t] - lnK;
    const float LOG_EPS = logf(1e-30f);                         // log(clamp(onehot, 1e-30)), :129
    float un[TD_MAXC];
    float umx = -INFINITY;
#pragma unroll
    for (int cc = 0; cc < TD_MAXC; ++cc) {
        if (cc < C) {
            const float lv0 = lg[cc] - lse;                     // log_softmax
            lg[cc] = lv0;
            const float lvt = cc == vt ? 0.f : LOG_EPS;
            un[cc] = td_log_add_exp(lv0 + l_ca, l_1mca) + td_log_add_exp(lvt + l_a, l_1ma);
            umx = fmaxf(umx, un[cc]);
        } else {
            un[cc] = -INFINITY;
        }
    }
    float us = 0.f;
#pragma unroll
    for (int cc = 0; cc < TD_MAXC; ++cc) us += cc < C ? expf(un[cc] - umx) : 0.f;
    const float ulse = umx + logf(us);
    int best = 0;
    float bestv = -INFINITY;
#pragma unroll
    for (int cc = 0; cc < TD_MAXC; ++cc) {
        if (cc < C) {
            const float lp = un[cc] - ulse;
            if (log_v0_out) log_v0_out[at * C + cc] = lg[cc];
            if (log_post_out) log_post_out[at * C + cc] = lp;
            const float gum = -logf(-logf(uni[at * C + cc] + 1e-30f) + 1e-30f);     // :160-166
            const float sc2 = gum + lp;
            if (sc2 > bestv) { bestv = sc2; best = cc; }        // first maximum, like argmax
        }
    }
    if (v_frozen) best = vt;
    v_next[at] = best;
    if (v_cur) v_cur[at] = best;
}

__global__ void posterior_kernel(TdSchedules sc, int T, const int32_t *__restrict__ tg,
                                 const int32_t *__restrict__ lptr, int64_t Nl, int B, int C,
                                 const float *__restrict__ pos, const int64_t *__restrict__ v,
                                 const float *__restrict__ pred_pos, const float *__restrict__ pred_v,
                                 const float *__restrict__ noise, const float *__restrict__ uni,
                                 float *__restrict__ pos_next, int64_t *__restrict__ v_next,
                                 float *__restrict__ log_v0_out, float *__restrict__ log_post_out, int mean_type) {
    const int64_t at = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (at >= Nl) return;
    td_posterior_atom(sc, T, tg, lptr, B, C, at, pos, v, pred_pos, pred_v, noise, uni, pos_next, v_next, log_v0_out, log_post_out,
                      nullptr, nullptr, false, mean_type);
}

// td_session_step: the same update with its per-step arguments taken from device memory -- step index s = step[0] selects the
// time-step row t_all[s] and slot s of the trajectories; the current state (pos / v) is updated in place.  The last workgroup
// to finish advances the step index (all workgroups have read it by then): the launch is replayable as a graph node.
__global__ void posterior_step_kernel(TdSchedules sc, int T, int32_t *__restrict__ step, const int32_t *__restrict__ t_all,
                                      int num_steps, const int32_t *__restrict__ lptr, int64_t Nl, int B, int C,
                                      float *pos, int64_t *v, const float *__restrict__ pred_pos,
                                      const float *__restrict__ pred_v, const float *__restrict__ noise,
                                      const float *__restrict__ uni, float *__restrict__ pos_traj, int64_t *__restrict__ v_traj,
                                      float *__restrict__ v0_traj, float *__restrict__ vt_traj, int pos_only, int mean_type) {
    int s = *reinterpret_cast<volatile int32_t *>(step);
    s = s < 0 ? 0 : (s >= num_steps ? num_steps - 1 : s);
    const int64_t at = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (at < Nl) {
        const size_t so = (size_t)s * (size_t)Nl;
        td_posterior_atom(sc, T, t_all + (size_t)s * B, lptr, B, C, at, pos, v, pred_pos, pred_v, noise, uni, pos_traj + so * 3,
                          v_traj + so, v0_traj ? v0_traj + so * C : nullptr, vt_traj ? vt_traj + so * C : nullptr, pos,
                          pos_only ? nullptr : v, pos_only != 0, mean_type);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(step + 1, 1) == (int)gridDim.x - 1) {
            step[1] = 0;
            atomicAdd(step, 1);
        }
    }
}


int td_launch_posterior(const TdSchedules &sc, int T, const int32_t *t, const int32_t *lptr, int64_t Nl, int64_t B,
                        int classes, const float *pos, const int64_t *v, const float *pred_pos,
                        const float *pred_v, const float *noise, const float *uni, float *pos_next,
                        int64_t *v_next, float *log_v0, float *log_post, hipStream_t s, int mean_type) {
    if (Nl == 0) return TD_OK;
    posterior_kernel<<<dim3((unsigned)((Nl + 127) / 128)), dim3(128), 0, s>>>(
        sc, T, t, lptr, Nl, (int)B, classes, pos, v, pred_pos, pred_v, noise, uni, pos_next, v_next, log_v0,
        log_post, mean_type);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

int td_launch_posterior_step(const TdSchedules &sc, int T, int32_t *step, const int32_t *t_all, int num_steps, const int32_t *lptr,
                             int64_t Nl, int64_t B, int classes, float *pos, int64_t *v, const float *pred_pos, const float *pred_v,
                             const float *noise, const float *uni, float *pos_traj, int64_t *v_traj, float *v0_traj, float *vt_traj,
                             int pos_only, hipStream_t s, int mean_type) {
    if (Nl == 0) return TD_OK;
    posterior_step_kernel<<<dim3((unsigned)((Nl + 127) / 128)), dim3(128), 0, s>>>(
        sc, T, step, t_all, num_steps, lptr, Nl, (int)B, classes, pos, v, pred_pos, pred_v, noise, uni, pos_traj, v_traj, v0_traj,
        vt_traj, pos_only, mean_type);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

// ------------------------------------------------------------------------------------------ centring
__global__ __launch_bounds__(256) void center_kernel(float *__restrict__ ppos, const int32_t *__restrict__ pptr,
                                                     float *__restrict__ lpos, const int32_t *__restrict__ lptr,
                                                     float *__restrict__ offset, int compute, float sign) {
    __shared__ float s_off[3];
    const int g = blockIdx.x;
    if (compute) {
        // scatter_mean (models/molopt_score_model.py:115): sum / clamp(count, 1).  The sum runs over the graph's atoms IN INDEX
        // ORDER, one thread per coordinate -- the order of the reference's CPU path (index_add_ walks the rows sequentially), so
        // the centred coordinates, and with them every near-tie of the k-NN search, come out bit-identical to it.  (A tree
        // reduction differs in the last bit of the offset: one neighbour flip in 1000 teacher-forced steps on 1h36, round 3.)
        if (threadIdx.x < 3) {
            const int b = pptr[g], e = pptr[g + 1];
            float sum = 0.f;
            for (int i = b; i < e; ++i) sum += ppos[3 * i + threadIdx.x];
            const int cnt = e - b;
            const float o = sum / (float)(cnt < 1 ? 1 : cnt);
            s_off[threadIdx.x] = o;
            offset[3 * g + threadIdx.x] = o;
        }
    } else if (threadIdx.x < 3) {
        s_off[threadIdx.x] = offset[3 * g + threadIdx.x];
    }
    __syncthreads();
    const float ox = sign * s_off[0], oy = sign * s_off[1], oz = sign * s_off[2];
    if (ppos) {
        for (int i = pptr[g] + threadIdx.x; i < pptr[g + 1]; i += blockDim.x) {
            ppos[3 * i] += ox; ppos[3 * i + 1] += oy; ppos[3 * i + 2] += oz;
        }
    }
    if (lpos) {
        for (int i = lptr[g] + threadIdx.x; i < lptr[g + 1]; i += blockDim.x) {
            lpos[3 * i] += ox; lpos[3 * i + 1] += oy; lpos[3 * i + 2] += oz;
        }
    }
}

int td_launch_center(float *ppos, const int32_t *pptr, float *lpos, const int32_t *lptr, int64_t B, float *offset,
                     int compute, int sign, hipStream_t s) {
    if (B == 0) return TD_OK;
    center_kernel<<<dim3((unsigned)B), dim3(256), 0, s>>>(ppos, pptr, lpos, lptr, offset, compute,
                                                         sign < 0 ? -1.f : 1.f);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

// ------------------------------------------------------------------------------------------ test hook
__global__ void reductions_kernel(const float *__restrict__ in, float *__restrict__ out) {
    const int l = threadIdx.x;
    const float v = in[l];
    out[0 * 64 + l] = td_sum8(v);
    out[1 * 64 + l] = td_sum32(v);
    out[2 * 64 + l] = td_sum64(v);
    out[3 * 64 + l] = td_sum_halves(v);
    out[4 * 64 + l] = td_max_halves(v);
    out[5 * 64 + l] = td_swap32(v);
}
int td_launch_reductions(const float *in, float *out, hipStream_t s) {
    reductions_kernel<<<dim3(1), dim3(64), 0, s>>>(in, out);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}
