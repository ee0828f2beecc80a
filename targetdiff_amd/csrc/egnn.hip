// Standalone EGNN refine net (models/egnn.py) for gfx950, as get_refine_net('egnn', config) builds it
// (models/molopt_score_model.py:34-42): num_r_gaussian = 1, k-NN graph rebuilt on the current coordinates in every layer,
// SiLU, no LayerNorm.  One EnBaseLayer.forward (models/egnn.py:36-64) =
//     node_proj_kernel (node.hip)   P_i = W1[:, 0:128] h_i + b1,  P_j = W1[:, 128:256] h_j     (first Linear, node side)
//     egnn_edge_kernel              m_ij = SiLU(W2 SiLU(P_i + P_j + w_d d^2 + W_t[type]) + b2);  e_ij = sigmoid(w . m_ij + b);
//                                   mi = sum_j m_ij e_ij;  ligand rows: x_i += sum_j (x_i - x_j) / (sqrt(d^2 + 1e-8) + 1) tanh(x_mlp(m_ij))
//     egnn_node_kernel              h_i += W2n SiLU(W1n [mi | h_i] + b1n) + b2n
// The 261-wide first Linear is split like the attention layers' (node side once per node, d^2 / type terms per edge); the
// two 128 x 128 per-edge Linears run on v_mfma_f32_16x16x4_f32 with the activations as the B operand exactly as they sit
// in the accumulators (k-step (hb, r) pairs lane group g with hidden unit 16hb + 4g + r on both operands).
#include "td_device.h"
#include "td_internal.h"

typedef float floatx4_t __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ floatx4_t eg_mfma16(float a, float b, floatx4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float eg_silu(float v) { return v * __frcp_rn(1.0f + __expf(-v)); }
__device__ __forceinline__ float eg_sum16(float v) {          // over the 16 lanes of a DPP row
    v += td_dpp<DPP_QUAD_XOR1>(v);
    v += td_dpp<DPP_QUAD_XOR2>(v);
    v += td_dpp<DPP_ROW_HALF_MIRROR>(v);
    v += td_dpp<DPP_ROW_MIRROR>(v);
    return v;
}
__device__ __forceinline__ float eg_sum_groups(float v) { return td_sum_halves(td_sum_rows16(v)); }   // over the 4 lane groups

constexpr int EG_WAVES = 8;
constexpr int EG_WF = 8 * 8 * 64 * 4;          // one 128 x 128 matrix as A fragments: [ot][hb][lane] x 4 r
constexpr int EG_VEC = 128 + 512 + 128 + 132 + 128 + 128;    // wd, wt, b2, winf (+ binf), bx, wx2
constexpr size_t EG_LDS_BYTES = (size_t)(2 * EG_WF + EG_VEC) * sizeof(float);

// acc2[eb][ot] = W . act  (W: 128 x 128 as A fragments in LDS; act: acc[eb][hb][r] = act^T[hidden 16hb + 4g + r][edge 16eb + lo])
__device__ __forceinline__ void eg_gemm128(const float4 *__restrict__ Wf, int lane, const floatx4_t (&act)[2][8],
                                           floatx4_t (&out)[2][8]) {
#pragma unroll
    for (int eb = 0; eb < 2; ++eb)
#pragma unroll
        for (int ot = 0; ot < 8; ++ot) out[eb][ot] = floatx4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int hb = 0; hb < 8; ++hb) {
        float4 w[8];
#pragma unroll
        for (int ot = 0; ot < 8; ++ot) w[ot] = Wf[(ot * 8 + hb) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int ot = 0; ot < 8; ++ot) {
                const float av = r == 0 ? w[ot].x : (r == 1 ? w[ot].y : (r == 2 ? w[ot].z : w[ot].w));
                out[0][ot] = eg_mfma16(av, act[0][hb][r], out[0][ot]);
                out[1][ot] = eg_mfma16(av, act[1][hb][r], out[1][ot]);
            }
        }
    }
}

__global__ __launch_bounds__(EG_WAVES * 64) void egnn_edge_kernel(TdEgnnLayer L, const float4 *__restrict__ x4,
                                                                  float4 *__restrict__ x4_out, const int32_t *__restrict__ nbr,
                                                                  const float *__restrict__ P, float *__restrict__ mi,
                                                                  int64_t N) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const float4 *W2f = reinterpret_cast<const float4 *>(lds);
    const float4 *Wxf = reinterpret_cast<const float4 *>(lds + EG_WF);
    float *vec = lds + 2 * EG_WF;
    const float *wd = vec, *wt = vec + 128, *b2 = vec + 640, *winf = vec + 768, *bx = vec + 900, *wx2 = vec + 1028;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lo = lane & 15, g = lane >> 4;
    td_stage_lds16(reinterpret_cast<const float4 *>(L.W2f), reinterpret_cast<float4 *>(lds), EG_WF / 4, tid, EG_WAVES * 64);
    td_stage_lds16(reinterpret_cast<const float4 *>(L.Wxf), reinterpret_cast<float4 *>(lds + EG_WF), EG_WF / 4, tid, EG_WAVES * 64);
    for (int u = tid; u < EG_VEC; u += EG_WAVES * 64) vec[u] = L.vec[u];
    __syncthreads();
    const float binf = winf[128];

    const int G = gridDim.x;
    const int64_t per = (N + G - 1) / G;
    const int64_t begin = (int64_t)blockIdx.x * per, end = begin + per < N ? begin + per : N;
    for (int64_t i = begin + wid; i < end; i += EG_WAVES) {
        const float4 xi = x4[i];
        const bool dst_l = xi.w > 0.5f;
        floatx4_t act[2][8];
        bool valid[2];
        float d2[2], rel[2][3];
        int type[2];
#pragma unroll
        for (int eb = 0; eb < 2; ++eb) {
            const int j = nbr[i * TD_K + 16 * eb + lo];
            valid[eb] = j >= 0;
            const int jj = valid[eb] ? j : (int)i;
            const float4 xj = x4[jj];
            const float rx = xi.x - xj.x, ry = xi.y - xj.y, rz = xi.z - xj.z;
            rel[eb][0] = rx; rel[eb][1] = ry; rel[eb][2] = rz;
            d2[eb] = (rx * rx + ry * ry) + rz * rz;                                   // models/egnn.py:41
            const bool src_l = xj.w > 0.5f;
            type[eb] = src_l ? (dst_l ? 0 : 1) : (dst_l ? 2 : 3);                     // :105-118
            const float *pj = P + (size_t)jj * (4 * TD_H) + TD_H + 4 * g;             // P_j: hidden 16hb + 4g .. + 3
#pragma unroll
            for (int hb = 0; hb < 8; ++hb) {
                const float4 v = *reinterpret_cast<const float4 *>(pj + 16 * hb);
                act[eb][hb][0] = v.x; act[eb][hb][1] = v.y; act[eb][hb][2] = v.z; act[eb][hb][3] = v.w;
            }
        }
        // ---- first Linear (edge side) + SiLU:  P_i + P_j + w_d d^2 + W_t[type] ------------------------------------------
#pragma unroll
        for (int hb = 0; hb < 8; ++hb) {
            const float4 pi = *reinterpret_cast<const float4 *>(P + (size_t)i * (4 * TD_H) + 16 * hb + 4 * g);
            const float4 wdv = *reinterpret_cast<const float4 *>(wd + 16 * hb + 4 * g);
#pragma unroll
            for (int eb = 0; eb < 2; ++eb) {
                const float4 tv = *reinterpret_cast<const float4 *>(wt + type[eb] * TD_H + 16 * hb + 4 * g);
                act[eb][hb][0] = eg_silu(fmaf(wdv.x, d2[eb], act[eb][hb][0] + pi.x) + tv.x);
                act[eb][hb][1] = eg_silu(fmaf(wdv.y, d2[eb], act[eb][hb][1] + pi.y) + tv.y);
                act[eb][hb][2] = eg_silu(fmaf(wdv.z, d2[eb], act[eb][hb][2] + pi.z) + tv.z);
                act[eb][hb][3] = eg_silu(fmaf(wdv.w, d2[eb], act[eb][hb][3] + pi.w) + tv.w);
            }
        }
        // ---- second Linear + SiLU (act_last): m_ij ------------------------------------------------------------------------
        floatx4_t m[2][8];
        eg_gemm128(W2f, lane, act, m);
        float part[2] = {0.f, 0.f};
#pragma unroll
        for (int ot = 0; ot < 8; ++ot) {
            const float4 bv = *reinterpret_cast<const float4 *>(b2 + 16 * ot + 4 * g);
            const float4 wv = *reinterpret_cast<const float4 *>(winf + 16 * ot + 4 * g);
#pragma unroll
            for (int eb = 0; eb < 2; ++eb) {
                m[eb][ot][0] = eg_silu(m[eb][ot][0] + bv.x);
                m[eb][ot][1] = eg_silu(m[eb][ot][1] + bv.y);
                m[eb][ot][2] = eg_silu(m[eb][ot][2] + bv.z);
                m[eb][ot][3] = eg_silu(m[eb][ot][3] + bv.w);
                part[eb] = fmaf(wv.x, m[eb][ot][0], part[eb]);
                part[eb] = fmaf(wv.y, m[eb][ot][1], part[eb]);
                part[eb] = fmaf(wv.z, m[eb][ot][2], part[eb]);
                part[eb] = fmaf(wv.w, m[eb][ot][3], part[eb]);
            }
        }
        // ---- e_ij = sigmoid(edge_inf(m_ij));  mi = sum_j m_ij e_ij  (:52-53) -----------------------------------------------
        float e[2];
#pragma unroll
        for (int eb = 0; eb < 2; ++eb) {
            const float s = eg_sum_groups(part[eb]) + binf;
            e[eb] = valid[eb] ? __frcp_rn(1.0f + __expf(-s)) : 0.f;
        }
#pragma unroll
        for (int ot = 0; ot < 8; ++ot) {
            float4 o;
            o.x = eg_sum16(m[0][ot][0] * e[0] + m[1][ot][0] * e[1]);
            o.y = eg_sum16(m[0][ot][1] * e[0] + m[1][ot][1] * e[1]);
            o.z = eg_sum16(m[0][ot][2] * e[0] + m[1][ot][2] * e[1]);
            o.w = eg_sum16(m[0][ot][3] * e[0] + m[1][ot][3] * e[1]);
            if (lo == 0) *reinterpret_cast<float4 *>(mi + (size_t)i * TD_H + 16 * ot + 4 * g) = o;
        }
        // ---- coordinate update, ligand rows only (:57-62: the result is masked for protein rows) ------------------------
        if (dst_l) {
            eg_gemm128(Wxf, lane, m, act);                        // x_mlp.0 on m_ij (act registers are free again)
            float px[2] = {0.f, 0.f};
#pragma unroll
            for (int ot = 0; ot < 8; ++ot) {
                const float4 bv = *reinterpret_cast<const float4 *>(bx + 16 * ot + 4 * g);
                const float4 wv = *reinterpret_cast<const float4 *>(wx2 + 16 * ot + 4 * g);
#pragma unroll
                for (int eb = 0; eb < 2; ++eb) {
                    px[eb] = fmaf(wv.x, eg_silu(act[eb][ot][0] + bv.x), px[eb]);
                    px[eb] = fmaf(wv.y, eg_silu(act[eb][ot][1] + bv.y), px[eb]);
                    px[eb] = fmaf(wv.z, eg_silu(act[eb][ot][2] + bv.z), px[eb]);
                    px[eb] = fmaf(wv.w, eg_silu(act[eb][ot][3] + bv.w), px[eb]);
                }
            }
            float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
            for (int eb = 0; eb < 2; ++eb) {
                const float s = tanhf(eg_sum_groups(px[eb]));
                const float cf = valid[eb] ? s / (sqrtf(d2[eb] + 1e-8f) + 1.0f) : 0.f;
                sx = fmaf(rel[eb][0], cf, sx);
                sy = fmaf(rel[eb][1], cf, sy);
                sz = fmaf(rel[eb][2], cf, sz);
            }
            sx = eg_sum16(sx); sy = eg_sum16(sy); sz = eg_sum16(sz);
            if (lane == 0) x4_out[i] = make_float4(xi.x + sx, xi.y + sy, xi.z + sz, xi.w);
        }
    }
}

}  // namespace

int td_launch_egnn_edge(const TdEgnnLayer &L, const float4 *x4, float4 *x4_out, const int32_t *nbr, const float *P, float *mi,
                        int64_t N, hipStream_t s) {
    if (N == 0) return TD_OK;
    {
        static TdLdsOnce once;
        int rc = td_set_lds(once, reinterpret_cast<const void *>(egnn_edge_kernel), EG_LDS_BYTES);
        if (rc != TD_OK) return rc;
    }
    int64_t g = (N + EG_WAVES - 1) / EG_WAVES;
    if (g > 256) g = 256;
    egnn_edge_kernel<<<dim3((unsigned)g), dim3(EG_WAVES * 64), EG_LDS_BYTES, s>>>(L, x4, x4_out, nbr, P, mi, N);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}
