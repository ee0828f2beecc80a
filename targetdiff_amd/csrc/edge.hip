// Edge-side kernels of the equivariant attention stack (gfx950, fp32 MFMA 32x32x2):
//   * edge_gate_kernel  -- global edge gate e_w = sigmoid(MLP(20->128->1)(Gaussian(dist)))
//                          (models/uni_transformer.py:312-316, models/common.py:24-26,60-80)
//   * edge_attn_kernel  -- x2h  (BaseX2HAttLayer.forward, models/uni_transformer.py:42-84) and
//                          h2x  (BaseH2XAttLayer.forward, :108-140 + the masked update :205-206)
//
// Structure of edge_mlp_kernel (one wavefront per dst node, no inter-wave communication).
// The kNN graph gives every node exactly 32 in-edges, stored as a dense row nbr[i][0..31]; scatter_softmax /
// scatter_sum over dst (torch_scatter, :73,78,135,139) are therefore fixed-length segment reductions over ONE
// 32-row MFMA tile -- no atomics, no edge lists.  A workgroup (8 waves) stages the MLP's second Linear
// (64 KiB of B fragments) and the first Linear's radial/type tables (48 KiB) in LDS once, then every wave walks
// its own dst nodes.  Per node and MLP:
//   (1) first layer:   pre[e][n] = P_i[n] + P_j[j_e][n] + sum_m R[type_e][m][n] g_m(d_e)
//       = node projections (node.hip) gathered straight into the MFMA accumulators (C layout: lane = column,
//       register = row) + a 32 x (24|48) x 128 MFMA whose A operand (Gaussians of the edge length, masked by the
//       edge's source class) is built in registers,
//   (2) LayerNorm + ReLU in C layout (row statistics: DPP reductions over the 32 lanes of a half wave),
//   (3) C -> A layout through a wave-private 4.5 KiB LDS tile, one 32-column tile at a time,
//   (4) second layer:  64 k-steps x 4 N-tiles of MFMA, B fragments streamed from LDS (one ds_read_b128 per
//       4 MFMAs; 4 independent accumulator chains keep the matrix pipe issue-bound),
//   (5) epilogue.  Key MLP: logits = <q_i, k_e> per head (8-lane DPP sums), softmax over the 32 rows, times the
//       edge gate -> alpha[N][16][32].  Value MLP: out = sum_e alpha_e v_e, residual add (x2h) /
//       delta_x = mean_heads sum_e alpha_e v_e rel_e (h2x).
// Key and value MLPs run as separate launches (their weights would not fit LDS together); alpha (2 KiB per
// node) crosses through L2.  Compared with an N-split over waves this does every per-edge VALU operation
// (geometry, LayerNorm, softmax) exactly once instead of 4-8 times and needs no barriers.
#include <cstdlib>

#include "td_device.h"
#include "td_internal.h"

constexpr float TD_ATT_SCALE = 0.35355339059327373f;   // 1/sqrt(8)   (models/uni_transformer.py:73,135)

// ------------------------------------------------------------------------------------------ edge gate
// One wave per dst node, all 128 hidden units (4 N-tiles).  Pure register kernel: no LDS, no barriers.
__global__ __launch_bounds__(256, 2) void edge_gate_kernel(TdGate g, const float4 *__restrict__ x4,
                                                        const int32_t *__restrict__ nbr, int64_t N,
                                                        const int32_t *__restrict__ rows,
                                                        const int32_t *__restrict__ count_ptr,
                                                        float *__restrict__ ew) {
    if (count_ptr) N = *count_ptr;
    const int lane = threadIdx.x & 63;
    const int c = lane & 31, hi = lane >> 5;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;

    float4 R[TD_SLOT_STEPS];
    const float4 *Rp = reinterpret_cast<const float4 *>(g.R);
#pragma unroll
    for (int s = 0; s < TD_SLOT_STEPS; ++s) R[s] = Rp[s * 64 + lane];
    float b0[4], gam[4], bet[4], w3[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        b0[t] = g.b0[32 * t + c]; gam[t] = g.gamma[32 * t + c]; bet[t] = g.beta[32 * t + c]; w3[t] = g.w3[32 * t + c];
    }
    float offk[TD_SLOT_STEPS];
#pragma unroll
    for (int s = 0; s < TD_SLOT_STEPS; ++s) {
        const int k = td_kmap(s, hi);
        offk[s] = k < TD_NG ? g.offsets[k] : 0.f;
    }

    for (int64_t it = wave0; it < N; it += nwaves) {
        const int64_t i = rows ? (int64_t)rows[it] : it;
        const int j = nbr[i * TD_K + c];
        const bool valid = j >= 0;
        const float4 xi = x4[i];
        const float4 xj = x4[valid ? j : i];
        const float dx = xi.x - xj.x, dy = xi.y - xj.y, dz = xi.z - xj.z;
        const float d = sqrtf(dx * dx + dy * dy + dz * dz);
        floatx16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = b0[t];
#pragma unroll
        for (int s = 0; s < TD_SLOT_STEPS; ++s) {
            const int k = td_kmap(s, hi);
            const float u = d - offk[s];
            const float av = k < TD_NG ? expf(g.coeff * u * u) : 0.f;
            acc[0] = td_mfma(av, R[s].x, acc[0]);
            acc[1] = td_mfma(av, R[s].y, acc[1]);
            acc[2] = td_mfma(av, R[s].z, acc[2]);
            acc[3] = td_mfma(av, R[s].w, acc[3]);
        }
        float outv = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float s1 = (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]);
            const float mean = td_sum32(s1) * (1.0f / TD_H);
            const float d0 = acc[0][r] - mean, d1 = acc[1][r] - mean, d2 = acc[2][r] - mean, d3 = acc[3][r] - mean;
            const float var = td_sum32((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) * (1.0f / TD_H);
            const float rstd = 1.0f / sqrtf(var + 1e-5f);
            float part = fmaxf(d0 * rstd * gam[0] + bet[0], 0.f) * w3[0];
            part += fmaxf(d1 * rstd * gam[1] + bet[1], 0.f) * w3[1];
            part += fmaxf(d2 * rstd * gam[2] + bet[2], 0.f) * w3[2];
            part += fmaxf(d3 * rstd * gam[3] + bet[3], 0.f) * w3[3];
            const float logit = td_sum32(part) + g.b3;
            if (c == r) outv = 1.0f / (1.0f + expf(-logit));
        }
        // lane (c < 16, hi) holds the gate of edge row erow(c, hi)
        const int row = td_erow(c & 15, hi);
        const int jrow = __shfl(j, row);
        if (c < 16) ew[i * TD_K + row] = jrow >= 0 ? outv : 0.f;
    }
}

int td_launch_gate(const TdGate &g, const float4 *x4, const int32_t *nbr, int64_t N, const int32_t *rows,
                   const int32_t *count_ptr, float *ew, hipStream_t s) {
    if (N == 0) return TD_OK;
    int64_t blocks = (N + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    edge_gate_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(g, x4, nbr, N, rows, count_ptr, ew);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

// ------------------------------------------------------------------------------------------ x2h / h2x
enum { MODE_X2H_K = 0, MODE_X2H_V = 1, MODE_H2X_K = 2, MODE_H2X_V = 3 };

struct EdgeArgs {
    const float4 *x4;        // [N] (x, y, z, is_ligand)
    float4 *x4_out;          // h2x value pass: updated coordinates (ligand rows only are written)
    const int32_t *nbr;      // [N][32]
    const float *ew;         // [N][32] global edge gate
    const float *P;          // [N][512] node projections of this stage
    const float *q;          // [N][128] query vectors of this stage
    const int32_t *lig_node; // h2x: list of dst nodes
    float *h;                // x2h value pass: updated in place
    float *alpha;            // [N][16 heads][32 edges]: softmax weight * edge gate (key pass writes, value pass reads)
    int64_t count;           // number of dst nodes to process
    TdEdgeMlp mlp;
    const float *offsets;
    float coeff;
    int stagger_sleeps;      // initial delay of waves 4-7 in units of s_sleep(127) (~8k cycles)
    long long *dbg;          // TIMING variant: [8 waves][dbg_nodes][8] cycle stamps of workgroup 0
    int dbg_nodes;
};

constexpr int TB_STRIDE = 36;                                   // 32 + 4: conflict-free b128 reads of a 32-column tile
constexpr int LDS_W2_FLOATS = TD_KSTEPS * 64 * 4;               // 16384
constexpr int LDS_R_FLOATS = 2 * 2 * TD_SLOT_STEPS * 64 * 4;    // 12288
constexpr int LDS_TB_FLOATS = 8 * 32 * TB_STRIDE;               // 9216
constexpr size_t EDGE_LDS_BYTES = (size_t)(LDS_W2_FLOATS + LDS_R_FLOATS + LDS_TB_FLOATS + 8 * 32 * 4) * sizeof(float);

#define TD_STAMP(k)                                                                                    \
    do {                                                                                               \
        if (TIMING && blockIdx.x == 0 && lane == 0 && nodeno < a.dbg_nodes)                            \
            a.dbg[((size_t)wid * a.dbg_nodes + nodeno) * 8 + (k)] = clock64();                         \
    } while (0)

template <int MODE, bool TIMING>
__global__ __launch_bounds__(512) void edge_mlp_kernel(EdgeArgs a) {
    constexpr bool IS_K = MODE == MODE_X2H_K || MODE == MODE_H2X_K;
    constexpr bool IS_H2X = MODE == MODE_H2X_K || MODE == MODE_H2X_V;
    constexpr bool NARROW = MODE == MODE_H2X_V;                 // second Linear 128 -> 16: one (half-empty) N tile
    constexpr int NT2 = NARROW ? 1 : 4;                         // N tiles of the second layer
    constexpr int POFF = IS_K ? 0 : 2 * TD_H;                   // column offset of this MLP inside P: [k_i k_j v_i v_j]

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float4 *W2s = reinterpret_cast<float4 *>(lds);                                   // [64 kstep][64 lane] x 4 tiles
    float *W2n = lds;                                                                // NARROW: [64 kstep][64 lane]
    const float4 *Rs = reinterpret_cast<const float4 *>(lds + LDS_W2_FLOATS);        // [cls][slot][12][64] x 4 tiles
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int c = lane & 31, hi = lane >> 5;
    float *TB = lds + LDS_W2_FLOATS + LDS_R_FLOATS + wid * 32 * TB_STRIDE;            // wave-private transpose tile
    float *RELB = lds + LDS_W2_FLOATS + LDS_R_FLOATS + LDS_TB_FLOATS + wid * 32 * 4; // wave-private rel_x (h2x value)

    // ---- stage the weights in LDS (once per workgroup) ---------------------------------------------------------
    {
        const float4 *src = reinterpret_cast<const float4 *>(a.mlp.W2);
        const int n4 = NARROW ? TD_KSTEPS * 64 / 4 : TD_KSTEPS * 64;
        for (int idx = tid; idx < n4; idx += 512) W2s[idx] = src[idx];
        const float4 *rsrc = reinterpret_cast<const float4 *>(a.mlp.R);
        float4 *rdst = reinterpret_cast<float4 *>(lds + LDS_W2_FLOATS);
        for (int idx = tid; idx < LDS_R_FLOATS / 4; idx += 512) rdst[idx] = rsrc[idx];
    }
    // per-lane constants: this lane's 4 hidden columns 32t + c
    float gam[4], bet[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        gam[t] = a.mlp.gamma[32 * t + c];
        bet[t] = a.mlp.beta[32 * t + c];
    }
    float b2[NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t) b2[t] = NARROW ? (c < TD_HEADS ? a.mlp.b2[c] : 0.f) : a.mlp.b2[32 * t + c];
    float offk[TD_SLOT_STEPS];
#pragma unroll
    for (int s = 0; s < TD_SLOT_STEPS; ++s) {
        const int k = td_kmap(s, hi);
        offk[s] = k < TD_NG ? a.offsets[k] : 0.f;
    }
    __syncthreads();

    // ---- XCD-aware contiguous node ranges: workgroup b runs on XCD b % 8 -> give XCD x the x-th eighth --------
    const int G = gridDim.x;
    int chunk = blockIdx.x;
    if ((G & 7) == 0) chunk = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int64_t per = (a.count + G - 1) / G;
    const int64_t begin = (int64_t)chunk * per;
    const int64_t end = begin + per < a.count ? begin + per : a.count;

    // Waves w and w + 4 share a SIMD.  Identical work keeps them in lockstep (both in the 256-MFMA second layer,
    // then both in VALU phases with the matrix pipe idle); starting the second wave of each pair half a node
    // period late makes one wave's MFMA burst overlap the other's gather / LayerNorm / softmax.
    if (a.stagger_sleeps > 0 && wid >= 4)
        for (int k = 0; k < a.stagger_sleeps; ++k) __builtin_amdgcn_s_sleep(127);

    int nodeno = -1;
    for (int64_t it = begin + wid; it < end; it += 8) {
        ++nodeno;
        TD_STAMP(0);
        const int64_t i = IS_H2X ? (int64_t)a.lig_node[it] : it;
        // ---- geometry of the 32 in-edges: lane (c, hi) looks at edge c ---------------------------------------------
        const int j = a.nbr[i * TD_K + c];
        const bool valid = j >= 0;
        const float4 xi = a.x4[i];
        const float4 xj = a.x4[valid ? j : i];
        const float relx = xi.x - xj.x, rely = xi.y - xj.y, relz = xi.z - xj.z;       // x[dst] - x[src]
        const float d = sqrtf(relx * relx + rely * rely + relz * relz);
        const int slot = xj.w > 0.5f ? 0 : 1;          // source class: 0 ligand, 1 protein
        const int cls = xi.w > 0.5f ? 0 : 1;           // destination class (wave uniform)
        const bool has_a = __ballot(valid && slot == 0) != 0ull;
        const bool has_b = __ballot(valid && slot == 1) != 0ull;
        if (MODE == MODE_H2X_V && hi == 0) *reinterpret_cast<float4 *>(RELB + 4 * c) = make_float4(relx, rely, relz, 0.f);

        // ---- first layer: gather P_j straight into the accumulators (C layout rows 8q + 4hi + rr) ------------------
        floatx16 acc[4];
        unsigned vmask = 0;
        const float *Pj = a.P + POFF + TD_H + c;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int4 jv = *reinterpret_cast<const int4 *>(a.nbr + i * TD_K + 8 * qd + 4 * hi);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int jr = rr == 0 ? jv.x : rr == 1 ? jv.y : rr == 2 ? jv.z : jv.w;
                vmask |= jr >= 0 ? (1u << (4 * qd + rr)) : 0u;
                const float *row = Pj + (size_t)(jr >= 0 ? jr : (int)i) * (4 * TD_H);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t][4 * qd + rr] = row[32 * t];
            }
        }
        float pi[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) pi[t] = a.P[(size_t)i * (4 * TD_H) + POFF + 32 * t + c];
        TD_STAMP(1);
        float gv[TD_SLOT_STEPS];
#pragma unroll
        for (int s = 0; s < TD_SLOT_STEPS; ++s) {
            const int k = td_kmap(s, hi);
            const float u = d - offk[s];
            gv[s] = k < TD_NG ? __expf(a.coeff * u * u) : (k == TD_NG ? 1.f : 0.f);
        }
        if (has_a) {
            const bool on = valid && slot == 0;
            const float4 *Rp = Rs + (size_t)((cls * 2 + 0) * TD_SLOT_STEPS) * 64 + lane;
#pragma unroll
            for (int s = 0; s < TD_SLOT_STEPS; ++s) {
                const float4 b = Rp[s * 64];
                const float av = on ? gv[s] : 0.f;
                acc[0] = td_mfma(av, b.x, acc[0]);
                acc[1] = td_mfma(av, b.y, acc[1]);
                acc[2] = td_mfma(av, b.z, acc[2]);
                acc[3] = td_mfma(av, b.w, acc[3]);
            }
        }
        if (has_b) {
            const bool on = valid && slot == 1;
            const float4 *Rp = Rs + (size_t)((cls * 2 + 1) * TD_SLOT_STEPS) * 64 + lane;
#pragma unroll
            for (int s = 0; s < TD_SLOT_STEPS; ++s) {
                const float4 b = Rp[s * 64];
                const float av = on ? gv[s] : 0.f;
                acc[0] = td_mfma(av, b.x, acc[0]);
                acc[1] = td_mfma(av, b.y, acc[1]);
                acc[2] = td_mfma(av, b.z, acc[2]);
                acc[3] = td_mfma(av, b.w, acc[3]);
            }
        }

        TD_STAMP(2);
        // ---- LayerNorm + ReLU in C layout: row r of this half wave lives in acc[0..3][r] across 32 lanes -----------
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v0 = acc[0][r] + pi[0], v1 = acc[1][r] + pi[1], v2 = acc[2][r] + pi[2], v3 = acc[3][r] + pi[3];
            const float mean = td_sum32((v0 + v1) + (v2 + v3)) * (1.0f / TD_H);
            const float d0 = v0 - mean, d1 = v1 - mean, d2 = v2 - mean, d3 = v3 - mean;
            const float var = td_sum32((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) * (1.0f / TD_H);
            const float rstd = 1.0f / sqrtf(var + 1e-5f);
            acc[0][r] = fmaxf(fmaf(d0 * rstd, gam[0], bet[0]), 0.f);
            acc[1][r] = fmaxf(fmaf(d1 * rstd, gam[1], bet[1]), 0.f);
            acc[2][r] = fmaxf(fmaf(d2 * rstd, gam[2], bet[2]), 0.f);
            acc[3][r] = fmaxf(fmaf(d3 * rstd, gam[3], bet[3]), 0.f);
        }

        TD_STAMP(3);
        // ---- C layout -> A layout, one 32-column tile at a time (lane (e = c, hi) gets k = 8m + 4hi .. + 3) -------
        float4 az[16];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) TB[td_erow(r, hi) * TB_STRIDE + c] = acc[t][r];
#pragma unroll
            for (int mm = 0; mm < 4; ++mm)
                az[4 * t + mm] = *reinterpret_cast<const float4 *>(TB + c * TB_STRIDE + 8 * mm + 4 * hi);
        }

        TD_STAMP(4);
        // ---- second layer ------------------------------------------------------------------------------------------
        floatx16 o[NT2];
#pragma unroll
        for (int t = 0; t < NT2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] = b2[t];
#pragma unroll
        for (int s = 0; s < TD_KSTEPS; ++s) {
            const float4 am = az[s >> 2];
            const float av = (s & 3) == 0 ? am.x : (s & 3) == 1 ? am.y : (s & 3) == 2 ? am.z : am.w;
            if (NARROW) {
                o[0] = td_mfma(av, W2n[s * 64 + lane], o[0]);
            } else {
                const float4 b = W2s[s * 64 + lane];
                o[0] = td_mfma(av, b.x, o[0]);
                o[NT2 > 1 ? 1 : 0] = td_mfma(av, b.y, o[NT2 > 1 ? 1 : 0]);
                o[NT2 > 2 ? 2 : 0] = td_mfma(av, b.z, o[NT2 > 2 ? 2 : 0]);
                o[NT2 > 3 ? 3 : 0] = td_mfma(av, b.w, o[NT2 > 3 ? 3 : 0]);
            }
        }

        TD_STAMP(5);
        // ---- epilogue ----------------------------------------------------------------------------------------------
        if (IS_K) {
            // attention logits per head (8 columns = 8 lanes), softmax over the 32 in-edges (scatter_softmax),
            // multiplied by the global edge gate (v = v * e_w in the reference) -> alpha[i][head][edge]
            float4 ewq[4];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
                ewq[qd] = *reinterpret_cast<const float4 *>(a.ew + i * TD_K + 8 * qd + 4 * hi);
#pragma unroll
            for (int t = 0; t < NT2; ++t) {
                const float qn = a.q[(size_t)i * TD_H + 32 * t + c];
                float lg[16];
                float mx = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    lg[r] = ((vmask >> r) & 1u) ? td_sum8(o[t][r] * qn) * TD_ATT_SCALE : -INFINITY;
                    mx = fmaxf(mx, lg[r]);
                }
                mx = td_max_halves(mx);
                if (mx == -INFINITY) mx = 0.f;
                float sm = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    lg[r] = ((vmask >> r) & 1u) ? __expf(lg[r] - mx) : 0.f;
                    sm += lg[r];
                }
                sm = td_sum_halves(sm);
                const float inv = sm > 0.f ? 1.0f / sm : 0.f;
                if ((c & 7) == 0) {
                    float *dst = a.alpha + ((size_t)i * TD_HEADS + 4 * t + (c >> 3)) * TD_K + 4 * hi;
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const float4 e4 = ewq[qd];
                        *reinterpret_cast<float4 *>(dst + 8 * qd) =
                            make_float4(lg[4 * qd] * inv * e4.x, lg[4 * qd + 1] * inv * e4.y, lg[4 * qd + 2] * inv * e4.z,
                                        lg[4 * qd + 3] * inv * e4.w);
                    }
                }
            }
        } else if (MODE == MODE_X2H_V) {
            // out_i = sum_e alpha_e * e_w * v_e ; h_i += out_i   (scatter_sum + residual, :77-83)
#pragma unroll
            for (int t = 0; t < NT2; ++t) {
                const float *al = a.alpha + ((size_t)i * TD_HEADS + 4 * t + (c >> 3)) * TD_K + 4 * hi;
                float out = 0.f;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const float4 av = *reinterpret_cast<const float4 *>(al + 8 * qd);
                    out = fmaf(av.x, o[t][4 * qd + 0], out);
                    out = fmaf(av.y, o[t][4 * qd + 1], out);
                    out = fmaf(av.z, o[t][4 * qd + 2], out);
                    out = fmaf(av.w, o[t][4 * qd + 3], out);
                }
                out = td_sum_halves(out);
                if (hi == 0) a.h[(size_t)i * TD_H + 32 * t + c] += out;
            }
        } else {
            // h2x value pass: lane (head = c < 16, hi) holds xv[row][head];
            // delta_x_i = mean_heads sum_e alpha_e e_w xv_e (x_i - x_j)   (:131-140), masked update (:205-206)
            float sx = 0.f, sy = 0.f, sz = 0.f;
            if (c < TD_HEADS) {
                const float *al = a.alpha + ((size_t)i * TD_HEADS + c) * TD_K + 4 * hi;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const float4 av = *reinterpret_cast<const float4 *>(al + 8 * qd);
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const float wgt = (rr == 0 ? av.x : rr == 1 ? av.y : rr == 2 ? av.z : av.w) * o[0][4 * qd + rr];
                        const float4 rel = *reinterpret_cast<const float4 *>(RELB + 4 * (8 * qd + 4 * hi + rr));
                        sx = fmaf(wgt, rel.x, sx);
                        sy = fmaf(wgt, rel.y, sy);
                        sz = fmaf(wgt, rel.z, sz);
                    }
                }
            }
            sx = td_sum64(sx) * (1.0f / TD_HEADS);
            sy = td_sum64(sy) * (1.0f / TD_HEADS);
            sz = td_sum64(sz) * (1.0f / TD_HEADS);
            if (lane == 0) a.x4_out[i] = make_float4(xi.x + sx, xi.y + sy, xi.z + sz, xi.w);
        }
        TD_STAMP(6);
    }
}

static int edge_grid(int64_t count) {
    int64_t g = (count + 7) / 8;                 // at least one node per wave
    if (g > 256) g = 256;
    if (g >= 8) g = (g / 8) * 8;
    return (int)(g < 1 ? 1 : g);
}

static long long *g_timing_buf = nullptr;
static int g_timing_nodes = 0;
void td_set_edge_timing(long long *buf, int nodes) { g_timing_buf = buf; g_timing_nodes = nodes; }

template <int MODE>
static int launch_edge(EdgeArgs a, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        TD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(edge_mlp_kernel<MODE, false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)EDGE_LDS_BYTES));
        TD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(edge_mlp_kernel<MODE, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)EDGE_LDS_BYTES));
        attr_set = true;
    }
    a.dbg = g_timing_buf; a.dbg_nodes = g_timing_nodes;
    if (g_timing_buf && MODE == MODE_X2H_K)
        edge_mlp_kernel<MODE, true><<<dim3(edge_grid(a.count)), dim3(512), EDGE_LDS_BYTES, s>>>(a);
    else
        edge_mlp_kernel<MODE, false><<<dim3(edge_grid(a.count)), dim3(512), EDGE_LDS_BYTES, s>>>(a);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

int td_launch_edge_pass(int mode, const TdLayer &L, const float4 *x4_in, float4 *x4_out, const int32_t *nbr,
                        const float *ew, const float *P, const float *q, const int32_t *lig_node, int64_t count,
                        float *h, float *alpha, hipStream_t s) {
    if (count == 0) return TD_OK;
    EdgeArgs a;
    a.x4 = x4_in; a.x4_out = x4_out; a.nbr = nbr; a.ew = ew; a.P = P; a.q = q; a.lig_node = lig_node; a.h = h;
    a.alpha = alpha; a.count = count; a.offsets = L.offsets; a.coeff = L.coeff;
    static int stagger = -1;
    if (stagger < 0) { const char *e = getenv("TD_EDGE_STAGGER"); stagger = e ? atoi(e) : 2; }
    a.stagger_sleeps = stagger;
    switch (mode) {
        case MODE_X2H_K: a.mlp = L.hk; return launch_edge<MODE_X2H_K>(a, s);
        case MODE_X2H_V: a.mlp = L.hv; return launch_edge<MODE_X2H_V>(a, s);
        case MODE_H2X_K: a.mlp = L.xk; return launch_edge<MODE_H2X_K>(a, s);
        case MODE_H2X_V: a.mlp = L.xv; return launch_edge<MODE_H2X_V>(a, s);
    }
    td_set_error("td_launch_edge_pass: bad mode %d", mode);
    return TD_EINVAL;
}
