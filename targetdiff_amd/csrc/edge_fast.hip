// Restructured key / value passes of the attention stages (gfx950, fp32 MFMA 32x32x2).
//
// Measured fact that drives this file: v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate and does not overlap
// with VALU work of the co-resident wave (MfmaBusy + VALUBusy ~ 90 % in the plain formulation), so the cost of a
// dst node is ~ 64 cycles x #MFMA + ~ 4 cycles x #VALU.  Both passes therefore avoid ever materialising the
// per-edge key / value vectors (the 32 x 128 x 128 second Linear, 256 MFMAs per node):
//
//   key pass    logits[e, a] = <q_i[a], k_e[a]> = z_e . U_i[:, a] + const(i, a),   U_i[k, a] = sum_{n in head a} q_i[n] W2k[n, k]
//               (models/uni_transformer.py:54,70,73 / :120,133,135).  The constant cancels in the softmax over e.
//               U_i costs 128 x 128 MACs per node (VALU), the logits one 32 x 128 x 16 MFMA product (64 MFMAs).
//               The first layer is evaluated TRANSPOSED (rows = hidden units, columns = edges): LayerNorm
//               statistics become in-lane sums, the neighbour gather 16-byte loads, and z^T is directly the B
//               operand of the logits product -- no LDS transposition.
//   value pass  out_i[n] = sum_e alpha[e, a(n)] (W2v z_e + b)[n] = W2v[n, :] . Zbar_i[a(n), :] + b[n] S_i[a(n)],
//               Zbar_i[a, k] = sum_e alpha[e, a] z_e[k]      (models/uni_transformer.py:56-66,77-83).
//               Zbar is one 16 x 32 x 128 MFMA product (64 MFMAs) whose B operand is z in the C layout it was
//               produced in; the block-diagonal 128 x 128 contraction with W2v runs on the VALU out of LDS.
// Both are exact re-associations of the reference arithmetic (fp32 throughout).
#include "td_device.h"
#include "td_internal.h"

constexpr float TD_ATT_SCALE_F = 0.35355339059327373f;   // 1/sqrt(8)

struct FastArgs {
    const float4 *x4;
    const int32_t *nbr;
    const float *ew;
    const float *P;
    const float *q;
    const int32_t *rows;       // optional list of dst nodes (h2x: ligand atoms; session layer 0: dirty rows)
    const int32_t *count_ptr;  // optional device-side length of `rows` (overrides count)
    float *h;                  // value pass: updated in place
    float *alpha;              // [N][16][32]
    int64_t count;
    TdEdgeMlp mlp;
    const float *offsets;
    float coeff;
    int p_off;                 // column offset of this MLP's i-part inside P (0 key, 256 value)
};

__device__ __forceinline__ float td_max8(float v) {
    v = fmaxf(v, td_dpp<DPP_QUAD_XOR1>(v));
    v = fmaxf(v, td_dpp<DPP_QUAD_XOR2>(v));
    v = fmaxf(v, td_dpp<DPP_ROW_HALF_MIRROR>(v));
    return v;
}
__device__ __forceinline__ float td_max32(float v) {
    v = td_max8(v);
    v = fmaxf(v, td_dpp<DPP_ROW_ROR8>(v));
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}

__device__ __forceinline__ void td_node_range(int64_t count, const int32_t *count_ptr, int64_t &begin, int64_t &end) {
    if (count_ptr) count = *count_ptr;
    // XCD-aware contiguous node ranges: workgroup b runs on XCD b % 8 -> give XCD x the x-th eighth of the nodes
    const int G = gridDim.x;
    int chunk = blockIdx.x;
    if ((G & 7) == 0) chunk = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int64_t per = (count + G - 1) / G;
    begin = (int64_t)chunk * per;
    end = begin + per < count ? begin + per : count;
}

// ================================================================================================ key pass
constexpr int KP_R_FLOATS = 2 * 2 * TD_SLOT_STEPS * 64 * 4;      // 12288
constexpr int KP_WQ_FLOATS = 4 * 16 * 2 * 2 * 16 * 4;            // 16384
constexpr size_t KP_LDS_BYTES = (size_t)(KP_R_FLOATS + KP_WQ_FLOATS + 2 * TD_H) * sizeof(float);

constexpr int KP_WAVES = 12;      // 158 VGPRs -> 3 waves per SIMD; no per-wave LDS, so 12 waves share one copy of the weights
__global__ __launch_bounds__(KP_WAVES * 64) void edge_key_kernel(FastArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const float4 *Rs = reinterpret_cast<const float4 *>(lds);                       // [cls][slot][12][64 lanes] x 4 tiles
    const float4 *Wq = reinterpret_cast<const float4 *>(lds + KP_R_FLOATS);         // [t][r][jq][hi][c < 16] x 4 j
    const float *GAM = lds + KP_R_FLOATS + KP_WQ_FLOATS, *BET = GAM + TD_H;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int c = lane & 31, hi = lane >> 5;
    {
        const float4 *rsrc = reinterpret_cast<const float4 *>(a.mlp.R);
        float4 *rdst = reinterpret_cast<float4 *>(lds);
        for (int idx = tid; idx < KP_R_FLOATS / 4; idx += KP_WAVES * 64) rdst[idx] = rsrc[idx];
        const float4 *wsrc = reinterpret_cast<const float4 *>(a.mlp.Walt);
        float4 *wdst = reinterpret_cast<float4 *>(lds + KP_R_FLOATS);
        for (int idx = tid; idx < KP_WQ_FLOATS / 4; idx += KP_WAVES * 64) wdst[idx] = wsrc[idx];
        if (tid < TD_H) lds[KP_R_FLOATS + KP_WQ_FLOATS + tid] = a.mlp.gamma[tid];
        else if (tid < 2 * TD_H) lds[KP_R_FLOATS + KP_WQ_FLOATS + tid] = a.mlp.beta[tid - TD_H];
    }
    float offk[TD_SLOT_STEPS];
#pragma unroll
    for (int s = 0; s < TD_SLOT_STEPS; ++s) {
        const int k = td_kmap(s, hi);
        offk[s] = k < TD_NG ? a.offsets[k] : 0.f;
    }
    __syncthreads();
    int64_t begin, end;
    td_node_range(a.count, a.count_ptr, begin, end);
    const int cq = c & 15;
    const float headmask = c < TD_HEADS ? 1.f : 0.f;

    for (int64_t it = begin + wid; it < end; it += KP_WAVES) {
        const int64_t i = a.rows ? (int64_t)a.rows[it] : it;
        // ---- geometry: lane (c, hi) owns edge c (both half-waves see the same 32 edges) -----------------------------
        const int j = a.nbr[i * TD_K + c];
        const bool valid = j >= 0;
        const float4 xi = a.x4[i];
        const float4 xj = a.x4[valid ? j : i];
        const float ewc = a.ew[i * TD_K + c];
        const float relx = xi.x - xj.x, rely = xi.y - xj.y, relz = xi.z - xj.z;
        const float d = sqrtf(relx * relx + rely * rely + relz * relz);
        const int slot = xj.w > 0.5f ? 0 : 1;
        const int cls = xi.w > 0.5f ? 0 : 1;
        const bool has_a = __ballot(valid && slot == 0) != 0ull;
        const bool has_b = __ballot(valid && slot == 1) != 0ull;

        // ---- first layer, transposed: acc[t][r] = pre[hidden 32t + erow(r, hi)][edge c] -----------------------------
        floatx16 acc[4];
        {
            const float *pj = a.P + (size_t)(valid ? j : (int)i) * (4 * TD_H) + a.p_off + TD_H + 4 * hi;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const float4 vj = *reinterpret_cast<const float4 *>(pj + 32 * t + 8 * qd);
                    acc[t][4 * qd + 0] = vj.x;
                    acc[t][4 * qd + 1] = vj.y;
                    acc[t][4 * qd + 2] = vj.z;
                    acc[t][4 * qd + 3] = vj.w;
                }
        }
        // The dst-side projection P_i[n] (same for all 32 edges) rides along in the radial MFMA: the table's padding
        // column k = 21 (k-step 9 of the hi half) gets A = P_i[32t + c], B = 1 for the edge's own source-class slot.
        float pit[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) pit[t] = a.P[(size_t)i * (4 * TD_H) + a.p_off + 32 * t + c];
        float gv[TD_SLOT_STEPS];
#pragma unroll
        for (int s = 0; s < TD_SLOT_STEPS; ++s) {
            const int k = td_kmap(s, hi);
            const float u = d - offk[s];
            gv[s] = k < TD_NG ? __expf(a.coeff * u * u) : (k <= TD_NG + 1 ? 1.f : 0.f);
        }
        if (has_a) {
            const bool on = valid && slot == 0;
            const float4 *Rp = Rs + (size_t)((cls * 2 + 0) * TD_SLOT_STEPS) * 64 + lane;
#pragma unroll
            for (int s = 0; s < TD_SLOT_STEPS; ++s) {
                float4 rfrag = Rp[s * 64];                  // A operand: R[kk(s, hi)][32t + c]
                if (s == 9 && hi) rfrag = make_float4(pit[0], pit[1], pit[2], pit[3]);
                const float bv = on ? gv[s] : 0.f;          // B operand: g_kk(d_edge c)
                acc[0] = td_mfma(rfrag.x, bv, acc[0]);
                acc[1] = td_mfma(rfrag.y, bv, acc[1]);
                acc[2] = td_mfma(rfrag.z, bv, acc[2]);
                acc[3] = td_mfma(rfrag.w, bv, acc[3]);
            }
        }
        if (has_b) {
            const bool on = valid && slot == 1;
            const float4 *Rp = Rs + (size_t)((cls * 2 + 1) * TD_SLOT_STEPS) * 64 + lane;
#pragma unroll
            for (int s = 0; s < TD_SLOT_STEPS; ++s) {
                float4 rfrag = Rp[s * 64];
                if (s == 9 && hi) rfrag = make_float4(pit[0], pit[1], pit[2], pit[3]);
                const float bv = on ? gv[s] : 0.f;
                acc[0] = td_mfma(rfrag.x, bv, acc[0]);
                acc[1] = td_mfma(rfrag.y, bv, acc[1]);
                acc[2] = td_mfma(rfrag.z, bv, acc[2]);
                acc[3] = td_mfma(rfrag.w, bv, acc[3]);
            }
        }

        // ---- LayerNorm over the 128 hidden units of edge c: 64 in this lane, 64 in lane c of the other half --------
        float s1 = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) s1 += acc[t][r];
        const float mean = td_sum_halves(s1) * (1.0f / TD_H);
        float s2 = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float dv = acc[t][r] - mean;
                s2 = fmaf(dv, dv, s2);
            }
        const float rstd = __frsqrt_rn(td_sum_halves(s2) * (1.0f / TD_H) + 1e-5f);
        const float nms = -mean * rstd;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const float4 gm = *reinterpret_cast<const float4 *>(GAM + 32 * t + 8 * qd + 4 * hi);
                const float4 bm = *reinterpret_cast<const float4 *>(BET + 32 * t + 8 * qd + 4 * hi);
                acc[t][4 * qd + 0] = fmaxf(fmaf(fmaf(acc[t][4 * qd + 0], rstd, nms), gm.x, bm.x), 0.f);
                acc[t][4 * qd + 1] = fmaxf(fmaf(fmaf(acc[t][4 * qd + 1], rstd, nms), gm.y, bm.y), 0.f);
                acc[t][4 * qd + 2] = fmaxf(fmaf(fmaf(acc[t][4 * qd + 2], rstd, nms), gm.z, bm.z), 0.f);
                acc[t][4 * qd + 3] = fmaxf(fmaf(fmaf(acc[t][4 * qd + 3], rstd, nms), gm.w, bm.w), 0.f);
            }

        // ---- logits^T[head][edge] = sum_n U_i[n][head] z[n][edge]; U_i built on the fly from q_i (A operand) --------
        const float4 q0 = *reinterpret_cast<const float4 *>(a.q + (size_t)i * TD_H + 8 * cq);
        const float4 q1 = *reinterpret_cast<const float4 *>(a.q + (size_t)i * TD_H + 8 * cq + 4);
        floatx16 lg;
#pragma unroll
        for (int r = 0; r < 16; ++r) lg[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float4 w0 = Wq[(((t * 16 + r) * 2 + 0) * 2 + hi) * 16 + cq];
                const float4 w1 = Wq[(((t * 16 + r) * 2 + 1) * 2 + hi) * 16 + cq];
                float u = w0.x * q0.x;
                u = fmaf(w0.y, q0.y, u); u = fmaf(w0.z, q0.z, u); u = fmaf(w0.w, q0.w, u);
                u = fmaf(w1.x, q1.x, u); u = fmaf(w1.y, q1.y, u); u = fmaf(w1.z, q1.z, u); u = fmaf(w1.w, q1.w, u);
                lg = td_mfma(u * headmask, acc[t][r], lg);
            }

        // ---- softmax over the 32 edges (lanes of a half wave) for this half's 8 heads; times the edge gate ----------
        // row r < 8 of the C layout is head erow(r, hi) = (r & 3) + 8 (r >> 2) + 4 hi
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float x = valid ? lg[r] * TD_ATT_SCALE_F : -INFINITY;
            float mx = td_max32(x);
            if (mx == -INFINITY) mx = 0.f;
            const float p = valid ? __expf(x - mx) : 0.f;
            const float sm = td_sum32(p);
            const float al = sm > 0.f ? p * ewc * __frcp_rn(sm) : 0.f;
            a.alpha[((size_t)i * TD_HEADS + td_erow(r, hi)) * TD_K + c] = al;
        }
    }
}

// ================================================================================================ value pass (x2h)
constexpr int VP_R_FLOATS = KP_R_FLOATS;                 // 12288
constexpr int VP_W_FLOATS = 32 * TD_H * 4;               // 16384: W2vK[kq][n][4]
constexpr int VP_ZB_STRIDE = 132;
constexpr int VP_TB_STRIDE = 36;
constexpr int VP_WAVE_FLOATS = 32 * VP_TB_STRIDE;        // 1152 >= 8 * 132: transpose tile, later reused as Zbar half
constexpr size_t VP_LDS_BYTES =
    (size_t)(VP_R_FLOATS + VP_W_FLOATS + 8 * VP_WAVE_FLOATS + 8 * 16 + 3 * TD_H) * sizeof(float);

__global__ __launch_bounds__(512) void edge_value_kernel(FastArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const float4 *Rs = reinterpret_cast<const float4 *>(lds);
    const float4 *Wv = reinterpret_cast<const float4 *>(lds + VP_R_FLOATS);          // [kq 32][n 128] x 4 k
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int c = lane & 31, hi = lane >> 5;
    float *TB = lds + VP_R_FLOATS + VP_W_FLOATS + wid * VP_WAVE_FLOATS;               // wave-private scratch tile
    float *SB = lds + VP_R_FLOATS + VP_W_FLOATS + 8 * VP_WAVE_FLOATS + wid * 16;      // wave-private S[16 heads]
    float *B2 = lds + VP_R_FLOATS + VP_W_FLOATS + 8 * VP_WAVE_FLOATS + 8 * 16;        // b2v[128]
    const float *GAM = B2 + TD_H, *BET = GAM + TD_H;
    {
        const float4 *rsrc = reinterpret_cast<const float4 *>(a.mlp.R);
        float4 *rdst = reinterpret_cast<float4 *>(lds);
        for (int idx = tid; idx < VP_R_FLOATS / 4; idx += 512) rdst[idx] = rsrc[idx];
        const float4 *wsrc = reinterpret_cast<const float4 *>(a.mlp.Walt);
        float4 *wdst = reinterpret_cast<float4 *>(lds + VP_R_FLOATS);
        for (int idx = tid; idx < VP_W_FLOATS / 4; idx += 512) wdst[idx] = wsrc[idx];
        if (tid < TD_H) B2[tid] = a.mlp.b2[tid];
        else if (tid < 2 * TD_H) B2[tid] = a.mlp.gamma[tid - TD_H];
        else if (tid < 3 * TD_H) B2[tid] = a.mlp.beta[tid - 2 * TD_H];
    }
    float offk[TD_SLOT_STEPS];
#pragma unroll
    for (int s = 0; s < TD_SLOT_STEPS; ++s) {
        const int k = td_kmap(s, hi);
        offk[s] = k < TD_NG ? a.offsets[k] : 0.f;
    }
    __syncthreads();
    int64_t begin, end;
    td_node_range(a.count, a.count_ptr, begin, end);

    for (int64_t it = begin + wid; it < end; it += 8) {
        const int64_t i = a.rows ? (int64_t)a.rows[it] : it;
        // ---- geometry: lane (c, hi) owns edge c ------------------------------------------------------------------------
        const int j = a.nbr[i * TD_K + c];
        const bool valid = j >= 0;
        const float4 xi = a.x4[i];
        const float4 xj = a.x4[valid ? j : i];
        const float relx = xi.x - xj.x, rely = xi.y - xj.y, relz = xi.z - xj.z;
        const float d = sqrtf(relx * relx + rely * rely + relz * relz);
        const int slot = xj.w > 0.5f ? 0 : 1;
        const int cls = xi.w > 0.5f ? 0 : 1;
        const bool has_a = __ballot(valid && slot == 0) != 0ull;
        const bool has_b = __ballot(valid && slot == 1) != 0ull;

        // ---- first layer, transposed (as in the key pass): acc[t][r] = pre[hidden 32t + erow(r, hi)][edge c] --------
        floatx16 acc[4];
        {
            const float *pj = a.P + (size_t)(valid ? j : (int)i) * (4 * TD_H) + a.p_off + TD_H + 4 * hi;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const float4 vj = *reinterpret_cast<const float4 *>(pj + 32 * t + 8 * qd);
                    acc[t][4 * qd + 0] = vj.x;
                    acc[t][4 * qd + 1] = vj.y;
                    acc[t][4 * qd + 2] = vj.z;
                    acc[t][4 * qd + 3] = vj.w;
                }
        }
        float pit[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) pit[t] = a.P[(size_t)i * (4 * TD_H) + a.p_off + 32 * t + c];
        float gv[TD_SLOT_STEPS];
#pragma unroll
        for (int s = 0; s < TD_SLOT_STEPS; ++s) {
            const int k = td_kmap(s, hi);
            const float u = d - offk[s];
            gv[s] = k < TD_NG ? __expf(a.coeff * u * u) : (k <= TD_NG + 1 ? 1.f : 0.f);
        }
        if (has_a) {
            const bool on = valid && slot == 0;
            const float4 *Rp = Rs + (size_t)((cls * 2 + 0) * TD_SLOT_STEPS) * 64 + lane;
#pragma unroll
            for (int s = 0; s < TD_SLOT_STEPS; ++s) {
                float4 rfrag = Rp[s * 64];
                if (s == 9 && hi) rfrag = make_float4(pit[0], pit[1], pit[2], pit[3]);   // P_i rides in padding column k = 21
                const float bv = on ? gv[s] : 0.f;
                acc[0] = td_mfma(rfrag.x, bv, acc[0]);
                acc[1] = td_mfma(rfrag.y, bv, acc[1]);
                acc[2] = td_mfma(rfrag.z, bv, acc[2]);
                acc[3] = td_mfma(rfrag.w, bv, acc[3]);
            }
        }
        if (has_b) {
            const bool on = valid && slot == 1;
            const float4 *Rp = Rs + (size_t)((cls * 2 + 1) * TD_SLOT_STEPS) * 64 + lane;
#pragma unroll
            for (int s = 0; s < TD_SLOT_STEPS; ++s) {
                float4 rfrag = Rp[s * 64];
                if (s == 9 && hi) rfrag = make_float4(pit[0], pit[1], pit[2], pit[3]);
                const float bv = on ? gv[s] : 0.f;
                acc[0] = td_mfma(rfrag.x, bv, acc[0]);
                acc[1] = td_mfma(rfrag.y, bv, acc[1]);
                acc[2] = td_mfma(rfrag.z, bv, acc[2]);
                acc[3] = td_mfma(rfrag.w, bv, acc[3]);
            }
        }

        // ---- LayerNorm over the 128 hidden units of edge c: in-lane sums + one exchange between the half waves ------
        float s1 = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) s1 += acc[t][r];
        const float mean = td_sum_halves(s1) * (1.0f / TD_H);
        float s2 = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float dv = acc[t][r] - mean;
                s2 = fmaf(dv, dv, s2);
            }
        const float rstd = __frsqrt_rn(td_sum_halves(s2) * (1.0f / TD_H) + 1e-5f);
        const float nms = -mean * rstd;

        // ---- attention weights of this node: A operand of the aggregation product (lane = head row) ----------------
        float al[16];
        {
            const float *ap = a.alpha + ((size_t)i * TD_HEADS + (c & 15)) * TD_K + 4 * hi;
            const float m = c < TD_HEADS ? 1.f : 0.f;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const float4 v = *reinterpret_cast<const float4 *>(ap + 8 * qd);
                al[4 * qd + 0] = v.x * m; al[4 * qd + 1] = v.y * m; al[4 * qd + 2] = v.z * m; al[4 * qd + 3] = v.w * m;
            }
        }
        float ssum = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) ssum += al[s];
        ssum = td_sum_halves(ssum);                      // S[head c] = sum over all 32 edges
        if (lane < TD_HEADS) SB[lane] = ssum;

        // ---- Zbar[head][k] = sum_e alpha[e][head] z[e][k].  z^T (lane = edge) is normalised tile by tile, flipped
        //      through the wave-private LDS tile into the B-operand layout (lane = hidden unit) and consumed at once.
        floatx16 zb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const float4 gm = *reinterpret_cast<const float4 *>(GAM + 32 * t + 8 * qd + 4 * hi);
                const float4 bm = *reinterpret_cast<const float4 *>(BET + 32 * t + 8 * qd + 4 * hi);
                float4 z;
                z.x = fmaxf(fmaf(fmaf(acc[t][4 * qd + 0], rstd, nms), gm.x, bm.x), 0.f);
                z.y = fmaxf(fmaf(fmaf(acc[t][4 * qd + 1], rstd, nms), gm.y, bm.y), 0.f);
                z.z = fmaxf(fmaf(fmaf(acc[t][4 * qd + 2], rstd, nms), gm.z, bm.z), 0.f);
                z.w = fmaxf(fmaf(fmaf(acc[t][4 * qd + 3], rstd, nms), gm.w, bm.w), 0.f);
                *reinterpret_cast<float4 *>(TB + c * VP_TB_STRIDE + 8 * qd + 4 * hi) = z;      // row = edge c, col = erow(r, hi)
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) zb[t][r] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s)                 // B[e = erow(s, hi)][k = 32t + c]
                zb[t] = td_mfma(al[s], TB[td_erow(s, hi) * VP_TB_STRIDE + c], zb[t]);
        }

        // ---- out[n] = W2v[n, :] . Zbar[head(n), :] + b2v[n] S[head(n)];  h_i += out   (two halves of 64 outputs) -----
        float *ZB = TB;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            // heads 8ph .. 8ph+7 live in rows r = 4ph + rr of half hi: head = rr + 8ph + 4hi
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) ZB[(rr + 4 * hi) * VP_ZB_STRIDE + 32 * t + c] = zb[t][4 * ph + rr];
            const int n = 64 * ph + lane;
            const float *zrow = ZB + (lane >> 3) * VP_ZB_STRIDE;
            float o = B2[n] * SB[8 * ph + (lane >> 3)];
#pragma unroll 8
            for (int kq = 0; kq < 32; ++kq) {
                const float4 w = Wv[kq * TD_H + n];
                const float4 z = *reinterpret_cast<const float4 *>(zrow + 4 * kq);
                o = fmaf(w.x, z.x, o); o = fmaf(w.y, z.y, o); o = fmaf(w.z, z.z, o); o = fmaf(w.w, z.w, o);
            }
            a.h[(size_t)i * TD_H + n] += o;
        }
    }
}

// ================================================================================================ launchers
static int fast_grid(int64_t count) {
    int64_t g = (count + 7) / 8;
    if (g > 256) g = 256;
    if (g >= 8) g = (g / 8) * 8;
    return (int)(g < 1 ? 1 : g);
}

int td_launch_edge_key(const TdEdgeMlp &mlp, const TdLayer &L, const float4 *x4, const int32_t *nbr, const float *ew,
                       const float *P, const float *q, const int32_t *rows, const int32_t *count_ptr, int64_t count,
                       float *alpha, hipStream_t s) {
    if (count == 0) return TD_OK;
    static bool attr_set = false;
    if (!attr_set) {
        TD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(edge_key_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)KP_LDS_BYTES));
        attr_set = true;
    }
    FastArgs a;
    a.x4 = x4; a.nbr = nbr; a.ew = ew; a.P = P; a.q = q; a.rows = rows; a.count_ptr = count_ptr; a.h = nullptr;
    a.alpha = alpha; a.count = count; a.mlp = mlp; a.offsets = L.offsets; a.coeff = L.coeff; a.p_off = 0;
    edge_key_kernel<<<dim3(fast_grid(count)), dim3(KP_WAVES * 64), KP_LDS_BYTES, s>>>(a);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

int td_launch_edge_value(const TdEdgeMlp &mlp, const TdLayer &L, const float4 *x4, const int32_t *nbr, const float *P,
                         const int32_t *rows, const int32_t *count_ptr, int64_t count, float *h, const float *alpha,
                         hipStream_t s) {
    if (count == 0) return TD_OK;
    static bool attr_set = false;
    if (!attr_set) {
        TD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(edge_value_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)VP_LDS_BYTES));
        attr_set = true;
    }
    FastArgs a;
    a.x4 = x4; a.nbr = nbr; a.ew = nullptr; a.P = P; a.q = nullptr; a.rows = rows; a.count_ptr = count_ptr; a.h = h;
    a.alpha = const_cast<float *>(alpha); a.count = count; a.mlp = mlp; a.offsets = L.offsets; a.coeff = L.coeff;
    a.p_off = 2 * TD_H;
    edge_value_kernel<<<dim3(fast_grid(count)), dim3(512), VP_LDS_BYTES, s>>>(a);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}
