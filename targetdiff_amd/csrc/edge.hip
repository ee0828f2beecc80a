// Edge-side kernels of the equivariant attention stack (gfx950, fp32 MFMA 32x32x2):
//   * edge_gate_kernel  -- global edge gate e_w = sigmoid(MLP(20->128->1)(Gaussian(dist)))
//                          (models/uni_transformer.py:312-316, models/common.py:24-26,60-80)
//   * edge_attn_kernel  -- x2h  (BaseX2HAttLayer.forward, models/uni_transformer.py:42-84) and
//                          h2x  (BaseH2XAttLayer.forward, :108-140 + the masked update :205-206)
//
// Structure of edge_attn_kernel.  The kNN graph gives every node exactly 32 in-edges, stored as a dense
// row nbr[i][0..31]; scatter_softmax / scatter_sum over dst (torch_scatter, :73,78,135,139) are therefore
// fixed-length segment reductions over ONE 32-row MFMA tile -- no atomics, no edge lists.
// One workgroup (8 waves) walks a contiguous range of dst nodes.  Per node:
//   waves 0-3 ("k role") evaluate the key MLP, waves 4-7 ("v role") the value MLP.  Wave w of a role owns
//   hidden/output columns [32w, 32w+32): its 128x32 slice of the second Linear lives in 64 VGPRs for the
//   whole kernel (weight-stationary), the first Linear's radial/type table for the current dst class in
//   24 more.
//   (1) first layer:   pre[e][n] = P_i[n] + P_j[j_e][n] + sum_m R[type_e][m][n] g_m(d_e)
//       = node projections (node.hip) gathered per edge + a 32x(24|48)x32 MFMA whose A operand
//       (Gaussians of the edge length, masked by the edge's source class) is built in registers.
//   (2) pre -> LDS (C layout -> row-major), barrier, each lane re-reads HALF A ROW (16 x ds_read_b128),
//       LayerNorm + ReLU in registers (row statistics need one cross-half shuffle),
//   (3) second layer:  64 MFMAs against the stationary W2 slice,
//   (4) k role: logits = <q_i, k_e> per head (8-lane reductions), softmax over the 32 rows, times e_w,
//       -> LDS;  barrier;  v role: out = sum_e alpha_e v_e, residual add, store h (x2h) /
//       delta_x = mean_heads sum_e alpha_e v_e rel_e (h2x).
// Two barriers per node; alpha is double buffered so the v role of node t overlaps the k role of t+1.
#include "td_device.h"
#include "td_internal.h"

constexpr int ZS = 132;                  // LDS row stride of the pre-activation tile (128 + 4)
constexpr float TD_ATT_SCALE = 0.35355339059327373f;   // 1/sqrt(8)   (models/uni_transformer.py:73,135)

// ------------------------------------------------------------------------------------------ edge gate
// One wave per dst node, all 128 hidden units (4 N-tiles).  Pure register kernel: no LDS, no barriers.
__global__ __launch_bounds__(256) void edge_gate_kernel(TdGate g, const float4 *__restrict__ x4,
                                                        const int32_t *__restrict__ nbr, int64_t N,
                                                        float *__restrict__ ew) {
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

    for (int64_t i = wave0; i < N; i += nwaves) {
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

int td_launch_gate(const TdGate &g, const float4 *x4, const int32_t *nbr, int64_t N, float *ew, hipStream_t s) {
    if (N == 0) return TD_OK;
    int64_t blocks = (N + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    edge_gate_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(g, x4, nbr, N, ew);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

// ------------------------------------------------------------------------------------------ x2h / h2x
struct EdgeArgs {
    const float4 *x4;        // [N] (x, y, z, is_ligand)
    float4 *x4_out;          // h2x: updated coordinates (ligand rows only are written)
    const int32_t *nbr;      // [N][32]
    const float *ew;         // [N][32] global edge gate
    const float *P;          // [N][512] node projections of this stage
    const float *q;          // [N][128] query vectors of this stage
    const int32_t *lig_node; // h2x: list of dst nodes
    float *h;                // x2h: updated in place
    int64_t count;           // number of dst nodes to process
    TdEdgeMlp mk, mv;
    const float *offsets;
    float coeff;
    long long *dbg;          // TIMING variant: [8 waves][dbg_segs][8 stamps] cycle counters of workgroup 0
    int dbg_segs;
};

// Raw geometry loads of one dst node, issued one pipeline stage ahead of their use.
struct GeoRaw {
    int64_t i;
    int j;
    float4 xi, xj;
    int4 jq[4];      // neighbour ids of this lane's 16 C-layout rows: rows 8q+4hi .. 8q+4hi+3
};

#define TD_STAMP(k)                                                                                   \
    do {                                                                                              \
        if (TIMING && blockIdx.x == 0 && lane == 0 && seg < a.dbg_segs)                               \
            a.dbg[((size_t)wid * a.dbg_segs + seg) * 8 + (k)] = clock64();                            \
    } while (0)

template <bool H2X, bool TIMING>
__global__ __launch_bounds__(512) void edge_attn_kernel(EdgeArgs a) {
    __shared__ __attribute__((aligned(16))) float Z[2][32][ZS];         // [role][edge][hidden]
    __shared__ __attribute__((aligned(16))) float ALPHA[2][TD_HEADS][32];  // [node parity][head][edge]  alpha * e_w
    __shared__ __attribute__((aligned(16))) float GB[2][2][TD_H];       // [role][gamma|beta][hidden]
    __shared__ __attribute__((aligned(16))) float XVP[4][32][16];       // h2x: per-wave partial xv

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int role = wid >> 2, w = wid & 3;
    const int c = lane & 31, hi = lane >> 5;
    const int n = 32 * w + c;
    const TdEdgeMlp mlp = role ? a.mv : a.mk;
    const bool full2 = !H2X || role == 0;       // second Linear with 128 outputs (N-split) vs xv's 16 (K-split)

    // ---- prologue: stationary weights ---------------------------------------------------------------
    float w2[TD_KSTEPS];
    if (full2) {
#pragma unroll
        for (int s = 0; s < TD_KSTEPS; ++s) w2[s] = mlp.W2[(size_t)(w * TD_KSTEPS + s) * 64 + lane];
    } else {
#pragma unroll
        for (int s = 0; s < 16; ++s) w2[s] = mlp.W2[(size_t)(w * 16 + s) * 64 + lane];
    }
    const float b2n = full2 ? mlp.b2[n] : 0.f;
    {
        const int t = tid & 255;
        if (t < TD_H) GB[role][0][t] = mlp.gamma[t];
        else GB[role][1][t - TD_H] = mlp.beta[t - TD_H];
    }
    float offk[TD_SLOT_STEPS];
#pragma unroll
    for (int s = 0; s < TD_SLOT_STEPS; ++s) {
        const int k = td_kmap(s, hi);
        offk[s] = k < TD_NG ? a.offsets[k] : 0.f;
    }
    float rf[2][TD_SLOT_STEPS];
    int cur_cls = -1;

    // ---- XCD-aware contiguous node ranges: workgroup b runs on XCD b % 8 -> give XCD x the x-th eighth -----
    const int G = gridDim.x;
    int chunk = blockIdx.x;
    if ((G & 7) == 0) chunk = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int64_t per = (a.count + G - 1) / G;
    const int64_t begin = (int64_t)chunk * per;
    const int64_t end = begin + per < a.count ? begin + per : a.count;
    const int cnt = end > begin ? (int)(end - begin) : 0;

    auto load_geo = [&](int t) {
        GeoRaw g;
        g.i = H2X ? (int64_t)a.lig_node[begin + t] : begin + t;
        g.j = a.nbr[g.i * TD_K + c];
        g.xi = a.x4[g.i];
        g.xj = a.x4[g.j >= 0 ? g.j : g.i];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
            g.jq[qd] = *reinterpret_cast<const int4 *>(a.nbr + g.i * TD_K + 8 * qd + 4 * hi);
        return g;
    };

    // state of the node a wave is working on (first layer in one segment, second layer in the next)
    GeoRaw nxt;
    if (cnt > 0) nxt = load_geo(0);
    int64_t ci = 0;            // node id
    unsigned cvalid = 0;       // bit r: C-layout row r of this lane is a real edge
    float crx = 0.f, cry = 0.f, crz = 0.f;     // x_i - x_j of edge c   (h2x)
    float4 cxi = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    // ---- software pipeline over segments --------------------------------------------------------------------
    // The two roles run half a node out of phase so that, on every SIMD, one wave's 64-MFMA second layer
    // overlaps the other wave's gather / first layer / softmax:
    //   segment 2t   : k role first(t)            | v role second(t-1) + reduce with alpha(t-1)
    //   segment 2t+1 : k role second(t) -> alpha  | v role first(t)
    // One barrier per segment.  Z[role] is written in the role's "first" segment and read in its "second".
    for (int seg = 0; seg <= 2 * cnt; ++seg) {
        const bool first_phase = ((seg & 1) == role);
        const int t = first_phase ? (seg - role) >> 1 : (seg - 1 - role) >> 1;   // node handled in this segment
        TD_STAMP(0);
        if (first_phase && t < cnt) {
            // ================= first layer of node t =======================================================
            const GeoRaw g = nxt;
            ci = g.i; cxi = g.xi;
            const bool valid = g.j >= 0;
            crx = g.xi.x - g.xj.x; cry = g.xi.y - g.xj.y; crz = g.xi.z - g.xj.z;        // x[dst] - x[src]
            const float d = sqrtf(crx * crx + cry * cry + crz * crz);
            const int slot = g.xj.w > 0.5f ? 0 : 1;         // source class: 0 ligand, 1 protein
            const int cls = g.xi.w > 0.5f ? 0 : 1;          // destination class (wave uniform)
            const bool has_a = __ballot(valid && slot == 0) != 0ull;
            const bool has_b = __ballot(valid && slot == 1) != 0ull;
            if (cls != cur_cls) {
                cur_cls = cls;
#pragma unroll
                for (int sl = 0; sl < 2; ++sl)
#pragma unroll
                    for (int s = 0; s < TD_SLOT_STEPS; ++s)
                        rf[sl][s] = mlp.R[(size_t)((((cls * 4 + w) * 2 + sl) * TD_SLOT_STEPS) + s) * 64 + lane];
            }
            float bj[16];
            const float *Pj = a.P + role * 256 + TD_H + n;
            cvalid = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int4 jv = g.jq[r >> 2];
                const int jr = (r & 3) == 0 ? jv.x : (r & 3) == 1 ? jv.y : (r & 3) == 2 ? jv.z : jv.w;
                cvalid |= jr >= 0 ? (1u << r) : 0u;
                bj[r] = Pj[(size_t)(jr >= 0 ? jr : (int)g.i) * (4 * TD_H)];
            }
            const float pi = a.P[(size_t)g.i * (4 * TD_H) + role * 256 + n];
            TD_STAMP(1);
            floatx16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = pi;
            float gv[TD_SLOT_STEPS];
#pragma unroll
            for (int s = 0; s < TD_SLOT_STEPS; ++s) {
                const int k = td_kmap(s, hi);
                const float u = d - offk[s];
                gv[s] = k < TD_NG ? expf(a.coeff * u * u) : (k == TD_NG ? 1.f : 0.f);
            }
            if (has_a) {
                const bool on = valid && slot == 0;
#pragma unroll
                for (int s = 0; s < TD_SLOT_STEPS; ++s) acc = td_mfma(on ? gv[s] : 0.f, rf[0][s], acc);
            }
            if (has_b) {
                const bool on = valid && slot == 1;
#pragma unroll
                for (int s = 0; s < TD_SLOT_STEPS; ++s) acc = td_mfma(on ? gv[s] : 0.f, rf[1][s], acc);
            }
            TD_STAMP(2);
#pragma unroll
            for (int r = 0; r < 16; ++r) Z[role][td_erow(r, hi)][n] = acc[r] + bj[r];
            TD_STAMP(3);
        } else if (!first_phase && t >= 0 && t < cnt) {
            // ================= second layer of node t ======================================================
            if (t + 1 < cnt) nxt = load_geo(t + 1);       // prefetch: lands while the MFMAs below run
            float hres = 0.f;
            if (!H2X && role == 1 && hi == 0) hres = a.h[(size_t)ci * TD_H + n];   // residual, needed after the MFMAs
            const int buf = t & 1;
            // LayerNorm + ReLU in A layout: lane (edge c, half hi) owns k in {8m+4hi .. 8m+4hi+3}.
            // Two LDS passes: (1) shifted sums -> mean / variance, (2) normalise each 16-byte chunk right
            // before the 4 MFMAs that consume it.
            const float *zrow = &Z[role][c][4 * hi];
            const float *gam = &GB[role][0][4 * hi];
            const float *bet = &GB[role][1][4 * hi];
            const float shift = Z[role][c][0];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                float4 v = *reinterpret_cast<const float4 *>(zrow + 8 * m);
                v.x -= shift; v.y -= shift; v.z -= shift; v.w -= shift;
                s1 += (v.x + v.y) + (v.z + v.w);
                s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            }
            s1 = td_sum_halves(s1);
            s2 = td_sum_halves(s2);
            const float dmean = s1 * (1.0f / TD_H);
            const float mean = shift + dmean;
            const float var = fmaxf(s2 * (1.0f / TD_H) - dmean * dmean, 0.f);
            const float rstd = 1.0f / sqrtf(var + 1e-5f);
            TD_STAMP(1);
            floatx16 acc2;
            const float nms = -mean * rstd;                 // z = relu((v*rstd + nms) * gamma + beta)
            auto norm4 = [&](const float4 &v, const float4 &gm, const float4 &bm) {
                return make_float4(fmaxf(fmaf(fmaf(v.x, rstd, nms), gm.x, bm.x), 0.f),
                                   fmaxf(fmaf(fmaf(v.y, rstd, nms), gm.y, bm.y), 0.f),
                                   fmaxf(fmaf(fmaf(v.z, rstd, nms), gm.z, bm.z), 0.f),
                                   fmaxf(fmaf(fmaf(v.w, rstd, nms), gm.w, bm.w), 0.f));
            };
            if (full2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[r] = b2n;
                // Software pipeline: while the 4 dependent MFMAs of chunk m occupy the matrix pipe (4 x 64
                // cycles), the VALU normalises chunk m+1 and the LDS returns chunk m+2.
                float4 zc = norm4(*reinterpret_cast<const float4 *>(zrow), *reinterpret_cast<const float4 *>(gam),
                                  *reinterpret_cast<const float4 *>(bet));
                float4 vn = *reinterpret_cast<const float4 *>(zrow + 8);
                float4 gn = *reinterpret_cast<const float4 *>(gam + 8);
                float4 bn = *reinterpret_cast<const float4 *>(bet + 8);
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    float4 zn = zc;
                    if (m + 1 < 16) zn = norm4(vn, gn, bn);
                    if (m + 2 < 16) {
                        vn = *reinterpret_cast<const float4 *>(zrow + 8 * (m + 2));
                        gn = *reinterpret_cast<const float4 *>(gam + 8 * (m + 2));
                        bn = *reinterpret_cast<const float4 *>(bet + 8 * (m + 2));
                    }
                    acc2 = td_mfma(zc.x, w2[4 * m + 0], acc2);
                    acc2 = td_mfma(zc.y, w2[4 * m + 1], acc2);
                    acc2 = td_mfma(zc.z, w2[4 * m + 2], acc2);
                    acc2 = td_mfma(zc.w, w2[4 * m + 3], acc2);
                    zc = zn;
                    // issue order hint: MFMA | 3 VALU + 1 LDS read | MFMA | ...   (masks: 0x8 MFMA, 0x2 VALU, 0x100 DS read)
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                }
            } else {
                // h2x value MLP: second Linear is 128 -> 16; wave w contracts hidden units [32w, 32w+32) (K split).
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
                for (int mm = 0; mm < 4; ++mm) {
                    const int m = 4 * w + mm;
                    const float4 v = *reinterpret_cast<const float4 *>(zrow + 8 * m);
                    const float4 gm = *reinterpret_cast<const float4 *>(gam + 8 * m);
                    const float4 bm = *reinterpret_cast<const float4 *>(bet + 8 * m);
                    const float4 z = norm4(v, gm, bm);
                    acc2 = td_mfma(z.x, w2[4 * mm + 0], acc2);
                    acc2 = td_mfma(z.y, w2[4 * mm + 1], acc2);
                    acc2 = td_mfma(z.z, w2[4 * mm + 2], acc2);
                    acc2 = td_mfma(z.w, w2[4 * mm + 3], acc2);
                }
            }

            TD_STAMP(2);
            if (role == 0) {
                // ---- attention logits + segment softmax over the 32 in-edges (scatter_softmax) -----------
                const float qn = a.q[(size_t)ci * TD_H + n];
                float4 ewq[4];
#pragma unroll
                for (int qd = 0; qd < 4; ++qd)
                    ewq[qd] = *reinterpret_cast<const float4 *>(a.ew + ci * TD_K + 8 * qd + 4 * hi);
                float lg[16];
                float mx = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    lg[r] = ((cvalid >> r) & 1u) ? td_sum8(acc2[r] * qn) * TD_ATT_SCALE : -INFINITY;
                    mx = fmaxf(mx, lg[r]);
                }
                mx = td_max_halves(mx);
                if (mx == -INFINITY) mx = 0.f;
                float sm = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    lg[r] = ((cvalid >> r) & 1u) ? __expf(lg[r] - mx) : 0.f;
                    sm += lg[r];
                }
                sm = td_sum_halves(sm);
                const float inv = sm > 0.f ? 1.0f / sm : 0.f;
                const int head = 4 * w + (c >> 3);
                if ((c & 7) == 0) {
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const float4 e4 = ewq[qd];
                        *reinterpret_cast<float4 *>(&ALPHA[buf][head][8 * qd + 4 * hi]) =
                            make_float4(lg[4 * qd] * inv * e4.x, lg[4 * qd + 1] * inv * e4.y, lg[4 * qd + 2] * inv * e4.z,
                                        lg[4 * qd + 3] * inv * e4.w);
                    }
                }
            } else if (!H2X) {
                // ---- out_i = sum_e alpha_e * e_w * v_e ; h_i += out_i (scatter_sum + residual, :77-83) -----
                // alpha(t) was published by the k role one segment ago.
                const float *al = &ALPHA[buf][4 * w + (c >> 3)][4 * hi];
                float out = 0.f;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const float4 av = *reinterpret_cast<const float4 *>(al + 8 * qd);
                    out += av.x * acc2[4 * qd + 0];
                    out += av.y * acc2[4 * qd + 1];
                    out += av.z * acc2[4 * qd + 2];
                    out += av.w * acc2[4 * qd + 3];
                }
                out = td_sum_halves(out);
                if (hi == 0) a.h[(size_t)ci * TD_H + n] = hres + out;
            } else {
                if (c < 16) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) XVP[w][td_erow(r, hi)][c] = acc2[r];
                }
            }
        }
        TD_STAMP(4);
        __syncthreads();
        TD_STAMP(5);
        if (H2X && role == 1 && w == 0 && !first_phase && t >= 0 && t < cnt) {
            // partial xv of node t are complete after the barrier; alpha(t) is in ALPHA[t & 1] (the k role
            // writes the other parity during the next segment).  This wave still holds node t's geometry.
            const int buf = t & 1;
            float sacc = 0.f;
#pragma unroll
            for (int hh = 0; hh < 8; ++hh) {
                const int hd = 8 * hi + hh;
                const float xv = ((XVP[0][c][hd] + XVP[1][c][hd]) + (XVP[2][c][hd] + XVP[3][c][hd])) + a.mv.b2[hd];
                sacc += ALPHA[buf][hd][c] * xv;
            }
            const float dxs = td_sum64(sacc * crx) * (1.0f / TD_HEADS);
            const float dys = td_sum64(sacc * cry) * (1.0f / TD_HEADS);
            const float dzs = td_sum64(sacc * crz) * (1.0f / TD_HEADS);
            if (lane == 0) a.x4_out[ci] = make_float4(cxi.x + dxs, cxi.y + dys, cxi.z + dzs, cxi.w);
        }
    }
}

static long long *g_timing_buf = nullptr;
static int g_timing_segs = 0;
void td_set_edge_timing(long long *buf, int segs) { g_timing_buf = buf; g_timing_segs = segs; }

static int edge_grid(int64_t count) {
    int64_t g = count < 256 ? count : 256;
    if (g >= 8) g = (g / 8) * 8;
    return (int)(g < 1 ? 1 : g);
}

int td_launch_x2h(const TdLayer &L, const float4 *x4, const int32_t *nbr, const float *ew, const float *P,
                  const float *q, int64_t N, float *h, hipStream_t s) {
    if (N == 0) return TD_OK;
    EdgeArgs a;
    a.x4 = x4; a.x4_out = nullptr; a.nbr = nbr; a.ew = ew; a.P = P; a.q = q; a.lig_node = nullptr; a.h = h;
    a.count = N; a.mk = L.hk; a.mv = L.hv; a.offsets = L.offsets; a.coeff = L.coeff;
    a.dbg = g_timing_buf; a.dbg_segs = g_timing_segs;
    if (g_timing_buf) edge_attn_kernel<false, true><<<dim3(edge_grid(N)), dim3(512), 0, s>>>(a);
    else edge_attn_kernel<false, false><<<dim3(edge_grid(N)), dim3(512), 0, s>>>(a);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

int td_launch_h2x(const TdLayer &L, const float4 *x4_in, float4 *x4_out, const int32_t *nbr, const float *ew,
                  const float *P, const float *q, const int32_t *lig_node, int64_t Nl, hipStream_t s) {
    if (Nl == 0) return TD_OK;
    EdgeArgs a;
    a.x4 = x4_in; a.x4_out = x4_out; a.nbr = nbr; a.ew = ew; a.P = P; a.q = q; a.lig_node = lig_node; a.h = nullptr;
    a.count = Nl; a.mk = L.xk; a.mv = L.xv; a.offsets = L.offsets; a.coeff = L.coeff;
    a.dbg = nullptr; a.dbg_segs = 0;
    edge_attn_kernel<true, false><<<dim3(edge_grid(Nl)), dim3(512), 0, s>>>(a);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}
