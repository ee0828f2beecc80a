// Node-side GEMMs of one attention stage (gfx950, fp32 MFMA 32x32x2).
//
// The reference feeds every edge MLP the 340-wide row [type | r_feat | h_i | h_j]
// (models/uni_transformer.py:45-51,111-118).  Linear(340,128) is linear in h_i and h_j, so their
// contributions are computed ONCE PER NODE here instead of once per edge:
//     P[i] = [ Wk[:,84:212] h_i + bk | Wk[:,212:340] h_i | Wv[:,84:212] h_i + bv | Wv[:,212:340] h_i ]   (4 x 128)
// and the per-node query MLP q_i = MLP_q(h_i) (models/uni_transformer.py:70,133) is evaluated in the same
// pass.  Exact up to fp32 re-association.
//
// Work decomposition: workgroup = 4 waves, wave w owns 32 consecutive nodes (M = 32 rows of the MFMA
// tile) and all 128 output columns (4 N-tiles -> 64 accumulator VGPRs).  A operands (the 32 x 128 h tile)
// stay in registers for all six GEMMs; B operands stream from L2 in pre-packed fragment order
// (one coalesced 1 KiB dwordx4 load feeds 4 MFMAs).
#include "td_device.h"
#include "td_internal.h"

constexpr int ZSTRIDE = 132;     // 128 + 4: conflict-free ds_read_b128 of rows (stride/4 odd)

// 128-deep GEMM of one 32-row A tile against all 4 N tiles.  B fragments stream from L2 (~600-900 cycles); a
// ring of PF 16-byte loads is kept in flight explicitly (hipcc on its own keeps 1-2, which left the kernel
// latency-bound: 4 MFMAs = 256 cycles per k-step).
constexpr int PF = 6;
__device__ __forceinline__ void gemm128(const float4 (&a)[16], const float4 *__restrict__ B, int lane,
                                        floatx16 (&acc)[4]) {
    float4 ring[PF];
#pragma unroll
    for (int s = 0; s < PF; ++s) ring[s] = B[s * 64 + lane];
#pragma unroll
    for (int s = 0; s < TD_KSTEPS; ++s) {
        const float4 b = ring[s % PF];
        if (s + PF < TD_KSTEPS) ring[s % PF] = B[(s + PF) * 64 + lane];
        const float4 am = a[s >> 2];
        const float av = (s & 3) == 0 ? am.x : (s & 3) == 1 ? am.y : (s & 3) == 2 ? am.z : am.w;
        acc[0] = td_mfma(av, b.x, acc[0]);
        acc[1] = td_mfma(av, b.y, acc[1]);
        acc[2] = td_mfma(av, b.z, acc[2]);
        acc[3] = td_mfma(av, b.w, acc[3]);
    }
}

// mat_mask: bit m (0..3) -> projection m of [k_i, k_j, v_i, v_j]; bit 4 -> query MLP.  rows != nullptr: process only
// the listed node ids (h2x needs the dst-side projections and queries of ligand atoms only).
__global__ __launch_bounds__(256, 2) void node_proj_kernel(TdNodeStage st, const float *__restrict__ h, int64_t N,
                                                        const int32_t *__restrict__ rows, unsigned mat_mask,
                                                        float *__restrict__ P, float *__restrict__ q) {
    extern __shared__ __attribute__((aligned(16))) float zbuf[];     // [4 waves][32 rows][ZSTRIDE]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, hi = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 32;
    const int64_t aslot = row0 + c;
    const int64_t arow = aslot < N ? (rows ? (int64_t)rows[aslot] : aslot) : -1;

    float4 a[16];
#pragma unroll
    for (int m = 0; m < 16; ++m)
        a[m] = (arow >= 0) ? *reinterpret_cast<const float4 *>(h + arow * TD_H + 8 * m + 4 * hi)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
    // output row of each C-layout register row
    int64_t orow[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t slot = row0 + td_erow(r, hi);
        orow[r] = slot < N ? (rows ? (int64_t)rows[slot] : slot) : -1;
    }

    const float4 *Bp = reinterpret_cast<const float4 *>(st.projB);
    floatx16 acc[4];
    for (int mat = 0; mat < 4; ++mat) {
        if (!((mat_mask >> mat) & 1u)) continue;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float bias = st.projBias[mat * TD_H + 32 * t + c];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = bias;
        }
        gemm128(a, Bp + (size_t)mat * TD_KSTEPS * 64, lane, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (orow[r] >= 0) {
#pragma unroll
                for (int t = 0; t < 4; ++t) P[orow[r] * (4 * TD_H) + mat * TD_H + 32 * t + c] = acc[t][r];
            }
        }
    }
    if (!((mat_mask >> 4) & 1u)) return;      // uniform over the workgroup: no barrier is skipped by a subset

    // ---- query MLP: Linear -> LayerNorm -> ReLU -> Linear (models/common.py:60-80) -------------------
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float bias = st.projBias[4 * TD_H + 32 * t + c];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = bias;
    }
    gemm128(a, Bp + (size_t)4 * TD_KSTEPS * 64, lane, acc);
    float gam[4], bet[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        gam[t] = st.qGamma[32 * t + c];
        bet[t] = st.qBeta[32 * t + c];
    }
    float *zw = zbuf + wave * 32 * ZSTRIDE;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float s1 = (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]);
        const float mean = td_sum32(s1) * (1.0f / TD_H);
        float d0 = acc[0][r] - mean, d1 = acc[1][r] - mean, d2 = acc[2][r] - mean, d3 = acc[3][r] - mean;
        float s2 = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        const float var = td_sum32(s2) * (1.0f / TD_H);
        const float rstd = __frsqrt_rn(var + 1e-5f);
        const int row = td_erow(r, hi);
        zw[row * ZSTRIDE + c] = fmaxf(d0 * rstd * gam[0] + bet[0], 0.f);
        zw[row * ZSTRIDE + 32 + c] = fmaxf(d1 * rstd * gam[1] + bet[1], 0.f);
        zw[row * ZSTRIDE + 64 + c] = fmaxf(d2 * rstd * gam[2] + bet[2], 0.f);
        zw[row * ZSTRIDE + 96 + c] = fmaxf(d3 * rstd * gam[3] + bet[3], 0.f);
    }
    __syncthreads();
    float4 a2[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) a2[m] = *reinterpret_cast<const float4 *>(zw + c * ZSTRIDE + 8 * m + 4 * hi);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float bias = st.q3Bias[32 * t + c];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = bias;
    }
    gemm128(a2, reinterpret_cast<const float4 *>(st.q3B), lane, acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (orow[r] >= 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t) q[orow[r] * TD_H + 32 * t + c] = acc[t][r];
        }
    }
}

int td_launch_node_proj(const TdNodeStage &st, const float *h, int64_t N, const int32_t *rows, unsigned mat_mask,
                        float *P, float *q, hipStream_t s) {
    if (N == 0 || mat_mask == 0) return TD_OK;
    static bool attr_set = false;
    const size_t lds = 4 * 32 * ZSTRIDE * sizeof(float);
    if (!attr_set) {
        TD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(node_proj_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    node_proj_kernel<<<dim3((unsigned)((N + 127) / 128)), dim3(256), lds, s>>>(st, h, N, rows, mat_mask, P, q);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}
