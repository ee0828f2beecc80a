// Global edge gate e_w = sigmoid(MLP(20 -> 128 -> 1)(GaussianSmearing(dist)))   (gfx950, fp32 MFMA 32x32x2)
//   models/uni_transformer.py:312-316 (edge_pred_layer on the step-start distances, reused by all layers),
//   models/common.py:24-26 (Gaussians), :60-80 (Linear -> LayerNorm -> ReLU -> Linear).
// One wave per 32-slot row of the neighbour table (a dst node on the default k = 32 graph, a chunk of a dst node's
// in-edges on general graphs: chunk_node != nullptr), all 128 hidden units as 4 N tiles.  Pure register kernel.
// This is the fp32 variant (model option edge_key_split = 0); the default is edge_gate16_kernel in edge16.hip.
#include "td_device.h"
#include "td_internal.h"

// ------------------------------------------------------------------------------------------ edge gate
// One wave per dst node, all 128 hidden units (4 N-tiles).  Pure register kernel: no LDS, no barriers.
__global__ __launch_bounds__(256, 2) void edge_gate_kernel(TdGate g, const float4 *__restrict__ x4,
                                                        const int32_t *__restrict__ nbr, int64_t N,
                                                        const int32_t *__restrict__ rows,
                                                        const int32_t *__restrict__ count_ptr,
                                                        const int32_t *__restrict__ chunk_node, float *__restrict__ ew) {
    if (count_ptr) N = *count_ptr;
    const int lane = threadIdx.x & 63;
    const int c = lane & 31, hi = lane >> 5;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;

    float4 R[TD_SLOT_STEPS];
    const float4 *Rp = reinterpret_cast<const float4 *>(g.R);
#pragma unroll
    for (int s = 0; s < TD_SLOT_STEPS; ++s) R[s] = Rp[s * 64 + lane];
    float b0[4], bet[4], w3[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        b0[t] = g.b0[32 * t + c]; bet[t] = g.beta[32 * t + c]; w3[t] = g.w3[32 * t + c];
    }
    float offk[TD_SLOT_STEPS];
#pragma unroll
    for (int s = 0; s < TD_SLOT_STEPS; ++s) {
        const int k = td_kmap(s, hi);
        offk[s] = k < TD_NG ? g.offsets[k] : 0.f;
    }

    for (int64_t it = wave0; it < N; it += nwaves) {
        const int64_t row_id = rows ? (int64_t)rows[it] : it;            // row of nbr / ew
        const int64_t i = chunk_node ? (int64_t)chunk_node[row_id] : row_id;   // its dst node
        const int j = nbr[row_id * TD_K + c];
        const bool valid = j >= 0;
        const float4 xi = x4[i];
        const float4 xj = x4[valid ? j : i];
        const float dx = xi.x - xj.x, dy = xi.y - xj.y, dz = xi.z - xj.z;
        const float d = td_sqrt_d2(dx * dx + dy * dy + dz * dz);      // the one distance function of every kernel that feeds Gaussians
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
        // LayerNorm in the folded form the gate's weights are packed for (FoldedMlp, pack.cpp): the accumulators hold the centred
        // pre-activation times the sign of the LayerNorm weight, bet = beta / (|gamma| M), w3 carries |gamma| M, the ReLU is the FMA's output clamp
        float outv = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d0 = acc[0][r], d1 = acc[1][r], d2 = acc[2][r], d3 = acc[3][r];
            const float sc = __frsqrt_rn(fmaf(td_sum32((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)), g.ln_c1, g.ln_c2));      // 1 / (sigma M)
            float part = td_clamp01(fmaf(d0, sc, bet[0])) * w3[0];
            part = fmaf(td_clamp01(fmaf(d1, sc, bet[1])), w3[1], part);
            part = fmaf(td_clamp01(fmaf(d2, sc, bet[2])), w3[2], part);
            part = fmaf(td_clamp01(fmaf(d3, sc, bet[3])), w3[3], part);
            const float logit = td_sum32(part) + g.b3;
            if (c == r) outv = 1.0f / (1.0f + expf(-logit));
        }
        // lane (c < 16, hi) holds the gate of edge row erow(c, hi)
        const int row = td_erow(c & 15, hi);
        const int jrow = __shfl(j, row);
        if (c < 16) ew[row_id * TD_K + row] = jrow >= 0 ? outv : 0.f;
    }
}

int td_launch_gate(const TdGate &g, const float4 *x4, const int32_t *nbr, int64_t N, const int32_t *rows,
                   const int32_t *count_ptr, float *ew, hipStream_t s, const int32_t *chunk_node) {
    if (N == 0) return TD_OK;
    if (g.use_split) return td_launch_gate16(g, x4, nbr, N, rows, count_ptr, ew, s, chunk_node);
    int64_t blocks = (N + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    edge_gate_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(g, x4, nbr, N, rows, count_ptr, chunk_node, ew);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}


// ------------------------------------------------------------------------------------------ per-layer gate (ew_net_type 'r')
// e_w = sigmoid(Linear(4 x 20 -> 1)(outer_product(edge type, GaussianSmearing(dist))))  (models/uni_transformer.py:34-35, 60-61, 102-103,
// 124-125): with a one-hot edge type the 80-wide dot product is the 20 Gaussians against the weights of the edge's type.  One thread per
// slot of the 32-slot neighbour table; w = [4 types][20] + bias.  ew_net_type 'none' (e_w = 1) runs the same kernel with zero weights and a
// bias of 40 (sigmoid = 1.0f exactly).  A non-default configuration: nothing here is tuned.
__global__ void layer_gate_kernel(const float *__restrict__ w, const float *__restrict__ offsets, float coeff,
                                  const float4 *__restrict__ x4, const int32_t *__restrict__ nbr, int64_t N, float *__restrict__ ew) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * TD_K) return;
    const int64_t i = t / TD_K;
    const int j = nbr[t];
    if (j < 0) { ew[t] = 0.f; return; }
    const float4 xi = x4[i], xj = x4[j];
    const float dx = xi.x - xj.x, dy = xi.y - xj.y, dz = xi.z - xj.z;
    const float d = td_sqrt_d2(dx * dx + dy * dy + dz * dz);      // the one distance function of every kernel that feeds Gaussians
    // edge type (models/uni_transformer.py:292-297): 0 l<-l, 1 src lig/dst prot, 2 src prot/dst lig, 3 p<-p
    const bool dl = xi.w > 0.5f, sl = xj.w > 0.5f;
    const int type = dl ? (sl ? 0 : 2) : (sl ? 1 : 3);
    float logit = w[4 * TD_NG];
#pragma unroll
    for (int k = 0; k < TD_NG; ++k) {
        const float u = d - offsets[k];
        logit = fmaf(w[type * TD_NG + k], expf(coeff * u * u), logit);
    }
    ew[t] = 1.0f / (1.0f + expf(-logit));
}

int td_launch_layer_gate(const float *w, const float *offsets, float coeff, const float4 *x4, const int32_t *nbr, int64_t N, float *ew,
                         hipStream_t s) {
    if (N == 0) return TD_OK;
    const int64_t total = N * TD_K;
    layer_gate_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(w, offsets, coeff, x4, nbr, N, ew);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}
