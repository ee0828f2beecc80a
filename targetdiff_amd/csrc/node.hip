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
// stay in registers for all six GEMMs; B operands (pre-packed fragment order) are staged through LDS in 16 KiB
// chunks shared by the workgroup's 4 waves (gemm128_lds).

#include <type_traits>
#include "td_device.h"
#include "td_internal.h"

constexpr int NP_CHUNK_STEPS = 16;                       // k-steps per staged B chunk
constexpr int NP_CHUNK_F4 = NP_CHUNK_STEPS * 64;         // float4 per chunk (16 KiB)
constexpr int NP_CHUNKS = TD_KSTEPS / NP_CHUNK_STEPS;    // 4 chunks per 128-deep GEMM
constexpr int NP_TSTRIDE = 36;                           // transpose tile [32 rows][32 + 4]
constexpr size_t NP_LDS_BYTES = (size_t)(2 * NP_CHUNK_F4 * 4 + 4 * 32 * NP_TSTRIDE) * sizeof(float);

// 128-deep GEMM of one 32-row A tile (registers) against all 4 N tiles.  The B fragments of the workgroup's current
// weight matrix are staged through LDS in 16 KiB chunks shared by its 4 waves (one L2 read per workgroup instead of
// one per wave; LDS latency instead of L2 latency in front of every 4 MFMAs): while chunk c is consumed from
// buffer c & 1, every thread fetches its 64 bytes of chunk c + 1 into registers and stores them to the other
// buffer afterwards.  One barrier per chunk.  `next` = first chunk of the matrix that follows (or nullptr).
__device__ __forceinline__ void gemm128_lds(const float4 (&a)[16], const float4 *__restrict__ B,
                                            const float4 *__restrict__ next, float4 *__restrict__ bufs, int &cur,
                                            int tid, int lane, floatx16 (&acc)[4]) {
#pragma unroll
    for (int ch = 0; ch < NP_CHUNKS; ++ch) {
        const float4 *src = ch + 1 < NP_CHUNKS ? B + (size_t)(ch + 1) * NP_CHUNK_F4 : next;
        if (src) {
            // async global -> LDS copy of the next chunk (no staging registers): each wave-instruction moves
            // 64 x 16 B to a wave-uniform LDS base + lane * 16; the barrier below waits for it (vmcnt(0)).
            float4 *dst = bufs + (cur ^ 1) * NP_CHUNK_F4 + (tid & ~63);
            const int nthr = blockDim.x;             // 256 (4 waves) or 64 (1 wave, small launches)
            for (int u = 0; u < NP_CHUNK_F4; u += nthr) td_glds16(src + u + tid, dst + u);
        }
        const float4 *bl = bufs + cur * NP_CHUNK_F4 + lane;
#pragma unroll
        for (int s = 0; s < NP_CHUNK_STEPS; ++s) {
            const float4 b = bl[s * 64];
            const float4 am = a[(ch * NP_CHUNK_STEPS + s) >> 2];
            const float av = (s & 3) == 0 ? am.x : (s & 3) == 1 ? am.y : (s & 3) == 2 ? am.z : am.w;
            acc[0] = td_mfma(av, b.x, acc[0]);
            acc[1] = td_mfma(av, b.y, acc[1]);
            acc[2] = td_mfma(av, b.z, acc[2]);
            acc[3] = td_mfma(av, b.w, acc[3]);
            if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // keep at most 4 B fragments (16 VGPRs) in flight
        }
        __syncthreads();
        cur ^= 1;
    }
}

// ---- exact 3-way bf16 splitting of both GEMM operands (model option node_proj_split, default on) ------------------------
// An fp32 significand (24 bits) is exactly the sum of three bf16 pieces (8 bits each): x = x1 + x2 + x3.  The six largest of
// the nine piece products (x1y1, x1y2, x2y1, x1y3, x3y1, x2y2) on v_mfma_f32_32x32x16_bf16, accumulated in fp32, reproduce
// the fp32 product to ~2e-7 relative at 16/6 of the fp32 MFMA rate.  B is pre-split at pack time (pack.cpp pack_B128_split:
// chunks of [k-step 4][piece 3][tile 2][lane 64] x 8 bf16, the 8 k-slots of lane half `hi` in k-step s being
// k = 16s + 8(j >> 2) + 4hi + (j & 3), i.e. exactly the two float4 of the fp32 A tile), A is split once per tile in registers.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
constexpr int NPS_CHUNK_U4 = 4 * 3 * 2 * 64;            // uint4 per staged chunk: 4 k-steps x 3 pieces x 2 tiles x 64 lanes (24 KiB)
constexpr int NPS_CHUNKS = 4;
constexpr int NPS_BIAS_FLOATS = 6 * TD_H;               // the stage's six bias vectors (read from LDS: a global load in the round
                                                        // loop would make the compiler wait on vmcnt, i.e. on the async B copy)
constexpr size_t NPS_LDS_BYTES = (size_t)2 * NPS_CHUNK_U4 * 16 + (size_t)(4 * 32 * NP_TSTRIDE + NPS_BIAS_FLOATS) * sizeof(float);

__device__ __forceinline__ floatx16 td_mfma_bf16(uint4 a, uint4 b, floatx16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned td_cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ void td_split2(float x, float y, unsigned &p1, unsigned &p2, unsigned &p3) {
    p1 = td_cvt_pk_bf16(x, y);
    float rx = x - __uint_as_float(p1 << 16), ry = y - __uint_as_float(p1 & 0xffff0000u);      // exact in fp32
    p2 = td_cvt_pk_bf16(rx, ry);
    rx -= __uint_as_float(p2 << 16);
    ry -= __uint_as_float(p2 & 0xffff0000u);
    p3 = td_cvt_pk_bf16(rx, ry);
}
// A tile (fp32, a[m] = h[row][8m + 4hi .. + 3]) -> three bf16 pieces per k-step
__device__ __forceinline__ void td_split_tile(const float4 (&a)[16], uint4 (&ap)[3][8]) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const float4 u = a[2 * s], v = a[2 * s + 1];
        td_split2(u.x, u.y, ap[0][s].x, ap[1][s].x, ap[2][s].x);
        td_split2(u.z, u.w, ap[0][s].y, ap[1][s].y, ap[2][s].y);
        td_split2(v.x, v.y, ap[0][s].z, ap[1][s].z, ap[2][s].z);
        td_split2(v.z, v.w, ap[0][s].w, ap[1][s].w, ap[2][s].w);
    }
}
// One launch processes up to three row segments, each with its own stage weights, row list and outputs:
//   mask: bit m (0..3) -> projection m of [k_i, k_j, v_i, v_j]; bit 4 -> query MLP
//   rows != nullptr: only the listed node ids; count_ptr != nullptr: device-side list length (N is then the bound the
//   grid was sized for, workgroups beyond the count exit before any barrier)
//   units > 1: one unit (a projection, or the two-GEMM query MLP) per workgroup, so that a 2 k-row segment spreads over
//   many CUs instead of walking five matrices on a few
// Segments are laid out along blockIdx.x in the order given: the segment with the long workgroups (six GEMMs per 128 rows)
// first, the short ones (one or two GEMMs) after it.  A full-size launch at C2 is 475 long + ~210 short workgroups for 512
// slots (2 per CU): dispatched in this order the short ones fill the slots the long ones leave, and the launch lasts about
// one long workgroup; with the short ones first (rounds 1-2) a third of the long workgroups started late (1.33 x).  This is
// how the h2x stage's projections (neighbourhood rows + ligand rows, h2x weights) ride in the same launch as the next
// layer's x2h-stage projections: all of them read the same h.
struct NpSeg {
    TdNodeStage st;
    const int32_t *rows;
    const int32_t *count_ptr;
    int64_t N;
    float *P, *q;
    unsigned mask;
    int units;         // 1, or popcount(mask & 0x1f) for per-unit workgroups
    int blocks;        // workgroups of this segment
};
struct NpArgs {
    NpSeg seg[3];
    int nseg;
};

__global__ __launch_bounds__(256, 2) void node_proj_kernel(NpArgs args, const float *__restrict__ h) {
    int bx = blockIdx.x, si = 0;
    while (si + 1 < args.nseg && bx >= args.seg[si].blocks) { bx -= args.seg[si].blocks; ++si; }
    const NpSeg &sg = args.seg[si];
    const TdNodeStage st = sg.st;
    const int32_t *__restrict__ rows = sg.rows;
    float *__restrict__ P = sg.P, *__restrict__ q = sg.q;
    unsigned mat_mask = sg.mask;
    int64_t N = sg.N;
    int unit = -1;
    if (sg.units > 1) {
        unit = bx % sg.units;
        bx /= sg.units;
    }
    if (sg.count_ptr) {
        N = *sg.count_ptr;
        if ((int64_t)bx * (blockDim.x >> 1) >= N) return;
    }
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float4 *bufs = reinterpret_cast<float4 *>(lds);                          // 2 x 16 KiB B chunks
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *tb = lds + 2 * NP_CHUNK_F4 * 4 + wave * 32 * NP_TSTRIDE;   // wave-private transpose tile
    const int c = lane & 31, hi = lane >> 5;
    const int64_t row0 = ((int64_t)bx * (blockDim.x >> 6) + wave) * 32;
    const int64_t aslot = row0 + c;
    const int64_t arow = aslot < N ? (rows ? (int64_t)rows[aslot] : aslot) : -1;

    float4 a[16];
#pragma unroll
    for (int m = 0; m < 16; ++m)
        a[m] = (arow >= 0) ? *reinterpret_cast<const float4 *>(h + arow * TD_H + 8 * m + 4 * hi)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
    // output row of C-layout register row r (recomputed at store time: 16 fewer live VGPRs)
    auto out_row = [&](int r) -> int {
        int slot = (int)row0 + td_erow(r, hi);
        asm volatile("" : "+v"(slot));      // keep the row-list address out of the loop-invariant set (it would be spilled)
        return slot < N ? (rows ? rows[slot] : slot) : -1;
    };

    // Small launches are split by matrix unit (sg.units > 1): one independent unit (a projection, or the two-GEMM query
    // MLP) per workgroup, so that a 3 k-row launch fills the chip instead of 28 CUs.
    if (unit >= 0) {
        int seen = 0;
        unsigned sel = 0;
        for (int b = 0; b < 5; ++b)
            if ((mat_mask >> b) & 1u) {
                if (seen == unit) sel = 1u << b;
                ++seen;
            }
        mat_mask = sel;
    }
    // the matrices this launch walks through, in order: selected projections, then q.net.0, then q.net.3
    constexpr size_t MAT_F4 = (size_t)TD_KSTEPS * 64;     // 16-byte units per matrix
    const float4 *Bp = reinterpret_cast<const float4 *>(st.projB);
    const float4 *seq[6];
    int mats[6], nseq = 0;
    for (int mat = 0; mat < 5; ++mat)
        if ((mat_mask >> mat) & 1u) { seq[nseq] = Bp + (size_t)mat * MAT_F4; mats[nseq++] = mat; }
    if ((mat_mask >> 4) & 1u) { seq[nseq] = reinterpret_cast<const float4 *>(st.q3B); mats[nseq++] = 5; }
    // prologue: first chunk of the first matrix
    for (int u = tid; u < NP_CHUNK_F4; u += blockDim.x) bufs[u] = seq[0][u];
    __syncthreads();
    int cur = 0;
    floatx16 acc[4];
    for (int si = 0; si < nseq; ++si) {
        const int mat = mats[si];
        const float *bias = mat < 5 ? st.projBias + mat * TD_H : st.q3Bias;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float bv = bias[32 * t + c];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = bv;
        }
        const float4 *next = si + 1 < nseq ? seq[si + 1] : nullptr;
        gemm128_lds(a, seq[si], next, bufs, cur, tid, lane, acc);     // for q.net.3 `a` holds the normalised hidden tile
        if (mat < 4) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int orow = out_row(r);
                if (orow >= 0) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) P[(size_t)orow * (4 * TD_H) + mat * TD_H + 32 * t + c] = acc[t][r];
                }
            }
        } else if (mat == 4) {
            // ---- query MLP: LayerNorm -> ReLU on the first Linear's output, then C layout -> A layout (into a) ---------
            float gam[4], bet[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                gam[t] = st.qGamma[32 * t + c];
                bet[t] = st.qBeta[32 * t + c];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float s1 = (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]);
                const float mean = td_sum32(s1) * (1.0f / TD_H);
                const float d0 = acc[0][r] - mean, d1 = acc[1][r] - mean, d2 = acc[2][r] - mean, d3 = acc[3][r] - mean;
                const float var = td_sum32((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) * (1.0f / TD_H);
                const float rstd = __frsqrt_rn(var + 1e-5f);
                acc[0][r] = fmaxf(d0 * rstd * gam[0] + bet[0], 0.f);
                acc[1][r] = fmaxf(d1 * rstd * gam[1] + bet[1], 0.f);
                acc[2][r] = fmaxf(d2 * rstd * gam[2] + bet[2], 0.f);
                acc[3][r] = fmaxf(d3 * rstd * gam[3] + bet[3], 0.f);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {          // one 32-column tile at a time through the wave-private tile
#pragma unroll
                for (int r = 0; r < 16; ++r) tb[td_erow(r, hi) * NP_TSTRIDE + c] = acc[t][r];
#pragma unroll
                for (int mm = 0; mm < 4; ++mm)
                    a[4 * t + mm] = *reinterpret_cast<const float4 *>(tb + c * NP_TSTRIDE + 8 * mm + 4 * hi);   // h tile no longer needed
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int orow = out_row(r);
                if (orow >= 0) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) q[(size_t)orow * TD_H + 32 * t + c] = acc[t][r];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ split path, half-N GEMMs
// The bf16 x 3 kernel.  Register budget is what shapes it: the A tile's piece triples take 96 VGPRs for the whole
// workgroup lifetime, so a GEMM runs as two half-N GEMMs (2 N tiles = 32 accumulator VGPRs, 6 B fragments in flight) and
// the wave stays under 256 VGPRs without spilling at 2 waves per SIMD.  B stream per matrix (packed by pack_B128_split):
// 4 chunks of 24 KiB = [half 2][k-chunk 2] x [k-step 4][piece 3][tile 2][lane 64] x 16 B, consumed in that order.
// ASYNC (default): the next B chunk is copied global -> LDS by global_load_lds issued from inline assembly, and the round
// ends with an explicit s_waitcnt vmcnt(0) in front of its barrier, so the copy flies behind the round's 48 MFMAs.  Issued
// through the builtin (ASYNC = false, model option node_proj_async = 0) the compiler cannot tell the copy's LDS
// destination from the buffer being read and puts s_waitcnt vmcnt(0) in front of the round's first ds_read: every one of the
// 24 rounds then starts by waiting out a full L2 round trip (98 us per full-size launch against 33 us of MFMA time, round 2).
// (Staging through registers instead costs 24 VGPRs the kernel does not have: the chunk lands in scratch.)
// The source is a scalar base + the lane's byte offset: the six copies of a round then need no address arithmetic on the
// vector unit (with a 64-bit vector address each copy is preceded by a v_lshl_add_u64 into the register pair the previous copy still reads
// from, and all of it competes for vector issue with the partner wave's products): node projections 0.702 -> 0.696 ms per C2 step, C3 10.48 -> 10.31
__device__ __forceinline__ void td_glds16_asm_s(const uint4 *gsrc_base, uint32_t lane_bytes, uint32_t lds_wave_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(lane_bytes), "s"(gsrc_base), "s"(lds_wave_base) : "memory", "m0");      // m0 named as clobbered: the compiler sets it itself for v_movrel / readlane selects / its own LDS-DMA (clang calls it "reserved" and warns: -Wno-inline-asm)
}
template <bool ASYNC, bool BPIPE>
__global__ __launch_bounds__(256, 2) void node_proj_split_kernel(NpArgs args, const float *__restrict__ h) {
    int bx = blockIdx.x, si = 0;
    while (si + 1 < args.nseg && bx >= args.seg[si].blocks) { bx -= args.seg[si].blocks; ++si; }
    const NpSeg &sg = args.seg[si];
    const TdNodeStage st = sg.st;
    const int32_t *__restrict__ rows = sg.rows;
    float *__restrict__ P = sg.P, *__restrict__ q = sg.q;
    unsigned mat_mask = sg.mask;
    int64_t N = sg.N;
    int unit = -1;
    if (sg.units > 1) {
        unit = bx % sg.units;
        bx /= sg.units;
    }
    if (sg.count_ptr) {
        N = *sg.count_ptr;
        if ((int64_t)bx * (blockDim.x >> 1) >= N) return;
    }
    extern __shared__ __attribute__((aligned(16))) float lds[];
    uint4 *bufs = reinterpret_cast<uint4 *>(lds);                                  // 2 x 24 KiB B chunks
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *tb = lds + 2 * NPS_CHUNK_U4 * 4 + wave * 32 * NP_TSTRIDE;               // wave-private transpose tile
    const int c = lane & 31, hi = lane >> 5;
    const int64_t row0 = ((int64_t)bx * (blockDim.x >> 6) + wave) * 32;
    if (unit >= 0) {          // one matrix unit per workgroup (small launches / split segments)
        int seen = 0;
        unsigned sel = 0;
        for (int b = 0; b < 5; ++b)
            if ((mat_mask >> b) & 1u) {
                if (seen == unit) sel = 1u << b;
                ++seen;
            }
        mat_mask = sel;
    }
    // matrices in order: selected projections (0..3), q.net.0 (4), q.net.3 (5) -- scalar bit tests, no indexed arrays
    const unsigned seq_mask = (mat_mask & 0x1fu) | (((mat_mask >> 4) & 1u) << 5);
    if (!seq_mask) return;
    constexpr size_t MAT_U4 = (size_t)NPS_CHUNKS * NPS_CHUNK_U4;
    const uint4 *Bp = reinterpret_cast<const uint4 *>(st.projB3), *Bq3 = reinterpret_cast<const uint4 *>(st.q3B3);
    auto mat_after = [&](int m) -> int {
        const unsigned rest = seq_mask & ~((2u << m) - 1u);
        return rest ? __builtin_ctz(rest) : -1;
    };
    auto mat_ptr = [&](int m) -> const uint4 * { return m < 5 ? Bp + (size_t)m * MAT_U4 : Bq3; };
    int mat = __builtin_ctz(seq_mask);
    float *sbias = lds + 2 * NPS_CHUNK_U4 * 4 + 4 * 32 * NP_TSTRIDE;
    {   // prologue: first chunk of the first matrix, the bias vectors
        const uint4 *first = mat_ptr(mat);
        for (int u = tid; u < NPS_CHUNK_U4; u += blockDim.x) bufs[u] = first[u];
        for (int u = tid; u < NPS_BIAS_FLOATS; u += blockDim.x) sbias[u] = u < 5 * TD_H ? st.projBias[u] : st.q3Bias[u - 5 * TD_H];
    }
    uint4 ap[3][8];
    // Node ids of the rows a lane stores, taken once from the A-tile lanes (lane c holds the id of tile row c; -1 beyond the
    // segment's rows) by lane exchange.  Fetched from the row list at store time (rounds 1-2), every one of the 16 loads sat in
    // front of the two stores whose addresses it produces, and its s_waitcnt vmcnt(0) also waited for the previous pair of
    // stores to be acknowledged: 7-10 k cycles per half-matrix epilogue on a row-list launch (in-kernel stamps, round 3).
    int orow4[4];              // node id of tile row (lane >> 3) + 8 k: the rows this lane stores in every epilogue
    {
        const int64_t aslot = row0 + c;
        const int64_t arow = aslot < N ? (rows ? (int64_t)rows[aslot] : aslot) : -1;
#pragma unroll
        for (int k = 0; k < 4; ++k) orow4[k] = __shfl((int)arow, (lane >> 3) + 8 * k);
        float4 a[16];
#pragma unroll
        for (int m = 0; m < 16; ++m)
            a[m] = (arow >= 0) ? *reinterpret_cast<const float4 *>(h + arow * TD_H + 8 * m + 4 * hi) : make_float4(0.f, 0.f, 0.f, 0.f);
        td_split_tile(a, ap);
    }
    __syncthreads();
    int cur = 0;
    const int64_t row0s = ((int64_t)bx * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(wave)) * 32;      // row0, as a scalar
    const bool direct = !rows && row0s + 32 <= N;       // wave-uniform: the epilogues' buffer-store path
    floatx16 keep[2];          // q.net.0: columns 0..63 while the second half is computed (LayerNorm needs the whole row)
    for (; mat >= 0; mat = mat_after(mat)) {
        const float *bias = sbias + mat * TD_H;
        const uint4 *B = mat_ptr(mat);
        const int mat_next = mat_after(mat);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            floatx16 acc[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float bv = bias[64 * half + 32 * t + c];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = bv;
            }
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                const int chunk = 2 * half + kc;
                const uint4 *src = chunk + 1 < NPS_CHUNKS ? B + (size_t)(chunk + 1) * NPS_CHUNK_U4
                                                            : (mat_next >= 0 ? mat_ptr(mat_next) : nullptr);
                if constexpr (ASYNC) {
                    if (src) {
                        const uint32_t lbase = __builtin_amdgcn_readfirstlane(
                            (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)(bufs + (cur ^ 1) * NPS_CHUNK_U4 + (tid & ~63)));
#pragma unroll
                        for (int u = 0; u < NPS_CHUNK_U4 / 256; ++u) {
                            td_glds16_asm_s(src + u * 256, (uint32_t)tid * 16u, lbase + u * 256 * 16);
                        }
                    }
                } else if (src) {
                    uint4 *dst = bufs + (cur ^ 1) * NPS_CHUNK_U4 + (tid & ~63);
                    for (int u = 0; u < NPS_CHUNK_U4; u += 256)
                        td_glds16(reinterpret_cast<const float4 *>(src + u + tid), reinterpret_cast<float4 *>(dst + u));
                }
                const uint4 *bl = bufs + cur * NPS_CHUNK_U4 + lane;
                // B fragments of k-step ss + 1 are read from LDS BEFORE the 12 MFMAs of k-step ss are issued (BPIPE: two
                // register sets; left to itself the scheduler issues the reads after them and the wave then waits out the
                // LDS latency with an empty matrix pipe)
                uint4 b[2][3][2];
                auto load_b = [&](int ss, uint4 (&bb)[3][2]) {
#pragma unroll
                    for (int p = 0; p < 3; ++p)
#pragma unroll
                        for (int t = 0; t < 2; ++t) bb[p][t] = bl[((ss * 3 + p) * 2 + t) * 64];
                };
                if constexpr (BPIPE) load_b(0, b[0]);
#pragma unroll
                for (int ss = 0; ss < 4; ++ss) {
                    const int s = 4 * kc + ss;
                    uint4 (&bc)[3][2] = b[BPIPE ? (ss & 1) : 0];
                    if constexpr (BPIPE) {
                        if (ss + 1 < 4) load_b(ss + 1, b[(ss + 1) & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                    } else
                        load_b(ss, bc);
                    // low-order products first
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[t] = td_mfma_bf16(ap[1][s], bc[1][t], acc[t]);
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[t] = td_mfma_bf16(ap[2][s], bc[0][t], acc[t]);
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[t] = td_mfma_bf16(ap[0][s], bc[2][t], acc[t]);
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[t] = td_mfma_bf16(ap[1][s], bc[0][t], acc[t]);
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[t] = td_mfma_bf16(ap[0][s], bc[1][t], acc[t]);
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[t] = td_mfma_bf16(ap[0][s], bc[0][t], acc[t]);
                    if constexpr (BPIPE) __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (ASYNC) asm volatile("s_waitcnt vmcnt(0)" : : : "memory");     // the copy the compiler does not know about
                __syncthreads();
                cur ^= 1;
            }
            // ---- epilogue of this half: columns 64 * half .. + 63
            if (mat != 4) {
                // 32 x 32 tile at a time through the wave-private LDS tile: C layout in (a register row per store would be 32
                // dword stores of 2 x 128 B per half-matrix), rows out -- each lane stores 16 contiguous bytes, a wave
                // instruction 8 full 128-byte row segments: 8 stores per half-matrix
                float *out = mat < 4 ? P + mat * TD_H + 64 * half : q + 64 * half;
                const size_t ld = mat < 4 ? (size_t)(4 * TD_H) : (size_t)TD_H;
                if (direct) {
                    // A full tile of consecutive node ids leaves straight from the C layout: a register is two 128-byte row segments (lanes
                    // 0 .. 31 row erow(r, 0), lanes 32 .. 63 four rows further).  Buffer stores: the tile's extent in the output as the
                    // resource and the row as the offset, both in scalar registers, the lane's place in the two segments as a vector offset
                    // that never changes -- no trip through the LDS tile (32 ds_write_b32 + 8 ds_read_b128 and their waits per half-matrix)
                    // and nothing per store on the vector unit.  Node projections 0.710 -> 0.686 ms per C2 step, C3 10.29 -> 10.07.
                    // (Row-list tiles the same way, with the lane's 16 row offsets in registers: no better than the LDS tile, 0.720 -> 0.703
                    // with everything direct -- left on the tile.)
                    auto store_tiles = [&](float *base, auto ldc) {
                        constexpr int LD = (int)decltype(ldc)::value;
                        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base + (size_t)row0s * LD, 0, 32 * LD * 4, 0x00020000);
                        const int voff = ((4 * hi) * LD + c) * 4;
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[t][r]), rs, voff, (td_erow(r, 0) * LD + 32 * t) * 4, 0);
                    };
                    if (mat < 4) store_tiles(out, std::integral_constant<int, 4 * TD_H>());
                    else store_tiles(out, std::integral_constant<int, TD_H>());
                } else
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) tb[td_erow(r, hi) * NP_TSTRIDE + c] = acc[t][r];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float4 v = *reinterpret_cast<const float4 *>(tb + ((lane >> 3) + 8 * k) * NP_TSTRIDE + 4 * (lane & 7));
                        if (orow4[k] >= 0) *reinterpret_cast<float4 *>(out + (size_t)orow4[k] * ld + 32 * t + 4 * (lane & 7)) = v;
                    }
                }
            } else if (half == 0) {
                keep[0] = acc[0];
                keep[1] = acc[1];
            } else {
                // ---- query MLP: LayerNorm -> ReLU over the 128 columns (keep = 0..63, acc = 64..127), then C layout -> A
                //      layout through the wave-private tile and a fresh split of the hidden tile
                float gam[4], bet[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    gam[t] = st.qGamma[32 * t + c];
                    bet[t] = st.qBeta[32 * t + c];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float s1 = (keep[0][r] + keep[1][r]) + (acc[0][r] + acc[1][r]);
                    const float mean = td_sum32(s1) * (1.0f / TD_H);
                    const float d0 = keep[0][r] - mean, d1 = keep[1][r] - mean, d2 = acc[0][r] - mean, d3 = acc[1][r] - mean;
                    const float var = td_sum32((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) * (1.0f / TD_H);
                    const float rstd = __frsqrt_rn(var + 1e-5f);
                    keep[0][r] = fmaxf(d0 * rstd * gam[0] + bet[0], 0.f);
                    keep[1][r] = fmaxf(d1 * rstd * gam[1] + bet[1], 0.f);
                    acc[0][r] = fmaxf(d2 * rstd * gam[2] + bet[2], 0.f);
                    acc[1][r] = fmaxf(d3 * rstd * gam[3] + bet[3], 0.f);
                }
                float4 a[16];
#pragma unroll
                for (int t = 0; t < 4; ++t) {          // one 32-column tile at a time through the wave-private tile
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        tb[td_erow(r, hi) * NP_TSTRIDE + c] = t == 0 ? keep[0][r] : t == 1 ? keep[1][r] : t == 2 ? acc[0][r] : acc[1][r];
#pragma unroll
                    for (int mm = 0; mm < 4; ++mm)
                        a[4 * t + mm] = *reinterpret_cast<const float4 *>(tb + c * NP_TSTRIDE + 8 * mm + 4 * hi);
                }
                td_split_tile(a, ap);
            }
        }
    }
}

static int np_fill(NpSeg &g, const TdNodeStage &st, const int32_t *rows, const int32_t *count_ptr, int64_t N, unsigned mask,
                   float *P, float *q, bool split_units, int rows_per_wg) {
    g.st = st; g.rows = rows; g.count_ptr = count_ptr; g.N = N; g.P = P; g.q = q; g.mask = mask;
    g.units = split_units ? __builtin_popcount(mask & 0x1fu) : 1;
    g.blocks = (int)((N + rows_per_wg - 1) / rows_per_wg) * g.units;
    return g.blocks;
}

static TdLdsOnce g_np_lds, g_nps_lds, g_npsa_lds, g_npsb_lds, g_egnn_node_lds;

static int np_launch(const NpArgs &a, const float *h, unsigned total_blocks, hipStream_t s) {
    int rc;
    // bf16 x 3 operand split (exact: an fp32 significand is three 8-bit pieces; fp32 accumulation) when every stage of the
    // launch carries the pre-split weights and the model option asks for it
    bool split = true;
    for (int i = 0; i < a.nseg; ++i) split = split && a.seg[i].st.use_split && a.seg[i].st.projB3 != nullptr;
    if (split && a.seg[0].st.async_copy && a.seg[0].st.bpipe) {
        if ((rc = td_set_lds(g_nps_lds, reinterpret_cast<const void *>(node_proj_split_kernel<true, true>), NPS_LDS_BYTES)) != TD_OK) return rc;
        node_proj_split_kernel<true, true><<<dim3(total_blocks), dim3(256), NPS_LDS_BYTES, s>>>(a, h);
    } else if (split && a.seg[0].st.async_copy) {
        if ((rc = td_set_lds(g_npsb_lds, reinterpret_cast<const void *>(node_proj_split_kernel<true, false>), NPS_LDS_BYTES)) != TD_OK) return rc;
        node_proj_split_kernel<true, false><<<dim3(total_blocks), dim3(256), NPS_LDS_BYTES, s>>>(a, h);
    } else if (split) {
        if ((rc = td_set_lds(g_npsa_lds, reinterpret_cast<const void *>(node_proj_split_kernel<false, false>), NPS_LDS_BYTES)) != TD_OK) return rc;
        node_proj_split_kernel<false, false><<<dim3(total_blocks), dim3(256), NPS_LDS_BYTES, s>>>(a, h);
    } else {
        if ((rc = td_set_lds(g_np_lds, reinterpret_cast<const void *>(node_proj_kernel), NP_LDS_BYTES)) != TD_OK) return rc;
        node_proj_kernel<<<dim3(total_blocks), dim3(256), NP_LDS_BYTES, s>>>(a, h);
    }
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

int td_launch_node_proj(const TdNodeStage &st, const float *h, int64_t N, const int32_t *rows, unsigned mat_mask,
                        float *P, float *q, hipStream_t s, const int32_t *count_ptr, const int32_t *rows2, int64_t N2,
                        unsigned mask2) {
    if (N == 0 || mat_mask == 0) return TD_OK;
    if (!rows2 || N2 == 0 || (mask2 & 0x1fu) == 0) { rows2 = nullptr; N2 = 0; mask2 = 0; }
    NpArgs a;
    a.nseg = 0;
    unsigned total = 0;
    // a small batch (N bounds the device-side count) cannot fill the chip with one workgroup per 128 rows walking through
    // all six GEMMs: one workgroup per (128 rows, matrix unit) instead -- 5x the parallelism, same arithmetic
    total += np_fill(a.seg[a.nseg++], st, rows, count_ptr, N, mat_mask, P, q, N <= TD_SMALL_BATCH_ROWS, 128);
    if (rows2) total += np_fill(a.seg[a.nseg++], st, rows2, nullptr, N2, mask2, P, q, true, 128);
    return np_launch(a, h, total, s);
}

// The h2x-stage projections of layer l (stage `hx`: src-side units on `hop_rows`, dst-side units + queries on the ligand
// rows, into Px / qx) and the x2h-stage projections of layer l + 1 (stage `nx`, all units on `rows` or every node, into
// P / q) in one launch: both read the features the value pass of layer l just wrote.
int td_launch_node_proj_pair(const TdNodeStage &hx, const int32_t *hop_rows, const int32_t *hop_count,
                             const int32_t *lig_rows, int64_t Nl, float *Px, float *qx, const TdNodeStage &nx,
                             const int32_t *rows, const int32_t *count_ptr, float *P, float *q, const float *h, int64_t N,
                             hipStream_t s) {
    if (N == 0) return TD_OK;
    NpArgs a;
    a.nseg = 0;
    unsigned total = 0;
    // After the long workgroups: the ligand rows' single-unit workgroups (their query MLP, two chained GEMMs, is the longest short
    // one: it takes the slots the long ones leave free from the start), then the neighbourhood rows'.  When the long workgroups fit
    // the chip's 512 slots in one round (C2: 475), the neighbourhood rows mostly run once the long ones are done, on an empty chip,
    // where a second workgroup costs nothing and a second GEMM in the same workgroup is 10 us of tail: one unit per workgroup
    // (projections 0.765 -> 0.729 ms per C2 step).  Over many rounds (C3: 8,140 long workgroups) the second tile load costs
    // throughput instead (+2.5 %): both units in one workgroup.
    const bool one_round = (N + 127) / 128 <= 512;
    total += np_fill(a.seg[a.nseg++], nx, rows, rows ? count_ptr : nullptr, N, 0x1f, P, q, N <= TD_SMALL_BATCH_ROWS, 128);
    if (Nl > 0) total += np_fill(a.seg[a.nseg++], hx, lig_rows, nullptr, Nl, 0x15, Px, qx, true, 128);
    total += np_fill(a.seg[a.nseg++], hx, hop_rows, hop_rows ? hop_count : nullptr, N, 0x0a, Px, qx, one_round, 128);
    return np_launch(a, h, total, s);
}

// ------------------------------------------------------------------------------------------ EGNN node update
// h_i += W2 SiLU(W1 [mi | h_i] + b1) + b2   (models/egnn.py:56, node_mlp = Linear(256,128) -> SiLU -> Linear(128,128)).
// Same decomposition as node_proj_kernel: 32 rows per wave, A tiles in registers, B staged through LDS; the 256-deep first
// Linear is two 128-deep GEMMs into one accumulator (mi tile, then h tile).
// LN = true: node_output of BaseX2HAttLayer with out_fc (models/uni_transformer.py:39-40, 81-84): h_i += W2 relu(LayerNorm(W1 [out_i | h_i]
// + b1)) + b2, `mi` = the attention output of the row, gamma / beta = the MLP's LayerNorm affine.
template <bool LN>
__global__ __launch_bounds__(256, 2) void egnn_node_kernel(const float4 *__restrict__ B, const float *__restrict__ b1,
                                                           const float *__restrict__ b2, const float *__restrict__ mi,
                                                           float *__restrict__ h, int64_t N, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float4 *bufs = reinterpret_cast<float4 *>(lds);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *tb = lds + 2 * NP_CHUNK_F4 * 4 + wave * 32 * NP_TSTRIDE;
    const int c = lane & 31, hi = lane >> 5;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
    const int64_t arow = row0 + c < N ? row0 + c : -1;
    const float4 *B0 = B, *B1 = B + (size_t)TD_KSTEPS * 64, *B2 = B + (size_t)2 * TD_KSTEPS * 64;
    for (int u = tid; u < NP_CHUNK_F4; u += blockDim.x) bufs[u] = B0[u];
    __syncthreads();
    int cur = 0;
    float4 a[16];
    floatx16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float bv = b1[32 * t + c];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = bv;
    }
#pragma unroll
    for (int m = 0; m < 16; ++m)
        a[m] = arow >= 0 ? *reinterpret_cast<const float4 *>(mi + arow * TD_H + 8 * m + 4 * hi) : make_float4(0.f, 0.f, 0.f, 0.f);
    gemm128_lds(a, B0, B1, bufs, cur, tid, lane, acc);
#pragma unroll
    for (int m = 0; m < 16; ++m)
        a[m] = arow >= 0 ? *reinterpret_cast<const float4 *>(h + arow * TD_H + 8 * m + 4 * hi) : make_float4(0.f, 0.f, 0.f, 0.f);
    gemm128_lds(a, B1, B2, bufs, cur, tid, lane, acc);
    if constexpr (LN) {          // LayerNorm over the row's 128 columns (4 tiles x the 32 lanes of a half-wave) + ReLU, in place
        float gam[4], bet[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { gam[t] = gamma[32 * t + c]; bet[t] = beta[32 * t + c]; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float mean = td_sum32((acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r])) * (1.0f / TD_H);
            const float d0 = acc[0][r] - mean, d1 = acc[1][r] - mean, d2 = acc[2][r] - mean, d3 = acc[3][r] - mean;
            const float rstd = __frsqrt_rn(td_sum32((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) * (1.0f / TD_H) + 1e-5f);
            acc[0][r] = fmaxf(d0 * rstd * gam[0] + bet[0], 0.f);
            acc[1][r] = fmaxf(d1 * rstd * gam[1] + bet[1], 0.f);
            acc[2][r] = fmaxf(d2 * rstd * gam[2] + bet[2], 0.f);
            acc[3][r] = fmaxf(d3 * rstd * gam[3] + bet[3], 0.f);
        }
    }
    // SiLU (EGNN), then C layout -> A layout through the wave-private tile
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = acc[t][r];
            tb[td_erow(r, hi) * NP_TSTRIDE + c] = LN ? v : v * __frcp_rn(1.0f + __expf(-v));
        }
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) a[4 * t + mm] = *reinterpret_cast<const float4 *>(tb + c * NP_TSTRIDE + 8 * mm + 4 * hi);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float bv = b2[32 * t + c];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = bv;
    }
    gemm128_lds(a, B2, nullptr, bufs, cur, tid, lane, acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int slot = (int)row0 + td_erow(r, hi);
        asm volatile("" : "+v"(slot));
        if (slot < N) {
#pragma unroll
            for (int t = 0; t < 4; ++t) h[(size_t)slot * TD_H + 32 * t + c] += acc[t][r];
        }
    }
}

int td_launch_egnn_node(const TdEgnnLayer &L, const float *mi, float *h, int64_t N, hipStream_t s) {
    if (N == 0) return TD_OK;
    int rc;
    if ((rc = td_set_lds(g_egnn_node_lds, reinterpret_cast<const void *>(egnn_node_kernel<false>), NP_LDS_BYTES)) != TD_OK) return rc;
    egnn_node_kernel<false><<<dim3((unsigned)((N + 127) / 128)), dim3(256), NP_LDS_BYTES, s>>>(
        reinterpret_cast<const float4 *>(L.nodeB), L.nb1, L.nb2, mi, h, N, nullptr, nullptr);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

// node_output of an x2h stage with out_fc: h += MLP([out | h]) on every row (fp32 MFMA; a non-default configuration)
int td_launch_node_output(const TdNodeOut &no, const float *out, float *h, int64_t N, hipStream_t s) {
    if (N == 0) return TD_OK;
    static TdLdsOnce once;
    int rc;
    if ((rc = td_set_lds(once, reinterpret_cast<const void *>(egnn_node_kernel<true>), NP_LDS_BYTES)) != TD_OK) return rc;
    egnn_node_kernel<true><<<dim3((unsigned)((N + 127) / 128)), dim3(256), NP_LDS_BYTES, s>>>(
        reinterpret_cast<const float4 *>(no.B), no.b1, no.b2, out, h, N, no.gamma, no.beta);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}
