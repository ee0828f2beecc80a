// Attention passes (x2h key / value, h2x) on 16 x 16 MFMA tiles (gfx950), re-associated arithmetic: logits = z . U_i,
// out = W2v . (alpha^T z).  The 16-row tile matches the 16 attention heads exactly, so the two head-shaped products cost
// 64 x 32 cycles and the U_i build uses all 64 lanes.  The per-edge 128-deep products run on v_mfma_f32_16x16x4_f32; the
// 21-wide radial/type first layer runs on v_mfma_f32_16x16x32_bf16 with both operands as exact bf16 piece triples
// (td_first_layer_split16; model option edge_key_split, default) or on the fp32 instruction (td_first_layer_compute16).
//
// Lane coordinates: lo = lane & 15, g = lane >> 4.   16x16x4 fragment maps (cdna_hip_programming.md section 3):
//   A: lane holds A[row = lo][k = g]      B: lane holds B[k = g][col = lo]      C/D: reg r holds D[row = 4g + r][col = lo]
//
// First layer, transposed (rows = hidden units, columns = edges), 8 hidden blocks x 2 edge blocks of 16 x 16:
//   acc[eb][hb][r] = pre[hidden 16hb + 4g + r][edge 16eb + lo]
// so one lane owns, for each of its two edges, 32 of the 128 hidden units; the other 96 sit in lanes lo + 16 g'.
// LayerNorm = in-lane sums + two lane-group exchanges.  The normalised z^T is directly the B operand of the
// logits product (k-step (hb, r) pairs lane group g with hidden unit 16hb + 4g + r on both operands).
#include <type_traits>

#include "td_device.h"
#include "td_internal.h"

typedef float floatx4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ floatx4_t td_mfma16(float a, float b, floatx4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- the second layer on v_mfma_f32_16x16x32_f16 (round 6) -----------------------------------------------------------------------------
// The per-edge 128-deep products (logits = z . U_i, Zbar = alpha^T z) used to run on v_mfma_f32_16x16x4_f32: 64 instructions of 32 cycles
// per row -- a quarter of a pass -- on the fp32 matrix path, which IS the vector unit's rate and does not overlap with vector instructions.
// Both operands now go in as PAIRS of f16 pieces, x = h1 + h2 with h1 = x truncated to f16 and h2 = f16(x - h1) (round to nearest; the
// residual x - h1 is exact in fp32), and three piece products per tile -- h1 h1', h1 h2', h2 h1' -- replace the fp32 product with fp32
// accumulation: 24 instructions of 16 cycles on the matrix cores proper.  What it represents: the activations z'' and the attention
// weights are in [0, 1] by construction (FoldedMlp's clamp; softmax x gate) and U_i is scaled per row by a power of two to below 2^13, so
// every operand is carried to 22 significant bits, and never worse than 2^-25 of the operand's scale (f16 subnormals, which the
// instruction honours exactly: tools/microbench/f16_split_probe.hip); the dropped product h2 h2' is below 2^-22 relative.  The fp32 form
// rounds each of its 128 (32) partial sums to 24 bits, so the two differ by about one fp32 rounding of the result -- measured on the
// goldens of the real reference incl. its float64 runs (profiles/r06*_second_layer_error.txt).  Model option "edge_second_layer_f16"
// (default 1; 0 = the fp32 products) selects between the instantiations.
#ifndef TD_ZPLAIN_OFF
#define TD_ZPLAIN_OFF 0   // 1: every MLP takes the scaled pieces (A/B of the two conversions)
#endif
#ifndef TD_ABL
#define TD_ABL 0          // timing ablations of the key pass (wrong results; EXPERIMENTS.md round 6): 1 no U_i FMAs, 2 no Wq reads, 3 no z'' split, 4 no first-layer products, 5 no P_j gathers, 6 no logits products
#endif
typedef _Float16 half8_16 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ floatx4_t td_mfma16h(uint4 a, uint4 b, floatx4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_16, a), __builtin_bit_cast(half8_16, b), c, 0, 0, 0);
}
// (x, y) -> packed f16 pairs h1 = (trunc x, trunc y), h2 = (x - h1.x, y - h1.y) rounded to nearest: v_cvt_pkrtz_f16_f32 and one
// v_fma_mix{lo,hi}_f16 per value (the f16 piece as a source of an fp32 FMA whose result is written back as f16)
__device__ __forceinline__ void td_split_h2(float x, float y, unsigned &h1, unsigned &h2) {
    h1 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x, y));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(h2) : "v"(h1), "v"(x), "v"(y));
}
// The same for values in [0, 1] that may sit far below 1 (z'' = activation / (|gamma| M): around 1 / M, and M grows with the largest
// LayerNorm bias / |weight| of the MLP): the pieces are taken of x S with S = 2^15, exactly, so that 22 bits survive down to x = 2^-18 instead
// of running into the f16 subnormal floor at 2^-25 -- h1 = f16(x S) (round to nearest), h2 = f16(x S - h1): two v_fma_mix per piece pair
// member, S as a scalar operand.  The consumer takes 1 / S off its result.
constexpr float TD_Z_SCALE = 32768.0f;
__device__ __forceinline__ void td_split_h2_scaled(float x, float y, unsigned &h1, unsigned &h2) {
    const float S = TD_Z_SCALE;
    asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(h1), "=&v"(h2) : "v"(x), "v"(y), "s"(S));
}
// Four pairs at once, the sixteen instructions interleaved so that no instruction reads the register the one before it wrote (a pair's four
// conversions are a dependent chain through the two halves of h1 and h2: issued back to back, as one statement per pair does, every one of
// them waits out its predecessor's latency -- PMC had 7.6 cycles of SQ_ACTIVE_INST_VALU per added instruction against 4.1 for the rest).
#ifndef TD_SPLIT_X4
#define TD_SPLIT_X4 1
#endif
__device__ __forceinline__ void td_split_h2_scaled_x4(const float (&x)[4], const float (&y)[4], unsigned (&h1)[4], unsigned (&h2)[4]) {
#if TD_SPLIT_X4
    const float S = TD_Z_SCALE;
    asm("v_fma_mixlo_f16 %0, %8, %16, 0\n\t"
        "v_fma_mixlo_f16 %1, %9, %16, 0\n\t"
        "v_fma_mixlo_f16 %2, %10, %16, 0\n\t"
        "v_fma_mixlo_f16 %3, %11, %16, 0\n\t"
        "v_fma_mixhi_f16 %0, %12, %16, 0\n\t"
        "v_fma_mixhi_f16 %1, %13, %16, 0\n\t"
        "v_fma_mixhi_f16 %2, %14, %16, 0\n\t"
        "v_fma_mixhi_f16 %3, %15, %16, 0\n\t"
        "v_fma_mixlo_f16 %4, %8, %16, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %5, %9, %16, -%1 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %6, %10, %16, -%2 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %7, %11, %16, -%3 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %4, %12, %16, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %5, %13, %16, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %6, %14, %16, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %7, %15, %16, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(h1[0]), "=&v"(h1[1]), "=&v"(h1[2]), "=&v"(h1[3]), "=&v"(h2[0]), "=&v"(h2[1]), "=&v"(h2[2]), "=&v"(h2[3])
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "s"(S));
#else
#pragma unroll
    for (int m = 0; m < 4; ++m) td_split_h2_scaled(x[m], y[m], h1[m], h2[m]);
#endif
}
// (plain form, four pairs: the truncations by the compiler's own v_cvt_pkrtz_f16_f32, the residuals interleaved)
__device__ __forceinline__ void td_split_h2_x4(const float (&x)[4], const float (&y)[4], unsigned (&h1)[4], unsigned (&h2)[4]) {
#if TD_SPLIT_X4
#pragma unroll
    for (int m = 0; m < 4; ++m) h1[m] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x[m], y[m]));
    asm("v_fma_mixlo_f16 %0, %4, -1.0, %8 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %5, -1.0, %9 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %2, %6, -1.0, %10 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %3, %7, -1.0, %11 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %4, -1.0, %12 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %5, -1.0, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %2, %6, -1.0, %14 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %3, %7, -1.0, %15 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(h2[0]), "=&v"(h2[1]), "=&v"(h2[2]), "=&v"(h2[3])
        : "v"(h1[0]), "v"(h1[1]), "v"(h1[2]), "v"(h1[3]), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]));
#else
#pragma unroll
    for (int m = 0; m < 4; ++m) td_split_h2(x[m], y[m], h1[m], h2[m]);
#endif
}

constexpr float TD_ATT_SCALE_16 = 0.35355339059327373f;   // 1/sqrt(8)
constexpr float TD_LOG2E = 1.4426950408889634f;
constexpr float TD_FAR_CENTRE = 1.0e4f;                    // "centre" of the K slots behind the 20 Gaussians: exp2(c (d - 1e4)^2) = 0
constexpr int E16_STEPS = TD_SLOTK / 4;                    // 6 k-steps of 4 over the 24-wide radial/type slot
constexpr int E16_R_FLOATS = 2 * 2 * E16_STEPS * 64 * 8;   // [cls][slot][step][lane][hb]   = 12288
constexpr int E16_WQ_FLOATS = 8 * 4 * 2 * 64 * 4;          // [hb][r][jq][lane][4 j]        = 16384

struct Args16 {
    const float4 *x4;
    const int32_t *nbr;
    const float *ew;
    const float *P;
    const float *q;
    const int32_t *rows;
    const int32_t *count_ptr;
    float *h;
    const float *gate_m;       // value pass, ew_net_type 'm': u' [128] = folded W2v^T w_m, then c = w_m . b2v + b_m (the gate from the value vector)
    float *out;                // value pass, x2h_out_fc: the attention output goes here WITHOUT the residual (nullptr: h += output)
    float *alpha;
    float4 *x4_out;            // XV mode: updated coordinates of the dst (ligand) nodes
    int64_t count;
    TdEdgeMlp mlp;
    const float *offsets;
    float coeff;
    int p_off;
    // general graphs: the in-edges of node i are the chunks cptr[i] .. cptr[i+1]-1 of nbr / ew / alpha (nullptr: chunk == node)
    const int32_t *chunk_node;
    const int32_t *cptr;
    // value pass with the bf16 first layer: the ligand rows (all of them are in the row list)
    const int32_t *lig_rows;
    int64_t lig_count;
    // general graphs, value pass: chunks per protein row and the chunks of all ligand rows together (the class split is sized
    // from the chunk counts: a hybrid ligand row has several times the chunks of a protein row)
    int cpn_p;
    int64_t lig_chunks;
    // value pass: device count of the rows that see both source classes (a session's dirty rows: the protein rows with a ligand atom
    // among their neighbours + the ligand rows), or nullptr (a typical share is assumed)
    const int32_t *mixed_count;
    int deal;                      // rows dealt round-robin inside an XCD's range (td_deal16; model option edge_row_dealing)
    unsigned long long *trace;     // td_debug_wg_trace: (start, end, first wave's end, sum of the waves' ends) per workgroup, or nullptr
};

// td_debug_wg_trace: a wave that has run out of rows leaves its end time (the earliest of the workgroup in slot 2, their sum in slot 3;
// both slots zeroed by the host before the step)
__device__ __forceinline__ void td_trace_wave_end(unsigned long long *trace, int lane) {
    if (lane == 0) {
        const unsigned long long t = __builtin_amdgcn_s_memrealtime();
        atomicMin(trace + 8 * blockIdx.x + 2, t);
        atomicAdd(trace + 8 * blockIdx.x + 3, t);
    }
}

// contiguous share of `count` rows for block b of a (sub-)grid of G blocks
__device__ __forceinline__ void td_node_range16(int64_t count, const int32_t *count_ptr, int64_t &begin, int64_t &end, int G = gridDim.x,
                                                int b = blockIdx.x) {
    if (count_ptr) count = *count_ptr;
    // XCD x (blocks b = x mod 8) gets a contiguous run of chunks: G / 8 of them, one more for the first G mod 8 XCDs
    const int x = b & 7, r = G & 7;
    const int chunk = x * (G >> 3) + (x < r ? x : r) + (b >> 3);
    const int64_t per = (count + G - 1) / G;
    begin = (int64_t)chunk * per;
    end = begin + per < count ? begin + per : count;
}

// Rows of a launch "dealt" to its workgroups: XCD x (workgroups b = x mod 8) still owns a contiguous range of the row list (its
// L2 keeps that range's neighbourhood), but inside the range units of U consecutive rows (one per wave) go round-robin to the
// XCD's workgroups instead of one contiguous share each.  A contiguous share is a third of one graph at C2, and shares differ
// in how many of their rows see both source classes (twice the first-layer products): workgroup busy times spread by +-15 %
// and the launch waits for the slowest (profiles/r03_wg_balance_c2.txt); dealt, every workgroup samples the whole range.
// The wave's rows are first + wid + k * stride (k = 0, 1, ..) below end.
__device__ __forceinline__ void td_deal16(int64_t count, int U, int64_t &first, int64_t &end, int64_t &stride, int G = gridDim.x,
                                          int b = blockIdx.x) {
    const int x = b & 7, r = G & 7, w = b >> 3;
    const int c0 = x * (G >> 3) + (x < r ? x : r), nx = (G >> 3) + (x < r ? 1 : 0);
    // the XCD's range in proportion to its workgroups (with ceil(count / G) rows per workgroup the last XCD came 4 % short at C2 and idled
    // while the others finished)
    const int64_t xb = count * c0 / G;
    end = count * (c0 + nx) / G;
    first = xb + (int64_t)w * U;
    stride = (int64_t)nx * U;
}

// sum / max over the 4 lane groups g (lanes lo, lo+16, lo+32, lo+48), result in all four
__device__ __forceinline__ float td_sum_groups(float v) { return td_sum_halves(td_sum_rows16(v)); }

// reductions over the 16 lanes of a DPP row (fixed g), result in every lane of the row
__device__ __forceinline__ float td_sum16(float v) {
    v += td_dpp<DPP_QUAD_XOR1>(v);
    v += td_dpp<DPP_QUAD_XOR2>(v);
    v += td_dpp<DPP_ROW_HALF_MIRROR>(v);
    v += td_dpp<DPP_ROW_MIRROR>(v);
    return v;
}
__device__ __forceinline__ float td_max16(float v) {
    v = fmaxf(v, td_dpp<DPP_QUAD_XOR1>(v));
    v = fmaxf(v, td_dpp<DPP_QUAD_XOR2>(v));
    v = fmaxf(v, td_dpp<DPP_ROW_HALF_MIRROR>(v));
    v = fmaxf(v, td_dpp<DPP_ROW_MIRROR>(v));
    return v;
}

// Four row-of-16 reductions at once (the four heads 4g .. 4g+3 a lane group owns in the softmax).  Every step is ONE v_max / v_add with
// the DPP modifier on its first source: through __builtin_amdgcn_update_dpp + fmaxf the compiler emits, per step, v_mov_b32 0 /
// v_mov_b32_dpp / a canonicalising v_max / v_max (and SLP-packs the sums into v_pk_add_f32 behind two v_mov_b32_dpp) -- ~230 VALU
// instructions for one row's softmax where ~100 do.  The four independent chains cover each other's "VALU write -> DPP read" hazard
// (2 wait states); the leading s_nop covers the producers of the inputs.
#define TD_DPP4(op, ctrl)                                            \
    op " %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf\n\t"         \
    op " %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf\n\t"         \
    op " %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf\n\t"         \
    op " %3, %3, %3 " ctrl " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void td_max16x4(float (&v)[4]) {
    asm volatile("s_nop 1\n\t" TD_DPP4("v_max_f32_dpp", "quad_perm:[1,0,3,2]") TD_DPP4("v_max_f32_dpp", "quad_perm:[2,3,0,1]")
                 TD_DPP4("v_max_f32_dpp", "row_half_mirror") TD_DPP4("v_max_f32_dpp", "row_mirror")
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
}
__device__ __forceinline__ void td_sum16x4(float (&v)[4]) {
    asm volatile("s_nop 1\n\t" TD_DPP4("v_add_f32_dpp", "quad_perm:[1,0,3,2]") TD_DPP4("v_add_f32_dpp", "quad_perm:[2,3,0,1]")
                 TD_DPP4("v_add_f32_dpp", "row_half_mirror") TD_DPP4("v_add_f32_dpp", "row_mirror")
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
}
#undef TD_DPP4

// softmax over the 32 edges of a row for the four heads 4g + r of the lane group (lg[eb][r] = logit of edge 16eb + lo), times the edge
// gate: p[eb][r] = exp(x - max) / sum * ew[eb].  A pad's logit is -inf, so its weight is exp(-inf) = 0 without a select; a row without
// edges gets zeros.  1 / sum is v_rcp_f32 (1 ulp; the correctly rounded __frcp_rn is an 11-instruction sequence per head).
// (scale: 1 / sqrt(8), times whatever power of two the caller's logits carry)
__device__ __forceinline__ void td_softmax16x4(const floatx4_t (&lg)[2], const bool (&valid)[2], const float (&ew)[2], floatx4_t (&p)[2],
                                               const float scale = TD_ATT_SCALE_16) {
    float x0[4], x1[4], mx[4], sm[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        x0[r] = valid[0] ? lg[0][r] * scale : -INFINITY;
        x1[r] = valid[1] ? lg[1][r] * scale : -INFINITY;
        mx[r] = fmaxf(x0[r], x1[r]);
    }
    td_max16x4(mx);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (mx[r] == -INFINITY) mx[r] = 0.f;
        x0[r] = __expf(x0[r] - mx[r]);
        x1[r] = __expf(x1[r] - mx[r]);
        sm[r] = x0[r] + x1[r];
    }
    td_sum16x4(sm);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float inv = sm[r] > 0.f ? __builtin_amdgcn_rcpf(sm[r]) : 0.f;
        p[0][r] = x0[r] * inv * ew[0];
        p[1][r] = x1[r] * inv * ew[1];
    }
}

// ---- first layer + LayerNorm + ReLU of one dst node: z^T in acc[eb][hb] ------------------------------------------------
struct Edge2 {          // the two edges (lo and 16 + lo) a lane looks at
    bool valid[2];
    bool any[2];       // wave-uniform: does the 16-edge block hold any edge at all (false: all pads, e.g. slots 48 .. 63 at k = 48)
    float ew[2];
    float rel[2][3];   // x_i - x_j
    float4 xi;
};

// Raw operands of one dst row, as loaded: the gathered positions / projections of its 32 neighbours (two per lane).  Kept
// separate from the arithmetic so that a kernel can issue the next row's gathers while it finishes the current row.
struct RowIn16 {
    float4 xi, xj[2];
    int j[2];
    float ew[2];
    float pit[8];          // dst-side projection P_i: hidden 16hb + lo
};

// neighbour indices of chunk c of dst node i (the only load the gathers depend on); c == i on the default k = 32 graph
__device__ __forceinline__ void td_row_index16(const Args16 &a, int64_t i, int64_t c, int lane, RowIn16 &r) {
    const int lo = lane & 15;
    r.xi = a.x4[i];
    r.j[0] = a.nbr[c * TD_K + lo];
    r.j[1] = a.nbr[c * TD_K + 16 + lo];
}

// gathers: neighbour positions, neighbour-side projections straight into the accumulators, dst-side projection
template <bool LOAD_EW>
__device__ __forceinline__ void td_row_gather16(const Args16 &a, int64_t i, int64_t c, int lane, RowIn16 &r,
                                                floatx4_t (&acc)[2][8]) {
    const int lo = lane & 15, g = lane >> 4;
#pragma unroll
    for (int eb = 0; eb < 2; ++eb) {
        const int jj = r.j[eb] >= 0 ? r.j[eb] : (int)i;
        r.xj[eb] = a.x4[jj];
        if (LOAD_EW) r.ew[eb] = a.ew[c * TD_K + 16 * eb + lo];
        // neighbour-side projection P_j, 16 bytes per hidden block: hidden 16hb + 4g .. + 3
        const float *pj = a.P + (size_t)jj * (4 * TD_H) + a.p_off + TD_H + 4 * g;
#pragma unroll
        for (int hb = 0; hb < 8; ++hb) {
            const float4 v = TD_ABL == 5 ? make_float4(0.1f, 0.2f, -0.1f, 0.3f) : *reinterpret_cast<const float4 *>(pj + 16 * hb);
            acc[eb][hb][0] = v.x; acc[eb][hb][1] = v.y; acc[eb][hb][2] = v.z; acc[eb][hb][3] = v.w;
        }
    }
#pragma unroll
    for (int hb = 0; hb < 8; ++hb) r.pit[hb] = a.P[(size_t)i * (4 * TD_H) + a.p_off + 16 * hb + lo];
}

// LayerNorm + ReLU of an edge MLP in the transposed accumulator layout (a lane owns 32 of an edge's 128 hidden units for each of its
// two edges; the other 96 sit in the lanes lo + 16 g'), in the folded form the weights are packed for (FoldedMlp, pack.cpp): the
// accumulators hold the CENTRED pre-activation with the sign of gamma applied, KB[n] = beta_n / (|gamma_n| M), and
//     z''_n = clamp_[0,1](acc_n s + KB[n]),   s = 1 / (sigma M) = rsqrt(sum_n acc_n^2 * ln.c1 + ln.c2)
// is the normalised activation over |gamma_n| M (M bounds it by 1, so the FMA's output clamp IS the ReLU); |gamma_n| M sits in the second
// Linear's columns, nothing is left for the consumer to apply.  Per hidden value: one FMA (variance) and one FMA with the clamp modifier.
// NEB = 1: only the first edge block is live (the caller has zeroed the second one)
struct TdLn { float c1, c2; };
template <int NEB = 2>
__device__ __forceinline__ void td_ln_relu16(const float *__restrict__ KB, int g, floatx4_t (&acc)[2][8], const TdLn ln) {
    float s2[NEB];
#pragma unroll
    for (int eb = 0; eb < NEB; ++eb) {
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int hb = 0; hb < 8; ++hb) {
            sa = fmaf(acc[eb][hb][0], acc[eb][hb][0], sa); sb = fmaf(acc[eb][hb][1], acc[eb][hb][1], sb);
            sa = fmaf(acc[eb][hb][2], acc[eb][hb][2], sa); sb = fmaf(acc[eb][hb][3], acc[eb][hb][3], sb);
        }
        s2[eb] = sa + sb;
    }
    float sc[NEB];
#pragma unroll
    for (int eb = 0; eb < NEB; ++eb) sc[eb] = __frsqrt_rn(fmaf(td_sum_groups(s2[eb]), ln.c1, ln.c2));
#pragma unroll
    for (int hb = 0; hb < 8; ++hb) {
        const float4 kb = *reinterpret_cast<const float4 *>(KB + 16 * hb + 4 * g);
#pragma unroll
        for (int eb = 0; eb < NEB; ++eb) {
            acc[eb][hb][0] = td_clamp01(fmaf(acc[eb][hb][0], sc[eb], kb.x));
            acc[eb][hb][1] = td_clamp01(fmaf(acc[eb][hb][1], sc[eb], kb.y));
            acc[eb][hb][2] = td_clamp01(fmaf(acc[eb][hb][2], sc[eb], kb.z));
            acc[eb][hb][3] = td_clamp01(fmaf(acc[eb][hb][3], sc[eb], kb.w));
        }
    }
}
// general graphs: a block without a single edge (any[eb] false, wave-uniform) gets z = 0 instead of the LayerNorm of its padding
__device__ __forceinline__ void td_ln_relu16_skip(const float *__restrict__ KB, int g, floatx4_t (&acc)[2][8], const TdLn ln,
                                                  const bool (&any)[2]) {
    if (any[1]) {
        td_ln_relu16<2>(KB, g, acc, ln);
    } else {
        td_ln_relu16<1>(KB, g, acc, ln);
#pragma unroll
        for (int hb = 0; hb < 8; ++hb) acc[1][hb] = floatx4_t{0.f, 0.f, 0.f, 0.f};
    }
}

// The same LayerNorm + ReLU with z'' leaving as f16 piece pairs for the value pass's aggregation product (td_split_h2): word r of
// z1[hb] / z2[hb] = pieces of hidden unit 16hb + 4g + r for the lane's two edges (low half: edge lo, high half: edge 16 + lo) -- after the
// flip through the wave's LDS tile a word is two K slots of the A operand.
// SCALED: the pieces are taken of z'' 2^15 (td_split_h2_scaled: four conversions per pair); otherwise of z'' itself (three: a truncation and
// two residuals, 30 % less conversion time) -- which the caller picks for MLPs whose folded scale M is small (TdEdgeMlp::z_plain, pack.cpp):
// z'' is around 1 / M, and only above 2^-6 do the pieces of the plain form keep 19 bits or more over the f16 subnormal floor.
// (NEB = 1, the chunk walk's half-empty chunk: the second block's halves of the words are zero)
template <bool SCALED, int NEB = 2>
__device__ __forceinline__ void td_ln_relu16_pairs_eb(const float *__restrict__ KB, int g, const floatx4_t (&acc)[2][8], const TdLn ln,
                                                      uint4 (&z1)[8], uint4 (&z2)[8]) {
    float sc[2] = {0.f, 0.f};
#pragma unroll
    for (int eb = 0; eb < NEB; ++eb) {
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int hb = 0; hb < 8; ++hb) {
            sa = fmaf(acc[eb][hb][0], acc[eb][hb][0], sa); sb = fmaf(acc[eb][hb][1], acc[eb][hb][1], sb);
            sa = fmaf(acc[eb][hb][2], acc[eb][hb][2], sa); sb = fmaf(acc[eb][hb][3], acc[eb][hb][3], sb);
        }
        sc[eb] = sa + sb;
    }
#pragma unroll
    for (int eb = 0; eb < NEB; ++eb) sc[eb] = __frsqrt_rn(fmaf(td_sum_groups(sc[eb]), ln.c1, ln.c2));
#pragma unroll
    for (int hb = 0; hb < 8; ++hb) {
        const float4 kb4 = *reinterpret_cast<const float4 *>(KB + 16 * hb + 4 * g);
        const float kb[4] = {kb4.x, kb4.y, kb4.z, kb4.w};
        unsigned w1[4], w2[4];
        float za[4], zb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            za[r] = td_clamp01(fmaf(acc[0][hb][r], sc[0], kb[r]));
            zb[r] = NEB == 2 ? td_clamp01(fmaf(acc[1][hb][r], sc[1], kb[r])) : 0.f;
        }
        if constexpr (SCALED) td_split_h2_scaled_x4(za, zb, w1, w2);
        else td_split_h2_x4(za, zb, w1, w2);
        z1[hb] = make_uint4(w1[0], w1[1], w1[2], w1[3]);
        z2[hb] = make_uint4(w2[0], w2[1], w2[2], w2[3]);
    }
}

// ... and for the key pass's logits product, where z''^T is the B operand and the K slots run over hidden units: quad t of edge block eb
// holds the hidden units 16 (2t + j / 4) + 4g + j % 4, j = 0 .. 7 (the two hidden blocks 2t, 2t + 1 of the lane), i.e. the pairs are
// (r = 0, 1) and (r = 2, 3) of one accumulator tile.
// SKIP1 (chunk walk): `any1` = false (wave-uniform) leaves the second block -- all padding -- out: its pieces are zero.
template <bool SCALED, bool SKIP1 = false>
__device__ __forceinline__ void td_ln_relu16_pairs_k(const float *__restrict__ KB, int g, const floatx4_t (&acc)[2][8], const TdLn ln,
                                                     uint4 (&z1)[2][4], uint4 (&z2)[2][4], bool any1 = true) {
    float sc[2];
#pragma unroll
    for (int eb = 0; eb < 2; ++eb) {
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int hb = 0; hb < 8; ++hb) {
            sa = fmaf(acc[eb][hb][0], acc[eb][hb][0], sa); sb = fmaf(acc[eb][hb][1], acc[eb][hb][1], sb);
            sa = fmaf(acc[eb][hb][2], acc[eb][hb][2], sa); sb = fmaf(acc[eb][hb][3], acc[eb][hb][3], sb);
        }
        sc[eb] = sa + sb;
    }
#pragma unroll
    for (int eb = 0; eb < 2; ++eb) sc[eb] = __frsqrt_rn(fmaf(td_sum_groups(sc[eb]), ln.c1, ln.c2));
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float4 kb0 = *reinterpret_cast<const float4 *>(KB + 16 * (2 * t) + 4 * g), kb1 = *reinterpret_cast<const float4 *>(KB + 16 * (2 * t + 1) + 4 * g);
#pragma unroll
        for (int eb = 0; eb < 2; ++eb) {
            if (SKIP1 && eb == 1 && !any1) {
                z1[eb][t] = make_uint4(0u, 0u, 0u, 0u);
                z2[eb][t] = make_uint4(0u, 0u, 0u, 0u);
                continue;
            }
            // pairs (r = 0, 1), (r = 2, 3) of the hidden blocks 2t, 2t + 1: x = the even member, y = the odd one
            const float x[4] = {td_clamp01(fmaf(acc[eb][2 * t][0], sc[eb], kb0.x)), td_clamp01(fmaf(acc[eb][2 * t][2], sc[eb], kb0.z)),
                                td_clamp01(fmaf(acc[eb][2 * t + 1][0], sc[eb], kb1.x)), td_clamp01(fmaf(acc[eb][2 * t + 1][2], sc[eb], kb1.z))};
            const float y[4] = {td_clamp01(fmaf(acc[eb][2 * t][1], sc[eb], kb0.y)), td_clamp01(fmaf(acc[eb][2 * t][3], sc[eb], kb0.w)),
                                td_clamp01(fmaf(acc[eb][2 * t + 1][1], sc[eb], kb1.y)), td_clamp01(fmaf(acc[eb][2 * t + 1][3], sc[eb], kb1.w))};
            unsigned w1[4], w2[4];
            if (TD_ABL == 3) { for (int m = 0; m < 4; ++m) { w1[m] = __float_as_uint(x[m]); w2[m] = __float_as_uint(y[m]); } }
            else if constexpr (SCALED) td_split_h2_scaled_x4(x, y, w1, w2);
            else td_split_h2_x4(x, y, w1, w2);
            z1[eb][t] = make_uint4(w1[0], w1[1], w1[2], w1[3]);
            z2[eb][t] = make_uint4(w2[0], w2[1], w2[2], w2[3]);
        }
    }
}

// radial / type first layer on the gathered operands + LayerNorm + ReLU: z^T in acc[eb][hb]
template <bool LOAD_EW, bool SKIP_EMPTY = false>
__device__ __forceinline__ void td_first_layer_compute16(const Args16 &a, const float4 *__restrict__ Rt,
                                                         const float *__restrict__ KB,
                                                         const float (&offk)[E16_STEPS], const RowIn16 &r, int lane,
                                                         floatx4_t (&acc)[2][8], Edge2 &ed) {
    const int g = lane >> 4;
    const float4 xi = r.xi;
    ed.xi = xi;
    const int cls = xi.w > 0.5f ? 0 : 1;
    float dist[2];
    int slot[2];
    bool has[2][2];        // [source class][edge block]: does any edge of the block have that source class (wave-uniform)
#pragma unroll
    for (int eb = 0; eb < 2; ++eb) {
        ed.valid[eb] = r.j[eb] >= 0;
        const float4 xj = r.xj[eb];
        if (LOAD_EW) ed.ew[eb] = r.ew[eb];
        const float rx = xi.x - xj.x, ry = xi.y - xj.y, rz = xi.z - xj.z;
        dist[eb] = td_sqrt_d2(rx * rx + ry * ry + rz * rz);
        ed.rel[eb][0] = rx; ed.rel[eb][1] = ry; ed.rel[eb][2] = rz;
        slot[eb] = xj.w > 0.5f ? 0 : 1;
        has[0][eb] = __ballot(ed.valid[eb] && slot[eb] == 0) != 0ull;
        has[1][eb] = __ballot(ed.valid[eb] && slot[eb] == 1) != 0ull;
        ed.any[eb] = has[0][eb] || has[1][eb];
    }
    const bool has_a = has[0][0] || has[0][1], has_b = has[1][0] || has[1][1];
    // dst-side projection P_i rides in the table's padding column k = 21 (k-step 5, lane group 1)
    float gv[2][E16_STEPS];
#pragma unroll
    for (int eb = 0; eb < 2; ++eb)
#pragma unroll
        for (int s = 0; s < E16_STEPS; ++s) {
            const int k = 4 * s + g;
            const float u = dist[eb] - offk[s];
            gv[eb][s] = k < TD_NG ? __expf(a.coeff * u * u) : (k <= TD_NG + 1 ? 1.f : 0.f);
        }
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        if (sl == 0 ? !has_a : !has_b) continue;
        const float4 *Rp = Rt + (size_t)((cls * 2 + sl) * E16_STEPS) * 128 + lane * 2;
#pragma unroll
        for (int s = 0; s < E16_STEPS; ++s) {
            const float4 r0 = Rp[s * 128], r1 = Rp[s * 128 + 1];
            float av[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};     // A: R[k = 4s + g][hidden 16hb + lo]
            if (s == 5 && g == 1) {
#pragma unroll
                for (int hb = 0; hb < 8; ++hb) av[hb] = r.pit[hb];
            }
#pragma unroll
            for (int eb = 0; eb < 2; ++eb) {
                if (!has[sl][eb]) continue;        // no edge of this block has this source class: its B columns are all zero
                const float bv = (ed.valid[eb] && slot[eb] == sl) ? gv[eb][s] : 0.f;   // B: g_k(d_edge) for the edge's slot
#pragma unroll
                for (int hb = 0; hb < 8; ++hb) acc[eb][hb] = td_mfma16(av[hb], bv, acc[eb][hb]);
            }
        }
    }
    const TdLn ln{a.mlp.ln_c1, a.mlp.ln_c2};
    if (SKIP_EMPTY) td_ln_relu16_skip(KB, g, acc, ln, ed.any);
    else td_ln_relu16<2>(KB, g, acc, ln);
}

// ---- the same first layer on v_mfma_f32_16x16x32_bf16 -------------------------------------------------------------------
// Both operands as exact bf16 piece triples (the table pre-split at pack time, the per-edge Gaussians split in registers),
// 6 of the 9 piece products, fp32 accumulation: fp32-equivalent (the dropped products are ~2^-24 relative, the size of one fp32
// rounding; the measured errors against the reference goldens are unchanged, profiles/*_split_error_table.txt).  One
// instruction has 32 K slots; the six kept products of the 21 inputs (20 Gaussians + the edge-type column) are K-PACKED into four
// instructions per tile (td_pk4_bquads / td_pk4_tiles below; round 4 spent six, one per product, with 11 of every 32 slots empty).
// P_i joins the accumulator by vector adds (before or after the products).
typedef __bf16 bf16x8_16 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ floatx4_t td_mfma16b(uint4 a, uint4 b, floatx4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_16, a), __builtin_bit_cast(bf16x8_16, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned td_cvt_pk_bf16_e(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// exact split of a pair of fp32 values into three packed bf16 pairs
__device__ __forceinline__ void td_split_pair(float x, float y, unsigned &p1, unsigned &p2, unsigned &p3) {
    p1 = td_cvt_pk_bf16_e(x, y);
    float rx = x - __uint_as_float(p1 << 16), ry = y - __uint_as_float(p1 & 0xffff0000u);
    p2 = td_cvt_pk_bf16_e(rx, ry);
    rx -= __uint_as_float(p2 << 16);
    ry -= __uint_as_float(p2 & 0xffff0000u);
    p3 = td_cvt_pk_bf16_e(rx, ry);
}

// K-packed piece tables (pack_pk4_table, pack.cpp): one (dst class, source class) table is QA[hb 8][lane 64] (16 B), QB (16 B), H7 (8 B),
// QC (16 B) = 28 KiB in global memory; in LDS a kernel keeps QA, QB and either H7 (PK = 2: 20 KiB) or QC (PK = 1: 24 KiB)
constexpr int E16Q_GLOBAL_CS_U4 = 512 + 512 + 256 + 512;
template <int PK> constexpr int e16q_cs_u4() { return PK == 5 ? 512 + 512 : (PK == 1 ? 512 + 512 + 512 : 512 + 512 + 256); }     // (PK = 5: the f16 piece-pair table, pack_h2_table)
template <int PK> constexpr int e16q_u4() { return PK == 3 ? 2 * e16q_cs_u4<2>() + 2 * e16q_cs_u4<1>() : 4 * e16q_cs_u4<PK>(); }   // all four (dst class, source class) tables
template <int PK> constexpr int e16q_half_u4() { return PK == 4 ? e16q_cs_u4<2>() + e16q_cs_u4<1>() : 2 * e16q_cs_u4<PK>(); }      // one destination class
// stage `ncs` consecutive (dst class, source class) tables from the packed blob into LDS (all waves of the workgroup; the caller's
// barrier publishes them)
template <int PK>
__device__ __forceinline__ void td_stage_pk4(const float *__restrict__ src, float *__restrict__ dst, int ncs, int tid, int nthreads) {
    for (int cs = 0; cs < ncs; ++cs) {
        const float4 *sp = reinterpret_cast<const float4 *>(src) + (size_t)cs * E16Q_GLOBAL_CS_U4;
        float4 *dp = reinterpret_cast<float4 *>(dst) + (size_t)cs * e16q_cs_u4<PK>();
        td_stage_lds16(sp, dp, 1024, tid, nthreads);
        if (PK == 1) td_stage_lds16(sp + 1280, dp + 1024, 512, tid, nthreads);
        else td_stage_lds16(sp + 1024, dp + 1024, 256, tid, nthreads);
    }
}

// B operands of the K-packed products for one 16-edge block: the five inputs m[0..4] of the lane's Gaussians (already masked to the
// source class), exactly split; ctype = the type column's constant for this lane (td_pk4_ctype; 0: no such column).
//   t0 = (b1 | b2)[k0..k3]   t1 = (b2 | b1)[k0..k3]   t2 = b1[k0..k3] | (p1 p1 p2 p2)[k4]   t3 = b3[k0..k3] | (p1 p3)[k4], type constant
__device__ __forceinline__ unsigned td_pk4_ctype(int g) { return g == 0 ? 0x3f803f80u : (g == 1 ? 0x00003f80u : 0u); }      // bf16 1.0
__device__ __forceinline__ void td_pk4_bquads(const float (&m)[5], unsigned ctype, uint4 (&bq)[4]) {
    unsigned d1a, d2a, d3a, d1b, d2b, d3b;
    td_split_pair(m[0], m[1], d1a, d2a, d3a);
    td_split_pair(m[2], m[3], d1b, d2b, d3b);
    // k4 alone: both halves of a word hold the same piece
    const unsigned u1 = td_cvt_pk_bf16_e(m[4], m[4]);
    const float r1 = m[4] - __uint_as_float(u1 & 0xffff0000u);
    const unsigned u2 = td_cvt_pk_bf16_e(r1, r1);
    const float r2 = r1 - __uint_as_float(u2 & 0xffff0000u);
    const unsigned u3 = td_cvt_pk_bf16_e(r2, r2);
    const unsigned e13 = __builtin_amdgcn_perm(u3, u1, 0x07060100u);      // low half p1, high half p3
    bq[0] = make_uint4(d1a, d1b, d2a, d2b);
    bq[1] = make_uint4(d2a, d2b, d1a, d1b);
    bq[2] = make_uint4(d1a, d1b, u1, u2);
    bq[3] = make_uint4(d3a, d3b, e13, ctype);
}
// The 4 x 8 x NEB products of one (dst class, source class) table at Rs (LDS; already offset by the lane) into acc.
// PKR = 1: QA, QB, QC resident (three 16-byte reads per hidden block); PKR = 2: QA, QB, H7 (t3's operand is two 8-byte reads).
// Twelve steps = 4 pairs of hidden blocks x {QC (t3), QB (t2), QA (t1, t0)}, low-order instructions first within a pair; a step's two
// A quads feed 4 (8) products on four interleaved accumulator chains.  AH > 0: the quads of step n + AH are read BEFORE step n's
// products are issued (scheduling barrier) -- left to itself (AH = 0) the compiler keeps two quads live and reads each pair right in
// front of its products, twelve exposed LDS round trips per row and source class; that is what the key pass's 168 registers allow.
template <int PKR, int AH, int NEB>
__device__ __forceinline__ void td_pk4_tiles(const uint4 *__restrict__ Rs, int lane, const uint4 (&bq)[NEB][4], floatx4_t (&acc)[2][8]) {
    // PKR = 2: the a1 half of QA is read a second time, as the first half of t3's operand, through a pointer the compiler cannot
    // see through (it would otherwise forward the 16-byte read and assemble the operand with two v_mov per hidden block)
    const uint2 *Ra = reinterpret_cast<const uint2 *>(Rs);
    if constexpr (PKR == 2) asm volatile("" : "+v"(Ra));
    auto quad = [&](int step, int h2) -> uint4 {
        const int hb = 2 * (step / 3) + h2, kind = step % 3;
        if (kind == 2) return Rs[hb * 64];
        if (kind == 1) return Rs[512 + hb * 64];
        if constexpr (PKR == 1) return Rs[1024 + hb * 64];
        else {
            const uint2 lo2 = Ra[hb * 128], hi2 = reinterpret_cast<const uint2 *>(Rs + 1024 - lane)[hb * 64 + lane];
            return make_uint4(lo2.x, lo2.y, hi2.x, hi2.y);
        }
    };
    if constexpr (AH < 0) {
        // one quad ahead (4 more live registers than the compiler's own order): 24 steps of one A quad each, its 2 (4) products on the
        // two edge blocks' accumulator chains
        auto quad1 = [&](int st) -> uint4 { return quad(3 * (st / 6) + (st % 6) / 2, st & 1); };
        uint4 cur = quad1(0);
#pragma unroll
        for (int st = 0; st < 24; ++st) {
            uint4 nxt = cur;
            if (st + 1 < 24) nxt = quad1(st + 1);
            __builtin_amdgcn_sched_barrier(0);
            const int hb = 2 * (st / 6) + (st & 1), kind = (st % 6) / 2;
#pragma unroll
            for (int pass = 0; pass < (kind == 2 ? 2 : 1); ++pass) {
                const int tb = kind == 0 ? 3 : (kind == 1 ? 2 : 1 - pass);
#pragma unroll
                for (int eb = 0; eb < NEB; ++eb) { if (TD_ABL != 4) acc[eb][hb] = td_mfma16b(cur, bq[eb][tb], acc[eb][hb]); else acc[eb][hb][0] += __uint_as_float(cur.x ^ bq[eb][tb].y); }
            }
            cur = nxt;
        }
    } else if constexpr (AH == 0) {
#pragma unroll
        for (int hp = 0; hp < 4; ++hp) {
            uint4 ar[2][3];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int kind = 0; kind < 3; ++kind) ar[h2][kind] = quad(3 * hp + kind, h2);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int kind = t < 2 ? t : 2, tb = 3 - t;
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                    for (int eb = 0; eb < NEB; ++eb)
                        acc[eb][2 * hp + h2] = td_mfma16b(ar[h2][kind], bq[eb][tb], acc[eb][2 * hp + h2]);
            }
        }
    } else {
        uint4 ring[AH + 1][2];
#pragma unroll
        for (int n = 0; n < AH; ++n) { ring[n][0] = quad(n, 0); ring[n][1] = quad(n, 1); }
#pragma unroll
        for (int step = 0; step < 12; ++step) {
            if (step + AH < 12) {
                ring[(step + AH) % (AH + 1)][0] = quad(step + AH, 0);
                ring[(step + AH) % (AH + 1)][1] = quad(step + AH, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            const int hp = step / 3, kind = step % 3;
            const uint4 (&aq)[2] = ring[step % (AH + 1)];
#pragma unroll
            for (int pass = 0; pass < (kind == 2 ? 2 : 1); ++pass) {
                const int tb = kind == 0 ? 3 : (kind == 1 ? 2 : 1 - pass);
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                    for (int eb = 0; eb < NEB; ++eb)
                        acc[eb][2 * hp + h2] = td_mfma16b(aq[h2], bq[eb][tb], acc[eb][2 * hp + h2]);
            }
        }
    }
}

// ---- PK = 5: the first layer on f16 piece PAIRS (pack_h2_table, pack.cpp; the x2h key pass) -- two products per tile instead of four, two
// 16-byte table reads per hidden block instead of three.  B operands of the NEB edge blocks: the five inputs m[eb][0..4] of the lane's
// Gaussians (masked to the source class) as h1 = truncation to f16, h2 = f16 of the exact residual; ct[eb]: f16 1.0 in the high half (the type
// column's input) or 0.
//   I0 x (b1_0 b1_1 | b1_2 b1_3 | b1_0 b1_1 | b1_2 b1_3)      I1 x (b2_0 b2_1 | b2_2 b2_3 | b1_4 b2_4 | b1_4 1)
template <int NEB>
__device__ __forceinline__ void td_h2_bquads(const float (&m)[NEB][5], const unsigned (&ct)[NEB], uint4 (&bq)[NEB][2]) {
    if constexpr (NEB == 2) {
        const float x[4] = {m[0][0], m[0][2], m[1][0], m[1][2]}, y[4] = {m[0][1], m[0][3], m[1][1], m[1][3]};
        unsigned a1[4], a2[4], u1, u2;
        td_split_h2_x4(x, y, a1, a2);
        td_split_h2(m[0][4], m[1][4], u1, u2);          // k4 of the two blocks share a word
        bq[0][0] = make_uint4(a1[0], a1[1], a1[0], a1[1]);
        bq[1][0] = make_uint4(a1[2], a1[3], a1[2], a1[3]);
        bq[0][1] = make_uint4(a2[0], a2[1], __builtin_amdgcn_perm(u2, u1, 0x05040100u), __builtin_amdgcn_perm(ct[0], u1, 0x07060100u));
        bq[1][1] = make_uint4(a2[2], a2[3], __builtin_amdgcn_perm(u2, u1, 0x07060302u), __builtin_amdgcn_perm(ct[1], u1, 0x07060302u));
    } else {
        unsigned a1[2], a2[2], u1, u2;
        td_split_h2(m[0][0], m[0][1], a1[0], a2[0]);
        td_split_h2(m[0][2], m[0][3], a1[1], a2[1]);
        td_split_h2(m[0][4], 0.f, u1, u2);
        bq[0][0] = make_uint4(a1[0], a1[1], a1[0], a1[1]);
        bq[0][1] = make_uint4(a2[0], a2[1], __builtin_amdgcn_perm(u2, u1, 0x05040100u), __builtin_amdgcn_perm(ct[0], u1, 0x07060100u));
    }
}
// the 2 x 8 x NEB products of one (dst class, source class) table at Rs (LDS, offset by the lane): I0[hb][lane] at Rs[hb * 64], I1 at
// Rs[512 + hb * 64].  AHEAD: the quad of step n + 1 is read before step n's products are issued (as td_pk4_tiles<.., AH < 0>).
template <bool AHEAD, int NEB>
__device__ __forceinline__ void td_h2_tiles(const uint4 *__restrict__ Rs, const uint4 (&bq)[NEB][2], floatx4_t (&acc)[2][8]) {
    auto quad = [&](int st) -> uint4 { return Rs[((st & 1) ? 0 : 512) + (st >> 1) * 64]; };      // I1 (the small products) first
    if constexpr (AHEAD) {
        uint4 cur = quad(0);
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            uint4 nxt = cur;
            if (st + 1 < 16) nxt = quad(st + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int eb = 0; eb < NEB; ++eb) acc[eb][st >> 1] = td_mfma16h(cur, bq[eb][(st & 1) ? 0 : 1], acc[eb][st >> 1]);
            cur = nxt;
        }
    } else {
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const uint4 q = quad(st);
#pragma unroll
            for (int eb = 0; eb < NEB; ++eb) acc[eb][st >> 1] = td_mfma16h(q, bq[eb][(st & 1) ? 0 : 1], acc[eb][st >> 1]);
        }
    }
}

// Rp: the piece table in LDS -- all of it, or (ONE_CLASS) the half of the one destination class the workgroup serves.
// PK = 1 / 2: the K-packed form -- FOUR products per (hidden block, edge block) and source class instead of six: the 21 inputs' six piece
// products are 123 (piece, piece, k) slot pairs and fit 4 x 32 K slots (pack_pk4_table, pack.cpp); lane group g owns the Gaussians
// 5g .. 5g+4 (offj[0..4]; ten exponentials per lane instead of sixteen) and splits five values per edge instead of eight.  PK = 1: the
// table in LDS holds QA, QB, QC (three 16-byte reads per hidden block); PK = 2: QA, QB, H7 (40 bytes per lane and hidden block; the
// fourth product's operand is two 8-byte reads, QA's first half and H7).
struct TdNoHook { __device__ __forceinline__ void operator()() const {} };
// before_products: called once the gathered operands have been consumed (Gaussians computed, P_i added) and before the matrix products
// are issued -- the place to start loads whose data is needed after the first layer (the key pass's query): the live set is at its
// lowest there and the products give them ~ 1.5 k cycles of cover.
// NEB = 1 (general graphs): the chunk's second 16-edge block is all padding (blocks fill from slot 0; e.g. slots 48 .. 63 of a row at
// k = 48; the caller tests it, wave-uniform) and costs nothing: no P_i adds, no Gaussians, no products, z = 0 and 1 / sigma = 0.
// LN_SKIP (NEB = 2 only): the LayerNorm leaves out a second block without edges (z = 0, 1 / sigma = 0) -- what the chunk-walking key pass, which
// has no registers to spare for an NEB = 1 path, still saves on a half-empty chunk.
// NO_LN: stop in front of the LayerNorm (the caller runs one of the forms that leave z'' as f16 piece pairs: td_ln_relu16_pairs_*).
template <bool LOAD_EW, bool ONE_CLASS, bool PI_LATE, int NEB, bool LN_SKIP, int PK, int AH = 0, class Hook = TdNoHook, bool NO_LN = false>
__device__ __forceinline__ void td_first_layer_split16(const Args16 &a, const uint4 *__restrict__ Rp,
                                                       const float *__restrict__ KB,
                                                       const float (&offj)[8], const RowIn16 &r, int64_t i, int lane,
                                                       floatx4_t (&acc)[2][8], Edge2 &ed, Hook before_products = Hook()) {
    const int g = lane >> 4;
    const float4 xi = r.xi;
    ed.xi = xi;
    const int cls = xi.w > 0.5f ? 0 : 1;
    // dst-side projection P_i in the accumulator layout (hidden 16hb + 4g + r)
    float4 pi[8];
    {
        const float *pp = a.P + (size_t)i * (4 * TD_H) + a.p_off + 4 * g;
#pragma unroll
        for (int hb = 0; hb < 8; ++hb) pi[hb] = *reinterpret_cast<const float4 *>(pp + 16 * hb);
    }
    auto add_pi = [&]() {
#pragma unroll
        for (int hb = 0; hb < 8; ++hb)
#pragma unroll
            for (int eb = 0; eb < NEB; ++eb) {
                acc[eb][hb][0] += pi[hb].x; acc[eb][hb][1] += pi[hb].y; acc[eb][hb][2] += pi[hb].z; acc[eb][hb][3] += pi[hb].w;
            }
    };
    if (!PI_LATE) add_pi();
    int slot[NEB];
    bool has[2][NEB];
    float gv[NEB][5];
    if (NEB == 1) {
        ed.valid[1] = false; ed.any[1] = false;
        ed.rel[1][0] = ed.rel[1][1] = ed.rel[1][2] = 0.f;
        if (LOAD_EW) ed.ew[1] = 0.f;
#pragma unroll
        for (int hb = 0; hb < 8; ++hb) acc[1][hb] = floatx4_t{0.f, 0.f, 0.f, 0.f};
    }
    // g_k(d) = exp(coeff (d - mu_k)^2) = exp2(c2 (d - mu_k)^2): four instructions per entry (the compiler cannot fold the two scale
    // factors of __expf(coeff * u * u) itself); lane group g owns the Gaussians k = 5g .. 5g + 4 (offj[0..4]).  A pad (neighbour
    // index -1, gathered as the row itself) is NOT zeroed: its column of z is finite garbage that the softmax (logit -inf) and the
    // aggregations (alpha = 0, `valid` selects) never let through.
    const float c2 = a.coeff * TD_LOG2E;
#pragma unroll
    for (int eb = 0; eb < NEB; ++eb) {
        ed.valid[eb] = r.j[eb] >= 0;
        const float4 xj = r.xj[eb];
        if (LOAD_EW) ed.ew[eb] = r.ew[eb];
        const float rx = xi.x - xj.x, ry = xi.y - xj.y, rz = xi.z - xj.z;
        const float dist = td_sqrt_d2(rx * rx + ry * ry + rz * rz);
        ed.rel[eb][0] = rx; ed.rel[eb][1] = ry; ed.rel[eb][2] = rz;
        slot[eb] = xj.w > 0.5f ? 0 : 1;
        has[0][eb] = __ballot(ed.valid[eb] && slot[eb] == 0) != 0ull;
        has[1][eb] = __ballot(ed.valid[eb] && slot[eb] == 1) != 0ull;
        ed.any[eb] = has[0][eb] || has[1][eb];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const float u = dist - offj[j];
            gv[eb][j] = __builtin_amdgcn_exp2f(c2 * (u * u));
        }
    }
    before_products();
    // products of source class sl with the edge inputs (B operand: zeros for the edges of the other class)
    auto products4 = [&](int sl) {
        if constexpr (PK == 5) {
            float mm[NEB][5];
            unsigned ct[NEB];
            uint4 bh[NEB][2];
#pragma unroll
            for (int eb = 0; eb < NEB; ++eb) {
                const bool keep = !has[1 - sl][eb] || slot[eb] == sl;       // edges of the other class contribute nothing
#pragma unroll
                for (int j = 0; j < 5; ++j) mm[eb][j] = keep ? gv[eb][j] : 0.f;
                ct[eb] = keep ? 0x3c000000u : 0u;                           // f16 1.0: the type column's input
            }
            td_h2_bquads<NEB>(mm, ct, bh);
            td_h2_tiles<AH != 0, NEB>(Rp + (size_t)((ONE_CLASS ? 0 : cls * 2) + sl) * e16q_cs_u4<5>() + lane, bh, acc);
            return;
        } else {
        uint4 bq[NEB][4];
        const unsigned ctype = td_pk4_ctype(g);
#pragma unroll
        for (int eb = 0; eb < NEB; ++eb) {
            const bool keep = !has[1 - sl][eb] || slot[eb] == sl;       // edges of the other class contribute nothing
            float m[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) m[j] = keep ? gv[eb][j] : 0.f;
            td_pk4_bquads(m, keep ? ctype : 0u, bq[eb]);
        }
        // PK = 3 (all four tables resident, key pass): the protein-destination tables in the 48-byte form, the ligand-destination ones
        // (one row in 25) in the 40-byte form -- together 88 KiB
        if constexpr (PK == 3) {
            static_assert(PK != 3 || !ONE_CLASS, "PK = 3 holds both destination classes");
            if (cls) td_pk4_tiles<1, AH, NEB>(Rp + 2 * e16q_cs_u4<2>() + (size_t)sl * e16q_cs_u4<1>() + lane, lane, bq, acc);
            else td_pk4_tiles<2, AH, NEB>(Rp + (size_t)sl * e16q_cs_u4<2>() + lane, lane, bq, acc);
        } else if constexpr (PK == 4) {
            // one destination class resident (12-wave value pass): its ligand-source table (mixed rows only) in the 40-byte form, its
            // protein-source table in the 48-byte form -- 44 KiB
            static_assert(PK != 4 || ONE_CLASS, "PK = 4 holds one destination class");
            if (sl) td_pk4_tiles<1, AH, NEB>(Rp + e16q_cs_u4<2>() + lane, lane, bq, acc);
            else td_pk4_tiles<2, AH, NEB>(Rp + lane, lane, bq, acc);
        } else
            td_pk4_tiles<PK, AH, NEB>(Rp + (size_t)((ONE_CLASS ? 0 : cls * 2) + sl) * e16q_cs_u4<PK>() + lane, lane, bq, acc);
        }
    };
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        bool any_sl = has[sl][0];
        if (NEB == 2) any_sl = any_sl || has[sl][NEB - 1];
        if (any_sl) products4(sl);
    }
    if (PI_LATE) add_pi();
    if constexpr (NO_LN) return;
    const TdLn ln{a.mlp.ln_c1, a.mlp.ln_c2};
    if (NEB == 2 && LN_SKIP) td_ln_relu16_skip(KB, g, acc, ln, ed.any);
    else td_ln_relu16<NEB>(KB, g, acc, ln);
}

template <bool LOAD_EW, bool SKIP_EMPTY = false>
__device__ __forceinline__ void td_first_layer16(const Args16 &a, const float4 *__restrict__ Rt,
                                                 const float *__restrict__ KB,
                                                 const float (&offk)[E16_STEPS], int64_t i, int lane,
                                                 floatx4_t (&acc)[2][8], Edge2 &ed, int64_t c = -1) {
    RowIn16 r;
    if (c < 0) c = i;
    td_row_index16(a, i, c, lane, r);
    td_row_gather16<LOAD_EW>(a, i, c, lane, r, acc);
    td_first_layer_compute16<LOAD_EW, SKIP_EMPTY>(a, Rt, KB, offk, r, lane, acc, ed);
}

// ================================================================================================ key pass
constexpr int K16_WAVES = 16;      // key pass: 128 VGPRs -> 4 waves per SIMD, one LDS copy of the weights per CU
constexpr int XV16_WAVES = 8;      // h2x value pass keeps the edge vectors live: 2 waves per SIMD, no spills
constexpr size_t K16_LDS_BYTES = (size_t)(E16_R_FLOATS + E16_WQ_FLOATS + TD_H + 4) * sizeof(float);       // + the row counter
#ifndef TD_KEY_WAVES
#define TD_KEY_WAVES 12
#endif
#ifndef TD_KEY_WALK_F16
#define TD_KEY_WALK_F16 1          // f16 logits in the chunk-walking key pass, where its first layer runs on f16 pairs too (td_launch_edge_key16)
#endif
constexpr int K16S_WAVES = TD_KEY_WAVES;     // 16-bit first layer: <= 168 VGPRs -> 3 waves per SIMD, the piece tables (88 KiB bf16 triples / 64 KiB f16 pairs) + Wq in LDS
#ifndef TD_KEY_PK
#define TD_KEY_PK 3
#endif
#ifndef TD_KEY_AH
#define TD_KEY_AH -1       // key pass, default graph: the first layer's table quads read one quad ahead (td_pk4_tiles)
#endif
constexpr size_t K16S_LDS_BYTES = (size_t)(e16q_u4<TD_KEY_PK>() * 4 + E16_WQ_FLOATS + TD_H + 4 + 32) * sizeof(float);
static_assert(K16S_LDS_BYTES <= 160 * 1024, "key pass: LDS");        // + the row counter, the Gaussian centres

// XV = false: key pass (logits -> softmax -> alpha).
// XV = true : h2x value pass.  xv[e][head] = W2xv[head, :] . z_e + b has the shape of the logits product with a static
//             A operand (no U_i build); delta_x_i = mean_heads sum_e alpha[e, head] xv[e, head] (x_i - x_j)
//             (models/uni_transformer.py:121-140), masked update of the ligand row (:205-206).
// STAGE only tags the instantiation (0 = x2h, 1 = h2x) so that profilers report the two stages separately.
// GRAPH = 2 (`hybrid` graphs, x2h): the protein rows of the list -- one chunk each, index cptr[i] -- at the speed of the default graph; the
//             ligand rows (several chunks) are skipped here and walked by a GRAPH = 1 launch over the ligand row list.
// GRAPH = 1 = CHUNKED (general graphs): the in-edges of dst node i are the chunks cptr[i] .. cptr[i+1]-1 of nbr / ew / alpha (32
//             slots each, -1 padded).  One wave still owns one dst node and walks its chunks.  Key pass: the softmax runs
//             over ALL slots of the node (scatter_softmax over an arbitrary segment, models/uni_transformer.py:73,135) --
//             one chunk: in registers as on the default graph; several: the scaled logits go to alpha[c] with a running
//             (max, sum) per head, then every lane re-reads its own entries and writes exp(x - max) / sum * gate.
//             XV: delta_x accumulates over the chunks (scatter_sum, :139).
// SPLIT = true: the first layer on 16-bit pieces (td_first_layer_split16; the whole piece table in LDS) -- exact bf16 triples, or (FL = 1,
// x2h key pass) f16 pairs.
// L2 (x2h key pass of the default graph, bf16 first layer): the logits product -- 0: fp32 (v_mfma_f32_16x16x4_f32), 1: f16 piece pairs
// with z'' scaled by 2^15, 2: f16 piece pairs of z'' itself (the launcher's reading of TdEdgeMlp::l2_f16 / z_plain).
// FL (x2h key pass, bf16-class first layer): 1 = the first layer on f16 piece pairs (PK = 5, TdEdgeMlp::R16h; model option
// "edge_first_layer_f16"), 0 = on the exact bf16 piece triples
template <bool XV, int WAVES, int STAGE, int GRAPH = 0, bool SPLIT = false, int L2 = 0, int FL = 0>
__global__ __launch_bounds__(WAVES * 64) void edge_key16_kernel(Args16 a) {
    constexpr bool ZPLAIN = L2 == 2;
    constexpr bool CHUNKED = GRAPH == 1;       // walks the chunks of a row
    constexpr bool VIA = GRAPH == 2;           // one chunk per row, found through cptr; ligand rows are somebody else's
    // logits on v_mfma_f32_16x16x32_f16 (f16 piece pairs): the x2h key pass on rows of one chunk (the default graph, the protein rows of a
    // `hybrid` graph) and, with the first layer on f16 pairs, the chunk walk (TD_KEY_WALK_F16).  (The unfused h2x key pass keeps the fp32
    // product: the fused and the unfused form of that stage run different kernels on the same rows and are held bit-identical.)
    constexpr bool L2H = L2 != 0;
    static_assert(L2 == 0 || (SPLIT && !XV && STAGE == 0), "f16 logits: x2h key pass, bf16 first layer");
    // first layer on f16 piece pairs (PK = 5, TdEdgeMlp::R16h): the x2h key pass, and the unfused h2x key / value halves whenever the fused
    // h2x kernel takes them too (they are held bit-identical to it)
    static_assert(FL == 0 || SPLIT, "f16 first layer: the 16-bit instantiations");
    constexpr int KPK = FL ? 5 : TD_KEY_PK;
    constexpr int RF = SPLIT ? e16q_u4<KPK>() * 4 : E16_R_FLOATS;       // floats of the radial/type table (SPLIT: K-packed)
    constexpr int NOFF = SPLIT ? 8 : E16_STEPS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const float4 *Rt = reinterpret_cast<const float4 *>(lds);
    const float4 *Wq = reinterpret_cast<const float4 *>(lds + RF);       // [hb][r][jq][lane] x 4 j
    const float *KB = lds + RF + E16_WQ_FLOATS;          // beta / |gamma| of the folded LayerNorm (td_ln_relu16)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lo = lane & 15, g = lane >> 4;
    if (a.trace && threadIdx.x == 0) a.trace[8 * blockIdx.x + 4] = __builtin_amdgcn_s_memrealtime();      // kernel entry: slot 0 - slot 4 = table staging
    {
        if (SPLIT) {
            if constexpr (KPK == 5)
                td_stage_lds16(reinterpret_cast<const float4 *>(a.mlp.R16h), reinterpret_cast<float4 *>(lds), e16q_u4<5>(), tid, WAVES * 64);
            else if constexpr (TD_KEY_PK == 3) {
                td_stage_pk4<2>(a.mlp.R16q, lds, 2, tid, WAVES * 64);
                td_stage_pk4<1>(a.mlp.R16q + (size_t)2 * E16Q_GLOBAL_CS_U4 * 4, lds + 2 * e16q_cs_u4<2>() * 4, 2, tid, WAVES * 64);
            } else
                td_stage_pk4<TD_KEY_PK>(a.mlp.R16q, lds, 4, tid, WAVES * 64);
        }
        else td_stage_lds16(reinterpret_cast<const float4 *>(a.mlp.R16), reinterpret_cast<float4 *>(lds), RF / 4, tid, WAVES * 64);
        const int nw4 = XV ? 8 * 4 * 64 / 4 : E16_WQ_FLOATS / 4;      // XV: W2xv16[hb][r][lane]
        td_stage_lds16(reinterpret_cast<const float4 *>(a.mlp.Walt16), reinterpret_cast<float4 *>(lds + RF), nw4, tid,
                       WAVES * 64);
        if (tid < TD_H) lds[RF + E16_WQ_FLOATS + tid] = a.mlp.beta[tid];
        else if (tid == TD_H) *reinterpret_cast<int *>(lds + RF + E16_WQ_FLOATS + TD_H) = 0;
        else if (SPLIT && tid >= TD_H + 32 && tid < TD_H + 64) {
            const int s8 = tid - (TD_H + 32);          // K-packed products: entry 8g + i = centre of k = 5g + i (i < 5)
            lds[RF + E16_WQ_FLOATS + TD_H + 4 + s8] = (s8 & 7) < 5 ? a.offsets[5 * (s8 >> 3) + (s8 & 7)] : TD_FAR_CENTRE;
        }
    }
    float offk[NOFF];          // Gaussian centres of the lane's K slots: k = 4s + g (fp32 tiles), k = 8g + s (bf16 tiles)
#pragma unroll
    for (int s = 0; s < NOFF; ++s) {
        const int k = SPLIT ? 5 * g + s : 4 * s + g;
        offk[s] = SPLIT ? (s < 5 ? a.offsets[k] : TD_FAR_CENTRE) : (k < TD_NG ? a.offsets[k] : 0.f);
    }
    __syncthreads();
    if (a.trace && tid == 0) a.trace[8 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    int64_t begin, end, stride = WAVES;
    if (a.deal) td_deal16(a.count_ptr ? (int64_t)*a.count_ptr : a.count, a.deal == 2 ? 1 : WAVES, begin, end, stride);
    else td_node_range16(a.count, a.count_ptr, begin, end);
    // a.deal == 2: single rows are dealt to the XCD's workgroups (row counts differ by at most one; with units of one row per wave
    // a workgroup had 15 or 16 units of 12 rows) and handed to the workgroup's waves one at a time through an LDS counter -- the
    // waves that share a SIMD do not progress at the same rate (the first of a workgroup's 12 is done a third of the launch before
    // the last, tools/wg_balance.py), and a SIMD left with one wave no longer hides its gathers
    int *row_ctr = reinterpret_cast<int *>(lds + RF + E16_WQ_FLOATS + TD_H);
    const bool dyn = a.deal == 2;
    auto grab = [&]() -> int {
        int n = 0;
        if (lane == 0) n = __hip_atomic_fetch_add(row_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return n;
    };
    auto row_of = [&](int n) -> int64_t {       // n-th row of the workgroup
        return begin + (int64_t)__builtin_amdgcn_readfirstlane(n) * stride;
    };

    float4 qpre0, qpre1;           // the row's query, fetched from inside the first layer (bf16 path, non-XV)
    // first layer of chunk c of dst node i: z^T in acc
    auto first_layer = [&](int64_t i, int64_t c, floatx4_t (&acc)[2][8], Edge2 &ed) {
        constexpr bool EW = !XV && !CHUNKED;       // the chunked key pass fetches the gate in its second sweep
        if constexpr (SPLIT) {
            RowIn16 rin;
            td_row_index16(a, i, c, lane, rin);
            td_row_gather16<EW>(a, i, c, lane, rin, acc);
            // the lane group's eight Gaussian centres come from LDS for every row: as loop-invariant registers they were what the
            // allocator pushed to scratch at this kernel's 168-register budget, and a scratch reload is a vmcnt wait in the middle
            // of the row's gathers
            float offr[8];
            {
                int dep = 0;
                asm volatile("" : "+v"(dep));
                const float4 *op = reinterpret_cast<const float4 *>(lds + RF + E16_WQ_FLOATS + TD_H + 4 + 8 * g + dep);
                const float4 o0 = op[0], o1 = op[1];
                offr[0] = o0.x; offr[1] = o0.y; offr[2] = o0.z; offr[3] = o0.w; offr[4] = o1.x; offr[5] = o1.y; offr[6] = o1.z; offr[7] = o1.w;
            }
            auto fetch_q = [&]() {
                if constexpr (!XV && !CHUNKED) {        // (the chunk-walking kernel has no register to spare for it: measured slower)
                    qpre0 = *reinterpret_cast<const float4 *>(a.q + (size_t)i * TD_H + 8 * lo);
                    qpre1 = *reinterpret_cast<const float4 *>(a.q + (size_t)i * TD_H + 8 * lo + 4);
                }
            };
            // a chunk whose second block is all padding (wave-uniform): the xv pass (8 waves, 256 registers) has room for a path without
            // it; the key pass at 168 registers does not (126 spilled registers) and only leaves the block out of its LayerNorm and logits
            if (XV && CHUNKED && __ballot(rin.j[1] >= 0) == 0ull)
                td_first_layer_split16<EW, false, false, 1, false, KPK>(a, reinterpret_cast<const uint4 *>(lds), KB, offr, rin, i, lane, acc, ed, fetch_q);
            else
                td_first_layer_split16<EW, false, false, 2, CHUNKED, KPK, (CHUNKED || XV) ? 0 : TD_KEY_AH, decltype(fetch_q), L2H>(a, reinterpret_cast<const uint4 *>(lds), KB, offr, rin, i, lane, acc, ed, fetch_q);
        } else
            td_first_layer16<EW, CHUNKED>(a, Rt, KB, offk, i, lane, acc, ed, c);
    };

    for (int64_t it = dyn ? row_of(grab()) : begin + wid; it < end; it = dyn ? row_of(grab()) : it + stride) {
        const int64_t i = a.rows ? (int64_t)a.rows[it] : it;           // dst node
        int64_t c0 = i;
        int nch = 1;
        if (CHUNKED) {
            c0 = __builtin_amdgcn_readfirstlane(a.cptr[i]);
            nch = __builtin_amdgcn_readfirstlane(a.cptr[i + 1]) - (int)c0;
        }
        if (VIA) {
            if (__builtin_amdgcn_readfirstlane(__float_as_int(a.x4[i].w)) > __float_as_int(0.5f)) continue;
            c0 = __builtin_amdgcn_readfirstlane(a.cptr[i]);
        }

        if (XV) {
            const float *Wx = lds + RF;                 // [hb][r][lane]: W2xv[head lo][16hb + 4g + r]
            const float b2 = a.mlp.b2[lo];
            float sx = 0.f, sy = 0.f, sz = 0.f;
            float4 xi_keep = a.x4[i];
            for (int64_t c = c0; c < c0 + nch; ++c) {
                floatx4_t acc[2][8];
                Edge2 ed;
                first_layer(i, c, acc, ed);
                floatx4_t xv[2];
#pragma unroll
                for (int eb = 0; eb < 2; ++eb) xv[eb] = floatx4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int hb = 0; hb < 8; ++hb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float u = Wx[(hb * 4 + r) * 64 + lane];
                        xv[0] = td_mfma16(u, acc[0][hb][r], xv[0]);
                        if (!CHUNKED || ed.any[1]) xv[1] = td_mfma16(u, acc[1][hb][r], xv[1]);     // blocks fill from slot 0: only the second can be empty
                    }
                // xv[eb][r] = xv of edge 16eb + lo, head 4g + r (bias: b2 of that head, fetched through the head's lane)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float bias = __shfl(b2, 4 * g + r);
                    const float *ap = a.alpha + ((size_t)c * TD_HEADS + 4 * g + r) * TD_K + lo;
#pragma unroll
                    for (int eb = 0; eb < 2; ++eb) {
                        const float wgt = ed.valid[eb] ? ap[16 * eb] * (xv[eb][r] + bias) : 0.f;
                        sx = fmaf(wgt, ed.rel[eb][0], sx);
                        sy = fmaf(wgt, ed.rel[eb][1], sy);
                        sz = fmaf(wgt, ed.rel[eb][2], sz);
                    }
                }
            }
            sx = td_sum64(sx) * (1.0f / TD_HEADS);
            sy = td_sum64(sy) * (1.0f / TD_HEADS);
            sz = td_sum64(sz) * (1.0f / TD_HEADS);
            if (lane == 0) a.x4_out[i] = make_float4(xi_keep.x + sx, xi_keep.y + sy, xi_keep.z + sz, xi_keep.w);
            continue;
        }

        // ---- logits^T[head][edge] = sum_k U_i[k][head] z[k][edge];  A = U_i built from q_i: lane (head lo, group g) ----
        auto logits = [&](const floatx4_t (&acc)[2][8], floatx4_t (&lg)[2], const Edge2 &ed) {
            // bf16 path: fetched by the first layer in front of its products (td_first_layer_split16's hook); fp32 path: here
            const float4 q0 = (SPLIT && !CHUNKED) ? qpre0 : *reinterpret_cast<const float4 *>(a.q + (size_t)i * TD_H + 8 * lo);
            const float4 q1 = (SPLIT && !CHUNKED) ? qpre1 : *reinterpret_cast<const float4 *>(a.q + (size_t)i * TD_H + 8 * lo + 4);
#pragma unroll
            for (int eb = 0; eb < 2; ++eb) lg[eb] = floatx4_t{0.f, 0.f, 0.f, 0.f};
            // the weight fragments of k-step s + 1 are read while the eight products of k-step s run (left to itself the compiler
            // reads them right in front of their use and the wave sits out the LDS latency 32 times per row)
            float4 w0 = Wq[lane], w1 = Wq[64 + lane];
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) {
                float4 n0 = w0, n1 = w1;
                if (kk + 1 < 32) {
                    n0 = Wq[((kk + 1) * 2 + 0) * 64 + lane];
                    n1 = Wq[((kk + 1) * 2 + 1) * 64 + lane];
                }
                __builtin_amdgcn_sched_barrier(0);         // the reads stay in front of this k-step's arithmetic
                float u = w0.x * q0.x;
                u = fmaf(w0.y, q0.y, u); u = fmaf(w0.z, q0.z, u); u = fmaf(w0.w, q0.w, u);
                u = fmaf(w1.x, q1.x, u); u = fmaf(w1.y, q1.y, u); u = fmaf(w1.z, q1.z, u); u = fmaf(w1.w, q1.w, u);
                lg[0] = td_mfma16(u, acc[0][kk >> 2][kk & 3], lg[0]);
                if (!CHUNKED || ed.any[1]) lg[1] = td_mfma16(u, acc[1][kk >> 2][kk & 3], lg[1]);       // an all-pad second block is skipped
                w0 = n0; w1 = n1;
            }
        };

        // ---- logits on f16 piece pairs: z''^T (B operand, K = hidden units) leaves the LayerNorm as pairs scaled by 2^15; U_i (A operand) is
        // built eight K slots at a time from a query scaled by a power of two so that |U_i| < 2^13 (|U| <= 8 max |W2k'| max |q|), split, and its
        // six products (two edge blocks x {h2 h1', h1 h2', h1 h1'}) are issued between the next eight slots' FMAs.  Returns the factor that
        // takes the two scales off (and applies the attention scale).
        auto logits_h = [&](const floatx4_t (&acc)[2][8], floatx4_t (&lg)[2], const Edge2 &ed) -> float {
            uint4 zk1[2][4], zk2[2][4];
            td_ln_relu16_pairs_k<!ZPLAIN, CHUNKED>(KB, g, acc, TdLn{a.mlp.ln_c1, a.mlp.ln_c2}, zk1, zk2, ed.any[1]);
            float4 q0, q1;
            if constexpr (CHUNKED) {
                q0 = *reinterpret_cast<const float4 *>(a.q + (size_t)i * TD_H + 8 * lo);
                q1 = *reinterpret_cast<const float4 *>(a.q + (size_t)i * TD_H + 8 * lo + 4);
            } else { q0 = qpre0; q1 = qpre1; }
            float qm = fmaxf(fmaxf(fmaxf(fabsf(q0.x), fabsf(q0.y)), fmaxf(fabsf(q0.z), fabsf(q0.w))),
                             fmaxf(fmaxf(fabsf(q1.x), fabsf(q1.y)), fmaxf(fabsf(q1.z), fabsf(q1.w))));
            qm = td_max16(qm);                                    // over the row's 128 query entries (16 lanes x 8; every lane group holds them all)
            int qe = __builtin_amdgcn_frexp_expf(qm * a.mlp.w2_bound);      // the bound is below 2^qe
            qe = qe < -100 ? -100 : (qe > 100 ? 100 : qe);
            const float qs = __builtin_amdgcn_ldexpf(1.0f, 13 - qe);
            q0.x *= qs; q0.y *= qs; q0.z *= qs; q0.w *= qs; q1.x *= qs; q1.y *= qs; q1.z *= qs; q1.w *= qs;
            floatx4_t lgc[2];
#pragma unroll
            for (int eb = 0; eb < 2; ++eb) { lg[eb] = floatx4_t{0.f, 0.f, 0.f, 0.f}; lgc[eb] = floatx4_t{0.f, 0.f, 0.f, 0.f}; }
            uint4 pu1 = make_uint4(0u, 0u, 0u, 0u), pu2 = pu1;
            auto product = [&](int t, int m) {                  // product m of K block t: the small ones first, on their own accumulators
                const int eb = m & 1, kind = m >> 1;
                if (TD_ABL == 6) { lg[eb][0] += __uint_as_float(pu1.x ^ zk1[eb][t].x ^ pu2.y ^ zk2[eb][t].y); return; }
                if (CHUNKED && eb == 1 && !ed.any[1]) return;          // an all-pad second block is skipped
                if (kind == 0) lgc[eb] = td_mfma16h(pu2, zk1[eb][t], lgc[eb]);
                else if (kind == 1) lgc[eb] = td_mfma16h(pu1, zk2[eb][t], lgc[eb]);
                else lg[eb] = td_mfma16h(pu1, zk1[eb][t], lg[eb]);
            };
            float4 w0 = Wq[lane], w1 = Wq[64 + lane];
            float uu[8];
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) {
                float4 n0 = w0, n1 = w1;
                if (kk + 1 < 32 && TD_ABL != 2) {
                    n0 = Wq[((kk + 1) * 2 + 0) * 64 + lane];
                    n1 = Wq[((kk + 1) * 2 + 1) * 64 + lane];
                }
                __builtin_amdgcn_sched_barrier(0);         // the reads stay in front of this k-step's arithmetic
                float u = w0.x * q0.x;
                if (TD_ABL != 1) {
                u = fmaf(w0.y, q0.y, u); u = fmaf(w0.z, q0.z, u); u = fmaf(w0.w, q0.w, u);
                u = fmaf(w1.x, q1.x, u); u = fmaf(w1.y, q1.y, u); u = fmaf(w1.z, q1.z, u); u = fmaf(w1.w, q1.w, u);
                } else u += w1.w;
                uu[kk & 7] = u;
                if (kk >= 8 && (kk & 7) < 6) product((kk >> 3) - 1, kk & 7);
                if ((kk & 7) == 7) {
                    unsigned a1[4], a2[4];
                    const float ux[4] = {uu[0], uu[2], uu[4], uu[6]}, uy[4] = {uu[1], uu[3], uu[5], uu[7]};
                    td_split_h2_x4(ux, uy, a1, a2);
                    pu1 = make_uint4(a1[0], a1[1], a1[2], a1[3]);
                    pu2 = make_uint4(a2[0], a2[1], a2[2], a2[3]);
                }
                w0 = n0; w1 = n1;
            }
#pragma unroll
            for (int m = 0; m < 6; ++m) product(3, m);
#pragma unroll
            for (int eb = 0; eb < 2; ++eb) lg[eb] += lgc[eb];
            return __builtin_amdgcn_ldexpf(ZPLAIN ? TD_ATT_SCALE_16 : TD_ATT_SCALE_16 / TD_Z_SCALE, qe - 13);
        };

        if constexpr (L2H) {
            if constexpr (!CHUNKED) {          // (the chunk walk takes its two sweeps for a row of one chunk as well: a second copy of the products spills)
            floatx4_t acc[2][8], lg[2];
            Edge2 ed;
            first_layer(i, c0, acc, ed);
            const float sc = logits_h(acc, lg, ed);
            floatx4_t pr[2];
            td_softmax16x4(lg, ed.valid, ed.ew, pr, sc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float *dst = a.alpha + ((size_t)c0 * TD_HEADS + 4 * g + r) * TD_K + lo;
                dst[0] = pr[0][r];
                dst[16] = pr[1][r];
            }
            continue;
            }
        } else if (!CHUNKED || nch == 1) {
            floatx4_t acc[2][8], lg[2];
            Edge2 ed;
            first_layer(i, c0, acc, ed);
            logits(acc, lg, ed);
            if (CHUNKED) {
                ed.ew[0] = a.ew[c0 * TD_K + lo];
                ed.ew[1] = a.ew[c0 * TD_K + 16 + lo];
            }
            // ---- softmax over the 32 edges for heads 4g .. 4g+3 (register r), times the edge gate ---------------------
            floatx4_t pr[2];
            td_softmax16x4(lg, ed.valid, ed.ew, pr);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float *dst = a.alpha + ((size_t)c0 * TD_HEADS + 4 * g + r) * TD_K + lo;
                dst[0] = pr[0][r];
                dst[16] = pr[1][r];
            }
            continue;
        }

        // ---- several chunks: sweep 1 stores the scaled logits and keeps a running (max, sum) per head ---------------------
        float mrun[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, srun[4] = {0.f, 0.f, 0.f, 0.f};
        for (int64_t c = c0; c < c0 + nch; ++c) {
            floatx4_t acc[2][8], lg[2];
            Edge2 ed;
            first_layer(i, c, acc, ed);
            float sc0 = TD_ATT_SCALE_16;
            if constexpr (L2H) sc0 = logits_h(acc, lg, ed);
            else logits(acc, lg, ed);
            const float sc1 = sc0;
            float x0[4], x1[4], mn[4], ps[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                x0[r] = ed.valid[0] ? lg[0][r] * sc0 : -INFINITY;
                x1[r] = ed.valid[1] ? lg[1][r] * sc1 : -INFINITY;
                float *dst = a.alpha + ((size_t)c * TD_HEADS + 4 * g + r) * TD_K + lo;
                dst[0] = x0[r];
                dst[16] = x1[r];
                mn[r] = fmaxf(x0[r], x1[r]);
            }
            td_max16x4(mn);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                mn[r] = fmaxf(mrun[r], mn[r]);
                const float ms = mn[r] == -INFINITY ? 0.f : mn[r];       // nothing but pads so far: exp(-inf - 0) = 0
                ps[r] = __expf(x0[r] - ms) + __expf(x1[r] - ms);
            }
            td_sum16x4(ps);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (mn[r] != -INFINITY) {                 // uniform per row of 16 lanes; exp(-inf - mn) = 0 on the first hit
                    srun[r] = srun[r] * __expf(mrun[r] - mn[r]) + ps[r];
                    mrun[r] = mn[r];
                }
        }
        float inv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            inv[r] = srun[r] > 0.f ? __builtin_amdgcn_rcpf(srun[r]) : 0.f;
            if (mrun[r] == -INFINITY) mrun[r] = 0.f;
        }
        // ---- sweep 2: every lane re-reads the entries it wrote (same thread, same addresses) and normalises them -----------
        for (int64_t c = c0; c < c0 + nch; ++c) {
            const float ew0 = a.ew[c * TD_K + lo], ew1 = a.ew[c * TD_K + 16 + lo];       // 0 on pads
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float *dst = a.alpha + ((size_t)c * TD_HEADS + 4 * g + r) * TD_K + lo;
                const float x0 = dst[0], x1 = dst[16];
                dst[0] = __expf(x0 - mrun[r]) * inv[r] * ew0;
                dst[16] = __expf(x1 - mrun[r]) * inv[r] * ew1;
            }
        }
    }
    if (a.trace) {
        td_trace_wave_end(a.trace, lane);
        __syncthreads();
        if (tid == 0) a.trace[8 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

// ================================================================================================ fused h2x stage
// Key pass + value pass of the h2x stage for the listed ligand rows in one launch: the attention weights never leave
// registers (alpha[eb][r] and xv[eb][r] share the (edge 16eb + lo, head 4g + r) layout).  Every destination is a ligand
// atom, so only the ligand-destination half of each MLP's radial/type table is needed and both MLPs' tables are resident
// together (SPLIT: 36 + 64 + 36 + 8 KiB of bf16 piece tables, W2k copy, xv weights; fp32: 24 + 64 + 24 + 8): one staging
// per workgroup, no barrier in the row loop, and the value half's neighbour projections are gathered while the key half's
// logits and softmax run.  Arithmetic is identical to edge_key16_kernel<false> followed by edge_key16_kernel<true>.
constexpr int H2X16_WAVES = 8;
constexpr int H2X16_WX_FLOATS = 8 * 4 * 64;                                  // W2xv16[hb][r][lane]
template <bool SPLIT> constexpr int h2x16_table_floats() { return SPLIT ? e16q_half_u4<2>() * 4 : E16_R_FLOATS / 2; }
template <bool SPLIT> constexpr size_t h2x16_lds_bytes() {
    return (size_t)(2 * h2x16_table_floats<SPLIT>() + E16_WQ_FLOATS + H2X16_WX_FLOATS + 2 * TD_H) * sizeof(float);
}

struct ArgsH2x {
    Args16 a;              // a.mlp = xk MLP (keys), a.p_off = 0
    TdEdgeMlp mlp_v;       // xv MLP
};

// FL = 1: both halves' first layers on f16 piece pairs (PK = 5, TdEdgeMlp::R16h; model option "edge_first_layer_f16")
template <bool SPLIT, int FL = 0>
__global__ __launch_bounds__(H2X16_WAVES * 64) void edge_h2x16_kernel(ArgsH2x ar) {
    constexpr int WAVES = H2X16_WAVES;
    static_assert(FL == 0 || SPLIT, "f16 first layer: the 16-bit instantiation");
    constexpr int HPK = FL ? 5 : 2;
    constexpr int RH = FL ? e16q_half_u4<5>() * 4 : h2x16_table_floats<SPLIT>();
    constexpr int NOFF = SPLIT ? 8 : E16_STEPS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const Args16 &a = ar.a;
    float *Rk = lds, *WqF = Rk + RH, *Rv = WqF + E16_WQ_FLOATS, *WxF = Rv + RH, *GB = WxF + H2X16_WX_FLOATS;
    const float4 *Wq = reinterpret_cast<const float4 *>(WqF);
    const float *Wx = WxF;
    const float *KBk = GB, *KBv = GB + TD_H;              // beta / |gamma| of the two MLPs' folded LayerNorms
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lo = lane & 15, g = lane >> 4;
    if (a.trace && threadIdx.x == 0) a.trace[8 * blockIdx.x + 4] = __builtin_amdgcn_s_memrealtime();      // kernel entry: slot 0 - slot 4 = table staging
    {
        // destination class 0 (ligand) is the first half of either table
        if constexpr (FL) td_stage_lds16(reinterpret_cast<const float4 *>(a.mlp.R16h), reinterpret_cast<float4 *>(Rk), RH / 4, tid, WAVES * 64);
        else if (SPLIT) td_stage_pk4<2>(a.mlp.R16q, Rk, 2, tid, WAVES * 64);
        else td_stage_lds16(reinterpret_cast<const float4 *>(a.mlp.R16), reinterpret_cast<float4 *>(Rk), RH / 4, tid, WAVES * 64);
        td_stage_lds16(reinterpret_cast<const float4 *>(a.mlp.Walt16), reinterpret_cast<float4 *>(WqF), E16_WQ_FLOATS / 4, tid, WAVES * 64);
        if constexpr (FL) td_stage_lds16(reinterpret_cast<const float4 *>(ar.mlp_v.R16h), reinterpret_cast<float4 *>(Rv), RH / 4, tid, WAVES * 64);
        else if (SPLIT) td_stage_pk4<2>(ar.mlp_v.R16q, Rv, 2, tid, WAVES * 64);
        else td_stage_lds16(reinterpret_cast<const float4 *>(ar.mlp_v.R16), reinterpret_cast<float4 *>(Rv), RH / 4, tid, WAVES * 64);
        td_stage_lds16(reinterpret_cast<const float4 *>(ar.mlp_v.Walt16), reinterpret_cast<float4 *>(WxF), H2X16_WX_FLOATS / 4, tid, WAVES * 64);
        if (tid < TD_H) GB[tid] = a.mlp.beta[tid];
        else if (tid < 2 * TD_H) GB[tid] = ar.mlp_v.beta[tid - TD_H];
    }
    float offk[NOFF];
#pragma unroll
    for (int s = 0; s < NOFF; ++s) {
        const int k = SPLIT ? 5 * g + s : 4 * s + g;         // (SPLIT: the K-packed products' assignment, five per lane group)
        offk[s] = SPLIT ? (s < 5 ? a.offsets[k] : TD_FAR_CENTRE) : (k < TD_NG ? a.offsets[k] : 0.f);
    }
    int64_t begin, end;
    td_node_range16(a.count, a.count_ptr, begin, end);
    Args16 av = a;
    av.p_off = 2 * TD_H;
    av.mlp.ln_c1 = ar.mlp_v.ln_c1; av.mlp.ln_c2 = ar.mlp_v.ln_c2;      // the value half's LayerNorm constants (its tables come through Rv / KBv)
    const float b2 = ar.mlp_v.b2[lo];
    __syncthreads();
    if (a.trace && tid == 0) a.trace[8 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();

    for (int64_t it = begin + wid; it < end; it += WAVES) {
        const int64_t i = a.rows ? (int64_t)a.rows[it] : it;
        // ---- key half: first layer -> z, logits = z . U_i, softmax x gate -> al ---------------------------------------
        RowIn16 rin;
        floatx4_t acc[2][8];
        Edge2 ed;
        td_row_index16(a, i, i, lane, rin);
        td_row_gather16<true>(a, i, i, lane, rin, acc);
        if constexpr (SPLIT) td_first_layer_split16<true, true, false, 2, false, HPK, 1>(a, reinterpret_cast<const uint4 *>(Rk), KBk, offk, rin, i, lane, acc, ed);
        else td_first_layer_compute16<true>(a, reinterpret_cast<const float4 *>(Rk), KBk, offk, rin, lane, acc, ed);
        // the value half's gathers (its own accumulators) fly while the logits and the softmax run.  The query is fetched BEFORE they are
        // issued: vmcnt counts in order, so a load issued after the gathers could only be waited for together with them -- and the
        // logits, which need the query first, would start when the gathers have landed instead of while they fly
        const float4 q0 = *reinterpret_cast<const float4 *>(a.q + (size_t)i * TD_H + 8 * lo);
        const float4 q1 = *reinterpret_cast<const float4 *>(a.q + (size_t)i * TD_H + 8 * lo + 4);
        RowIn16 rv = rin;
        floatx4_t accv[2][8];
        td_row_gather16<false>(av, i, i, lane, rv, accv);
        floatx4_t lg[2];
#pragma unroll
        for (int eb = 0; eb < 2; ++eb) lg[eb] = floatx4_t{0.f, 0.f, 0.f, 0.f};
        {   // weight fragments read a k-step ahead (see edge_key16_kernel)
            float4 w0 = Wq[lane], w1 = Wq[64 + lane];
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) {
                float4 n0 = w0, n1 = w1;
                if (kk + 1 < 32) {
                    n0 = Wq[((kk + 1) * 2 + 0) * 64 + lane];
                    n1 = Wq[((kk + 1) * 2 + 1) * 64 + lane];
                }
                __builtin_amdgcn_sched_barrier(0);
                float u = w0.x * q0.x;
                u = fmaf(w0.y, q0.y, u); u = fmaf(w0.z, q0.z, u); u = fmaf(w0.w, q0.w, u);
                u = fmaf(w1.x, q1.x, u); u = fmaf(w1.y, q1.y, u); u = fmaf(w1.z, q1.z, u); u = fmaf(w1.w, q1.w, u);
                lg[0] = td_mfma16(u, acc[0][kk >> 2][kk & 3], lg[0]);
                lg[1] = td_mfma16(u, acc[1][kk >> 2][kk & 3], lg[1]);
                w0 = n0; w1 = n1;
            }
        }
        floatx4_t al[2];
        td_softmax16x4(lg, ed.valid, ed.ew, al);
        // ---- value half: xv MLP on the same edges, delta x = mean_heads sum_e alpha xv (x_i - x_j) ----------------------
        Edge2 ev;
        if constexpr (SPLIT) td_first_layer_split16<false, true, false, 2, false, HPK, 1>(av, reinterpret_cast<const uint4 *>(Rv), KBv, offk, rv, i, lane, accv, ev);
        else td_first_layer_compute16<false>(av, reinterpret_cast<const float4 *>(Rv), KBv, offk, rv, lane, accv, ev);
        floatx4_t xv[2];
#pragma unroll
        for (int eb = 0; eb < 2; ++eb) xv[eb] = floatx4_t{0.f, 0.f, 0.f, 0.f};
        {
            float u = Wx[lane];
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) {
                float un = u;
                if (kk + 1 < 32) un = Wx[(kk + 1) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
                xv[0] = td_mfma16(u, accv[0][kk >> 2][kk & 3], xv[0]);
                xv[1] = td_mfma16(u, accv[1][kk >> 2][kk & 3], xv[1]);
                u = un;
            }
        }
        float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float bias = __shfl(b2, 4 * g + r);
#pragma unroll
            for (int eb = 0; eb < 2; ++eb) {
                const float wgt = ev.valid[eb] ? al[eb][r] * (xv[eb][r] + bias) : 0.f;
                sx = fmaf(wgt, ev.rel[eb][0], sx);
                sy = fmaf(wgt, ev.rel[eb][1], sy);
                sz = fmaf(wgt, ev.rel[eb][2], sz);
            }
        }
        sx = td_sum64(sx) * (1.0f / TD_HEADS);
        sy = td_sum64(sy) * (1.0f / TD_HEADS);
        sz = td_sum64(sz) * (1.0f / TD_HEADS);
        if (lane == 0) a.x4_out[i] = make_float4(ev.xi.x + sx, ev.xi.y + sy, ev.xi.z + sz, ev.xi.w);
    }
    if (a.trace) {
        td_trace_wave_end(a.trace, lane);
        __syncthreads();
        if (tid == 0) a.trace[8 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

// The fused h2x stage on general graphs (chunk-walking; bf16 first layer only): one wave per ligand row, two sweeps over the row's chunks in
// ONE launch instead of a key launch (logits -> softmax over all chunks -> alpha) and an xv launch.  Sweep 1 stores the scaled logits in
// alpha[c] and keeps a running (max, sum) per head; sweep 2 runs the xv MLP on each chunk and weights it with exp(x - max) / sum * gate
// computed on the fly from the stored logits (same lane layout), accumulating delta x.  Arithmetic as in edge_key16_kernel<false, .., 1> +
// edge_key16_kernel<true, .., 1>; tables as in edge_h2x16_kernel (both MLPs' ligand-destination halves resident).
template <int FL = 0>
__global__ __launch_bounds__(H2X16_WAVES * 64) void edge_h2x16_chunked_kernel(ArgsH2x ar) {
    constexpr int WAVES = H2X16_WAVES;
    constexpr int HPK = FL ? 5 : 2;
    constexpr int RH = FL ? e16q_half_u4<5>() * 4 : h2x16_table_floats<true>();
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const Args16 &a = ar.a;
    float *Rk = lds, *WqF = Rk + RH, *Rv = WqF + E16_WQ_FLOATS, *WxF = Rv + RH, *GB = WxF + H2X16_WX_FLOATS;
    const float4 *Wq = reinterpret_cast<const float4 *>(WqF);
    const float *Wx = WxF;
    const float *KBk = GB, *KBv = GB + TD_H;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lo = lane & 15, g = lane >> 4;
    {
        if constexpr (FL) td_stage_lds16(reinterpret_cast<const float4 *>(a.mlp.R16h), reinterpret_cast<float4 *>(Rk), RH / 4, tid, WAVES * 64);
        else td_stage_pk4<2>(a.mlp.R16q, Rk, 2, tid, WAVES * 64);
        td_stage_lds16(reinterpret_cast<const float4 *>(a.mlp.Walt16), reinterpret_cast<float4 *>(WqF), E16_WQ_FLOATS / 4, tid, WAVES * 64);
        if constexpr (FL) td_stage_lds16(reinterpret_cast<const float4 *>(ar.mlp_v.R16h), reinterpret_cast<float4 *>(Rv), RH / 4, tid, WAVES * 64);
        else td_stage_pk4<2>(ar.mlp_v.R16q, Rv, 2, tid, WAVES * 64);
        td_stage_lds16(reinterpret_cast<const float4 *>(ar.mlp_v.Walt16), reinterpret_cast<float4 *>(WxF), H2X16_WX_FLOATS / 4, tid, WAVES * 64);
        if (tid < TD_H) GB[tid] = a.mlp.beta[tid];
        else if (tid < 2 * TD_H) GB[tid] = ar.mlp_v.beta[tid - TD_H];
    }
    float offk[8];
#pragma unroll
    for (int sx = 0; sx < 8; ++sx) offk[sx] = sx < 5 ? a.offsets[5 * g + sx] : TD_FAR_CENTRE;          // K-packed products: five Gaussians per lane group
    int64_t begin, end;
    td_node_range16(a.count, a.count_ptr, begin, end);
    Args16 av = a;
    av.p_off = 2 * TD_H;
    av.mlp.ln_c1 = ar.mlp_v.ln_c1; av.mlp.ln_c2 = ar.mlp_v.ln_c2;      // the value half's LayerNorm constants (its tables come through Rv / KBv)
    const float b2 = ar.mlp_v.b2[lo];
    __syncthreads();
    for (int64_t it = begin + wid; it < end; it += WAVES) {
        const int64_t i = a.rows ? (int64_t)a.rows[it] : it;
        const int c0 = __builtin_amdgcn_readfirstlane(a.cptr[i]), c1 = __builtin_amdgcn_readfirstlane(a.cptr[i + 1]);
        const float4 q0 = *reinterpret_cast<const float4 *>(a.q + (size_t)i * TD_H + 8 * lo);
        const float4 q1 = *reinterpret_cast<const float4 *>(a.q + (size_t)i * TD_H + 8 * lo + 4);
        // ---- sweep 1: keys -> scaled logits into alpha[c], running (max, sum) per head ---------------------------------------------
        float mrun[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, srun[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = c0; c < c1; ++c) {
            RowIn16 rin;
            floatx4_t acc[2][8];
            Edge2 ed;
            td_row_index16(a, i, c, lane, rin);
            td_row_gather16<false>(a, i, c, lane, rin, acc);
            if (__ballot(rin.j[1] >= 0) == 0ull)              // wave-uniform: the chunk's second block is all padding
                td_first_layer_split16<false, true, false, 1, false, HPK>(a, reinterpret_cast<const uint4 *>(Rk), KBk, offk, rin, i, lane, acc, ed);
            else
                td_first_layer_split16<false, true, false, 2, false, HPK>(a, reinterpret_cast<const uint4 *>(Rk), KBk, offk, rin, i, lane, acc, ed);
            floatx4_t lg[2];
#pragma unroll
            for (int eb = 0; eb < 2; ++eb) lg[eb] = floatx4_t{0.f, 0.f, 0.f, 0.f};
            float4 w0 = Wq[lane], w1 = Wq[64 + lane];
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) {
                float4 n0 = w0, n1 = w1;
                if (kk + 1 < 32) {
                    n0 = Wq[((kk + 1) * 2 + 0) * 64 + lane];
                    n1 = Wq[((kk + 1) * 2 + 1) * 64 + lane];
                }
                __builtin_amdgcn_sched_barrier(0);
                float u = w0.x * q0.x;
                u = fmaf(w0.y, q0.y, u); u = fmaf(w0.z, q0.z, u); u = fmaf(w0.w, q0.w, u);
                u = fmaf(w1.x, q1.x, u); u = fmaf(w1.y, q1.y, u); u = fmaf(w1.z, q1.z, u); u = fmaf(w1.w, q1.w, u);
                lg[0] = td_mfma16(u, acc[0][kk >> 2][kk & 3], lg[0]);
                if (ed.any[1]) lg[1] = td_mfma16(u, acc[1][kk >> 2][kk & 3], lg[1]);
                w0 = n0; w1 = n1;
            }
            float x0[4], x1[4], mn[4], ps[4];
            const float sc0 = TD_ATT_SCALE_16, sc1 = TD_ATT_SCALE_16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                x0[r] = ed.valid[0] ? lg[0][r] * sc0 : -INFINITY;
                x1[r] = ed.valid[1] ? lg[1][r] * sc1 : -INFINITY;
                float *dst = a.alpha + ((size_t)c * TD_HEADS + 4 * g + r) * TD_K + lo;
                dst[0] = x0[r];
                dst[16] = x1[r];
                mn[r] = fmaxf(x0[r], x1[r]);
            }
            td_max16x4(mn);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                mn[r] = fmaxf(mrun[r], mn[r]);
                const float ms = mn[r] == -INFINITY ? 0.f : mn[r];
                ps[r] = __expf(x0[r] - ms) + __expf(x1[r] - ms);
            }
            td_sum16x4(ps);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (mn[r] != -INFINITY) {
                    srun[r] = srun[r] * __expf(mrun[r] - mn[r]) + ps[r];
                    mrun[r] = mn[r];
                }
        }
        float inv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            inv[r] = srun[r] > 0.f ? __builtin_amdgcn_rcpf(srun[r]) : 0.f;
            if (mrun[r] == -INFINITY) mrun[r] = 0.f;
        }
        // ---- sweep 2: xv MLP per chunk, weighted with the normalised, gated attention weights from the stored logits -----------------
        float sx = 0.f, sy = 0.f, sz = 0.f;
        float4 xi_keep = a.x4[i];
        for (int c = c0; c < c1; ++c) {
            RowIn16 rin;
            floatx4_t acc[2][8];
            Edge2 ed;
            td_row_index16(av, i, c, lane, rin);
            td_row_gather16<false>(av, i, c, lane, rin, acc);
            const float ew0 = a.ew[(size_t)c * TD_K + lo], ew1 = a.ew[(size_t)c * TD_K + 16 + lo];       // 0 on pads
            if (__ballot(rin.j[1] >= 0) == 0ull)
                td_first_layer_split16<false, true, false, 1, false, HPK>(av, reinterpret_cast<const uint4 *>(Rv), KBv, offk, rin, i, lane, acc, ed);
            else
                td_first_layer_split16<false, true, false, 2, false, HPK>(av, reinterpret_cast<const uint4 *>(Rv), KBv, offk, rin, i, lane, acc, ed);
            floatx4_t xv[2];
#pragma unroll
            for (int eb = 0; eb < 2; ++eb) xv[eb] = floatx4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int hb = 0; hb < 8; ++hb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float u = Wx[(hb * 4 + r) * 64 + lane];
                    xv[0] = td_mfma16(u, acc[0][hb][r], xv[0]);
                    if (ed.any[1]) xv[1] = td_mfma16(u, acc[1][hb][r], xv[1]);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float bias = __shfl(b2, 4 * g + r);
                const float *lp = a.alpha + ((size_t)c * TD_HEADS + 4 * g + r) * TD_K + lo;
                const float ex[2] = {__expf(lp[0] - mrun[r]) * inv[r] * ew0, __expf(lp[16] - mrun[r]) * inv[r] * ew1};
#pragma unroll
                for (int eb = 0; eb < 2; ++eb) {
                    const float wgt = ed.valid[eb] ? ex[eb] * (xv[eb][r] + bias) : 0.f;
                    sx = fmaf(wgt, ed.rel[eb][0], sx);
                    sy = fmaf(wgt, ed.rel[eb][1], sy);
                    sz = fmaf(wgt, ed.rel[eb][2], sz);
                }
            }
        }
        sx = td_sum64(sx) * (1.0f / TD_HEADS);
        sy = td_sum64(sy) * (1.0f / TD_HEADS);
        sz = td_sum64(sz) * (1.0f / TD_HEADS);
        if (lane == 0) a.x4_out[i] = make_float4(xi_keep.x + sx, xi_keep.y + sy, xi_keep.z + sz, xi_keep.w);
    }
}

// ================================================================================================ value pass (x2h)
constexpr int V16_WAVES = 8;
#ifndef TD_VALUE_PK
#define TD_VALUE_PK 1          // K-packed first layer with QC in LDS (48 bytes per lane and hidden block: the value pass has the room)
#endif
constexpr int V16_W_FLOATS = 32 * TD_H * 4;               // Wt[d][kq][head][4] (pack.cpp, td_value_out16)
constexpr int V16_TB_STRIDE = 20;                         // [32 edges][16 hidden + 4]
// Rows 8g .. 8g + 7 of a tile are read by lane group g (ds_read_b32, hidden column lo): with any 16-byte-aligned row stride 8 rows
// are a multiple of 32 banks, i.e. groups 0 and 1 (and 2, 3) of a half-wave collide two-way (PMC: 10 % conflict cycles).  16 floats of
// padding after every 8 rows put the odd groups on the other 16 banks.
constexpr int V16_TILE_FLOATS = 32 * V16_TB_STRIDE + 4 * 16;
__device__ __forceinline__ int td_tile_row16(int e) { return e * V16_TB_STRIDE + (e >> 3) * 16; }
constexpr int V16_WAVE_FLOATS = 2 * V16_TILE_FLOATS;      // two transpose tiles
constexpr int V16_SB_FLOATS = 48;                         // per wave (8-wave kernel): 16 spare floats + RS[32]: the 32 edges' gates of the GATE_M form (scale_edges) on
                                                          // their way from the z^T lanes to the A operand's lanes (the folded LayerNorm left no 1 / sigma to carry)
// ---- out[n] = W2v[n, :] . Zbar[head(n), :] (n = 8 head + d), the value pass's per-row output product ------------------------------------
// The aggregation product runs with z as the A operand, so the accumulators hold Zbar^T: zt[hb][r] = Zbar[head lo][hidden 16hb + 4g + r].
// A lane then owns 32 of its head's 128 Zbar values and needs no exchange through LDS (the form with Zbar in the accumulator rows wrote it
// out and read every head's row back: 64 ds_write_b32 + 64 ds_read_b128 per row, a quarter of the pass's LDS instructions): eight partial
// dot products, one per d, against Wt[d][kq = 4hb + g][head lo] (float4 over r; byte address = 16 lane + 8192 d + 1024 hb, conflict-free,
// one address register), then the four lane groups' partial sums meet in two swap + add steps that transpose on the way:
// lane (lo, g) ends with the outputs d0 = 2 (g & 1) + (g >> 1) and d0 + 4 of head lo (td_value_out_index16).
// The order of the sums is fixed here for every value kernel: the row distribution settings select different kernels and stay bit-identical.
__device__ __forceinline__ int td_value_out_index16(int lo, int g) { return 8 * lo + 2 * (g & 1) + (g >> 1); }
__device__ __forceinline__ void td_value_out16(const floatx4_t (&zt)[8], const float *Wt_lane, float &o0, float &o1) {
    const float4 *W = reinterpret_cast<const float4 *>(Wt_lane);          // this lane's column: &Wt[0][g][lo]
    float p[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) p[d] = 0.f;
#pragma unroll
    for (int hb = 0; hb < 8; ++hb) {
        float4 w[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) w[d] = W[d * 512 + hb * 64];
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            p[d] = fmaf(w[d].x, zt[hb][0], p[d]); p[d] = fmaf(w[d].y, zt[hb][1], p[d]);
            p[d] = fmaf(w[d].z, zt[hb][2], p[d]); p[d] = fmaf(w[d].w, zt[hb][3], p[d]);
        }
    }
    // lanes 32 .. 63 of p[2m] <-> lanes 0 .. 31 of p[2m + 1]: p[2m] + p[2m + 1] = output 2m + (g >> 1) summed over the groups g, g ^ 2
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\t"
                 "v_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7"
                 : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]));
    float s0 = p[0] + p[1], s1 = p[2] + p[3], s2 = p[4] + p[5], s3 = p[6] + p[7];
    // rows 1, 3 of s[2q] <-> rows 0, 2 of s[2q + 1]: s[2q] + s[2q + 1] = output 4q + 2 (g & 1) + (g >> 1), all four groups in
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
    o0 = s0 + s1;
    o1 = s2 + s3;
}

constexpr size_t V16_LDS_BYTES =
    (size_t)(E16_R_FLOATS + V16_W_FLOATS + V16_WAVES * V16_WAVE_FLOATS + V16_WAVES * V16_SB_FLOATS + 2 * TD_H + 4) * sizeof(float);
constexpr size_t V16S_LDS_BYTES =
    (size_t)(e16q_half_u4<TD_VALUE_PK>() * 4 + V16_W_FLOATS + V16_WAVES * V16_WAVE_FLOATS + V16_WAVES * V16_SB_FLOATS + 2 * TD_H + 4 + 32) * sizeof(float);
static_assert(V16S_LDS_BYTES <= 160 * 1024, "value pass: LDS");

// SPLIT = true: the first layer on bf16 piece triples.  LDS has room for one destination class of the K-packed piece table
// (48 KiB), so the workgroups of a launch specialise: the last GL stage the ligand-destination half and walk the ligand
// rows (a.lig_rows: every ligand atom; the row lists of a forward pass / sampling step always contain them all), 8
// neighbouring ligand rows at a time per workgroup; the others stage the protein half and walk the protein rows of their
// contiguous share of the row list.  GL is chosen inside the kernel from the list's length (device-side in a session) and
// the ligand count so that both kinds finish together: a ligand row costs about 1.25 protein rows (both source classes in
// its first layer).
// A wave looks at 64 candidate rows at a time (lane t reads the class of candidate t) and walks the ones of its class.
// Row costs (relative): a row whose neighbours are of one source class, and a row that sees both (every ligand row; the protein rows
// with a ligand atom among their neighbours).  Measured on the workgroup traces (tools/wg_balance.py --detail): with every row priced
// alike the ligand workgroups took 1.15 x the protein ones on the full-size lists (28 % mixed protein rows) and 1.0 x on the dirty-row
// list of layer 0 and the level-1 list of the last layer (all mixed).
constexpr int TD_ROW_COST_PURE = 100, TD_ROW_COST_MIXED = 122;
// CHUNKED = true (general graphs): a dst node's in-edges are the chunks cptr[i] .. cptr[i+1]-1; alpha (already normalised over
// the whole node and gated by the key pass) is indexed by chunk; Zbar accumulates over the chunks, then one output product.
// GATE_M (ew_net_type 'm', default graph only): the edge gate is e_w = sigmoid(Linear(128 -> 1)(v_e)) of the edge's VALUE vector
// (models/uni_transformer.py:36-37, 62-63).  v_e = W2v z_e + b2v is never formed here; the gate's logit is linear in it, so it is
// (W2v^T w) . z_e + (w . b2v + b) -- one 128-wide dot product with a vector packed at model creation --, and since the gate multiplies v_e,
// which enters the output linearly, it multiplies the attention weight instead: alpha_e e_w_e feeds both the aggregation and S.
// FL: 1 = the first layer on f16 piece pairs (PK = 5, TdEdgeMlp::R16h; see edge_key16_kernel)
template <bool SPLIT, bool CHUNKED = false, bool GATE_M = false, int L2 = 0, int FL = 0>
__global__ __launch_bounds__(V16_WAVES * 64) void edge_value16_kernel(Args16 a) {
    constexpr bool ZPLAIN = L2 == 2;
    static_assert(FL == 0 || SPLIT, "f16 first layer: the bf16-class instantiations");
    constexpr int VPK = FL ? 5 : TD_VALUE_PK;
    constexpr int RF = SPLIT ? e16q_half_u4<VPK>() * 4 : E16_R_FLOATS;
    constexpr int NOFF = SPLIT ? 8 : E16_STEPS;
    // the aggregation product on f16 piece pairs, exactly as in edge_value16t_kernel (the row distribution settings select between the two
    // kernels and stay bit-identical): the bf16-first-layer instantiations, chunk walk included
    constexpr bool L2H = L2 != 0;
    static_assert(L2 == 0 || (SPLIT && !GATE_M), "f16 aggregation: the bf16-first-layer instantiations");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const float4 *Rt = reinterpret_cast<const float4 *>(lds);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;          // lds + RF: Wt[d 8][kq 32][head 16] x 4 k (td_value_out16)
    const int lo = lane & 15, g = lane >> 4;
    const int nout = td_value_out_index16(lo, g);         // this lane's outputs: nout, nout + 4
    // (opaque: as a compile-time constant the table's offset is folded into the immediates of the 64 reads and no longer fits their 16 bits)
    int woff = RF + 4 * lane;
    asm volatile("" : "+v"(woff));
    const float *Wt_lane = lds + woff;
    float *TB = lds + RF + V16_W_FLOATS + wid * V16_WAVE_FLOATS;             // wave-private scratch
    float *SB = lds + RF + V16_W_FLOATS + V16_WAVES * V16_WAVE_FLOATS + wid * V16_SB_FLOATS, *RS = SB + 16;
    float *B2 = lds + RF + V16_W_FLOATS + V16_WAVES * V16_WAVE_FLOATS + V16_WAVES * V16_SB_FLOATS;
    // GATE_M: the attention weights of the aggregation product (A operand: edge 8g + s, head lo) times the edges' gates, which are computed
    // in the lanes of the z^T layout (edge 16eb + lo): 32 floats through the wave's own LDS
    auto scale_edges = [&](float (&alx)[8], float v0, float v1) {          // alx[s] *= v of edge 8g + s (v0 / v1: the lane's two edges)
        if (g < 2) RS[16 * g + lo] = g == 0 ? v0 : v1;
        const float4 r0 = *reinterpret_cast<const float4 *>(RS + 8 * g), r1 = *reinterpret_cast<const float4 *>(RS + 8 * g + 4);
        alx[0] *= r0.x; alx[1] *= r0.y; alx[2] *= r0.z; alx[3] *= r0.w; alx[4] *= r1.x; alx[5] *= r1.y; alx[6] *= r1.z; alx[7] *= r1.w;
    };
    const float *KB = B2 + TD_H;                          // beta / (|gamma| M) of the folded LayerNorm (td_ln_relu16)
    // SPLIT: workgroups [0, GP) serve the protein rows (class 1), [GP, gridDim.x) the ligand rows (class 0)
    int my_cls = 1, GL = 0;
    int64_t n_rows = 0;
    if (SPLIT) {
        n_rows = a.count_ptr ? (int64_t)*a.count_ptr : a.count;
        const int64_t G = gridDim.x, nl = a.lig_count, np = n_rows > nl ? n_rows - nl : 0;
        if (nl > 0) {
            // protein rows with a ligand neighbour (at most those in the list) run the first layer for both source classes, as a
            // ligand row does
            int64_t nm = a.mixed_count ? (int64_t)*a.mixed_count - nl : (np * 3) / 10;
            nm = nm < 0 ? 0 : (nm > np ? np : nm);
            const int64_t cp = CHUNKED ? a.cpn_p : 1;
            const int64_t wl = TD_ROW_COST_MIXED * (CHUNKED ? a.lig_chunks : nl), wp = cp * (TD_ROW_COST_PURE * (np - nm) + TD_ROW_COST_MIXED * nm);
            GL = (int)((wl * G + (wl + wp) / 2) / (wl + wp));
            const int cap = (int)G - (np > 0 ? 1 : 0);
            GL = GL < 1 ? 1 : (GL > cap ? cap : GL);
        }
        my_cls = (int)blockIdx.x >= (int)G - GL ? 0 : 1;
    }
    const int GP = gridDim.x - GL;
    if (a.trace && threadIdx.x == 0) a.trace[8 * blockIdx.x + 4] = __builtin_amdgcn_s_memrealtime();      // kernel entry: slot 0 - slot 4 = table staging
    {
        if (SPLIT) {
            if constexpr (VPK == 5)
                td_stage_lds16(reinterpret_cast<const float4 *>(a.mlp.R16h) + (size_t)my_cls * 2 * e16q_cs_u4<5>(), reinterpret_cast<float4 *>(lds), 2 * e16q_cs_u4<5>(), tid, V16_WAVES * 64);
            else
                td_stage_pk4<TD_VALUE_PK>(a.mlp.R16q + (size_t)my_cls * 2 * E16Q_GLOBAL_CS_U4 * 4, lds, 2, tid, V16_WAVES * 64);
        }
        else td_stage_lds16(reinterpret_cast<const float4 *>(a.mlp.R16), reinterpret_cast<float4 *>(lds), RF / 4, tid, V16_WAVES * 64);
        td_stage_lds16(reinterpret_cast<const float4 *>(a.mlp.Walt), reinterpret_cast<float4 *>(lds + RF), V16_W_FLOATS / 4, tid,
                       V16_WAVES * 64);
        if (tid < TD_H) B2[tid] = a.mlp.b2[tid];
        else if (tid < 2 * TD_H) B2[tid] = a.mlp.beta[tid - TD_H];
        else if (tid == 2 * TD_H) *reinterpret_cast<int *>(B2 + 2 * TD_H) = 0;
        else if (SPLIT && tid >= 2 * TD_H + 32 && tid < 2 * TD_H + 64) {           // Gaussian centres, read per row (see edge_key16_kernel)
            const int s8 = tid - (2 * TD_H + 32);          // K-packed products: entry 8g + i = centre of k = 5g + i (i < 5)
            B2[2 * TD_H + 4 + s8] = (s8 & 7) < 5 ? a.offsets[5 * (s8 >> 3) + (s8 & 7)] : TD_FAR_CENTRE;
        }
    }
    float offk[NOFF];
#pragma unroll
    for (int s = 0; s < NOFF; ++s) {
        const int k = SPLIT ? 5 * g + s : 4 * s + g;         // (SPLIT: the K-packed products' assignment, five per lane group)
        offk[s] = SPLIT ? (s < 5 ? a.offsets[k] : TD_FAR_CENTRE) : (k < TD_NG ? a.offsets[k] : 0.f);
    }
    __syncthreads();
    if (a.trace && tid == 0) a.trace[8 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    auto trace_end = [&]() {
        if (a.trace) {
            td_trace_wave_end(a.trace, lane);
            __syncthreads();
            if (tid == 0) a.trace[8 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        }
    };

    // Software pipeline over the wave's rows: the neighbour indices of row n + 1 are fetched at the top of row n, and its
    // gathers (32 neighbour projections straight into the accumulator registers, which are free once Zbar is done) are
    // issued before row n's output GEMV, so that they land while it runs.
    // The wave's candidates: entries scan + 8 t (t = 0, 1, ..) of `list` (nullptr: the identity), below `end`.
    const int32_t *list = a.rows;
    int64_t scan, end, stride = V16_WAVES;
    if (!SPLIT) {
        if (a.deal) td_deal16(a.count_ptr ? (int64_t)*a.count_ptr : a.count, a.deal == 2 ? 1 : V16_WAVES, scan, end, stride);
        else td_node_range16(a.count, a.count_ptr, scan, end);
    } else if (my_cls) {
        if (a.deal) td_deal16(n_rows, a.deal == 2 ? 1 : V16_WAVES, scan, end, stride, GP, (int)blockIdx.x);
        else td_node_range16(n_rows, nullptr, scan, end, GP, (int)blockIdx.x);
    } else {
        list = a.lig_rows;
        const int64_t per = (a.lig_count + GL - 1) / GL;
        scan = ((int)blockIdx.x - GP) * per;
        end = scan + per < a.lig_count ? scan + per : a.lig_count;
    }
    // a.deal == 2 (not the ligand workgroups, whose 8-row shares are static): single rows dealt to the XCD's workgroups, and the
    // workgroup's rows go to its waves one at a time through an LDS counter (see edge_key16_kernel).  A protein workgroup then meets the ligand rows of its share as candidates:
    // `of_class` below drops them once the candidate's x4 entry -- loaded anyway -- has arrived.
    const bool dyn = a.deal == 2 && (!SPLIT || my_cls == 1);
    const int64_t first = scan;
    if (!dyn) scan += wid;
    int *row_ctr = reinterpret_cast<int *>(B2 + 2 * TD_H);
    auto row_id = [&](int64_t itx) -> int64_t { return list ? (int64_t)list[itx] : itx; };
    int cand = 0;                      // SPLIT: lane t = row id of candidate t of the current window
    unsigned long long todo = 0ull;    // SPLIT: candidates of the window still to do
    auto next_row = [&]() -> int64_t {
        if (dyn) {
            int n = 0;
            if (lane == 0) n = __hip_atomic_fetch_add(row_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            n = __builtin_amdgcn_readfirstlane(n);
            const int64_t itx = first + (int64_t)n * stride;
            return itx < end ? row_id(itx) : -1;
        }
        if (!SPLIT) {
            if (scan >= end) return -1;
            const int64_t r = row_id(scan);
            scan += stride;
            return r;
        }
        while (true) {
            if (todo) {
                const int t = __ffsll(todo) - 1;
                todo &= todo - 1;
                return (int64_t)__builtin_amdgcn_readlane(cand, t);
            }
            if (scan >= end) return -1;
            const int64_t idx = scan + stride * lane;
            scan += stride * 64;
            bool mine = false;
            cand = 0;
            if (idx < end) {
                cand = (int)row_id(idx);
                mine = (a.x4[cand].w > 0.5f ? 0 : 1) == my_cls;
            }
            todo = __ballot(mine);
        }
    };
    auto of_class = [&](const float4 &xi) -> bool {        // wave-uniform: the row's x4 entry is the same in every lane
        return (__builtin_amdgcn_readfirstlane(__float_as_int(xi.w)) > __float_as_int(0.5f) ? 0 : 1) == my_cls;
    };
    // chunk of a single-chunk row (the pipelined loop below): the node itself on the default graph, cptr[i] on a general one
    auto chunk_of = [&](int64_t ix) -> int64_t { return CHUNKED ? (int64_t)a.cptr[ix] : ix; };
    auto load_side = [&](int64_t ix, int64_t cx, float (&alx)[8], float &h0, float &h1) {
        // A operand of the aggregation product: alpha[edge 8g + s][head lo], s = 0..7 (two 16-byte loads); residual row
        // (L2H: K slots (2m, 2m + 1) = edges 4g + m, 16 + 4g + m)
        if constexpr (L2H) {
            const float *ap = a.alpha + ((size_t)cx * TD_HEADS + lo) * TD_K + 4 * g;
            const float4 v0 = *reinterpret_cast<const float4 *>(ap), v1 = *reinterpret_cast<const float4 *>(ap + 16);
            alx[0] = v0.x; alx[1] = v1.x; alx[2] = v0.y; alx[3] = v1.y; alx[4] = v0.z; alx[5] = v1.z; alx[6] = v0.w; alx[7] = v1.w;
        } else {
            const float *ap = a.alpha + ((size_t)cx * TD_HEADS + lo) * TD_K + 8 * g;
            const float4 v0 = *reinterpret_cast<const float4 *>(ap), v1 = *reinterpret_cast<const float4 *>(ap + 4);
            alx[0] = v0.x; alx[1] = v0.y; alx[2] = v0.z; alx[3] = v0.w; alx[4] = v1.x; alx[5] = v1.y; alx[6] = v1.z; alx[7] = v1.w;
        }
        h0 = a.h[(size_t)ix * TD_H + nout];
        h1 = a.h[(size_t)ix * TD_H + nout + 4];
    };
    // General graphs: the protein workgroups of a graph whose protein rows are one chunk wide (`hybrid`: plain k-NN rows, k <= 32)
    // take the software-pipelined single-chunk loop below (chunk index through cptr); everything else walks chunks here.  (With the 16-bit
    // first layer td_launch_edge_value16 sends such protein rows to edge_value16t_kernel<.., VIA> and only the ligand rows here.)
    if (CHUNKED && !(SPLIT && my_cls == 1 && a.cpn_p == 1)) {
        for (int64_t i = next_row(); i >= 0; i = next_row()) {
            if (SPLIT && dyn && !of_class(a.x4[i])) continue;
            const int c0 = __builtin_amdgcn_readfirstlane(a.cptr[i]), c1 = __builtin_amdgcn_readfirstlane(a.cptr[i + 1]);
            const float hres0 = a.h[(size_t)i * TD_H + nout], hres1 = a.h[(size_t)i * TD_H + nout + 4];
            floatx4_t zb[8];
#pragma unroll
            for (int hb = 0; hb < 8; ++hb) zb[hb] = floatx4_t{0.f, 0.f, 0.f, 0.f};
            float asum = 0.f;
            // the neighbour indices of chunk c + 1 are fetched while chunk c runs (unconditionally: the row's last chunk stands in
            // for the one after it), so that a chunk starts with its gathers instead of two dependent round trips
            RowIn16 rin;
            td_row_index16(a, i, c0, lane, rin);
            for (int c = c0; c < c1; ++c) {
                floatx4_t acc[2][8];
                Edge2 ed;
                td_row_gather16<false>(a, i, c, lane, rin, acc);
                RowIn16 rcur = rin;
                {
                    const int cn = c + 1 < c1 ? c + 1 : c;
                    rin.j[0] = a.nbr[(int64_t)cn * TD_K + lo];
                    rin.j[1] = a.nbr[(int64_t)cn * TD_K + 16 + lo];
                }
                // The chunk's body in two versions (the choice is wave-uniform).  FULL: as in the pipelined loop below -- k-step s of the
                // aggregation product carries edge 8g + s.  Half (the second 16-edge block is all padding, e.g. slots 48 .. 63 of a row at
                // k = 48): first layer, flips and products of the first block only, k-step j = edge 4g + j.
                auto chunk_body = [&](auto full_tag) {
                    constexpr bool FULL = decltype(full_tag)::value;
                    if constexpr (L2H) {
                        // f16 piece pairs (see edge_value16t_kernel): K slots (g, 2m) / (g, 2m + 1) = edges 4g + m / 16 + 4g + m of the chunk; a
                        // half-empty chunk runs the first layer of its first block only and carries zeros in the second block's K slots
                        float al[8];
                        {
                            const float *ap = a.alpha + ((size_t)c * TD_HEADS + lo) * TD_K + 4 * g;
                            const float4 v0 = *reinterpret_cast<const float4 *>(ap);
                            float4 v1 = make_float4(0.f, 0.f, 0.f, 0.f);
                            if constexpr (FULL) v1 = *reinterpret_cast<const float4 *>(ap + 16);
                            al[0] = v0.x; al[1] = v1.x; al[2] = v0.y; al[3] = v1.y; al[4] = v0.z; al[5] = v1.z; al[6] = v0.w; al[7] = v1.w;
                        }
                        td_first_layer_split16<false, true, false, FULL ? 2 : 1, false, VPK, 0, TdNoHook, true>(a, reinterpret_cast<const uint4 *>(lds), KB, offk, rcur, i, lane, acc, ed);
                        asum += ((al[0] + al[1]) + (al[2] + al[3])) + ((al[4] + al[5]) + (al[6] + al[7]));
                        uint4 z1[8], z2[8];
                        td_ln_relu16_pairs_eb<!ZPLAIN, FULL ? 2 : 1>(KB, g, acc, TdLn{a.mlp.ln_c1, a.mlp.ln_c2}, z1, z2);
                        uint4 aq1, aq2;
                        {
                            unsigned p1[4], p2[4];
                            const float ax[4] = {al[0] * 1024.0f, al[2] * 1024.0f, al[4] * 1024.0f, al[6] * 1024.0f};
                            const float ay[4] = {al[1] * 1024.0f, al[3] * 1024.0f, al[5] * 1024.0f, al[7] * 1024.0f};
                            td_split_h2_x4(ax, ay, p1, p2);
                            aq1 = make_uint4(p1[0], p1[1], p1[2], p1[3]);
                            aq2 = make_uint4(p2[0], p2[1], p2[2], p2[3]);
                        }
                        constexpr int TWS = 20, TWP = 16 * TWS;
                        auto flip_store_h = [&](int hb) {
                            unsigned *TW = reinterpret_cast<unsigned *>(TB + (hb & 1) * V16_TILE_FLOATS);
                            *reinterpret_cast<uint4 *>(TW + lo * TWS + 4 * g) = z1[hb];
                            *reinterpret_cast<uint4 *>(TW + TWP + lo * TWS + 4 * g) = z2[hb];
                        };
                        auto flip_load_h = [&](int hb, uint4 (&zq)[2]) {
                            const unsigned *t = reinterpret_cast<const unsigned *>(TB + (hb & 1) * V16_TILE_FLOATS) + 4 * g * TWS + lo;
                            zq[0] = make_uint4(t[0], t[TWS], t[2 * TWS], t[3 * TWS]);
                            zq[1] = make_uint4(t[TWP], t[TWP + TWS], t[TWP + 2 * TWS], t[TWP + 3 * TWS]);
                        };
                        uint4 zqb[2][2];
                        flip_store_h(0);
                        flip_load_h(0, zqb[0]);
#pragma unroll
                        for (int hb = 0; hb < 8; ++hb) {
                            if (hb + 1 < 8) {
                                flip_store_h(hb + 1);
                                flip_load_h(hb + 1, zqb[(hb + 1) & 1]);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            zb[hb] = td_mfma16h(zqb[hb & 1][1], aq1, zb[hb]);          // (accumulates over the row's chunks)
                            zb[hb] = td_mfma16h(zqb[hb & 1][0], aq2, zb[hb]);
                            zb[hb] = td_mfma16h(zqb[hb & 1][0], aq1, zb[hb]);
                        }
                        return;
                    } else {
                    constexpr int NS = FULL ? 8 : 4;
                    float al[NS];
                    {   // A operand of the aggregation product: alpha[edge][head lo] of this chunk
                        const float *ap = a.alpha + ((size_t)c * TD_HEADS + lo) * TD_K + (FULL ? 8 : 4) * g;
                        const float4 v0 = *reinterpret_cast<const float4 *>(ap);
                        al[0] = v0.x; al[1] = v0.y; al[2] = v0.z; al[3] = v0.w;
                        if constexpr (FULL) {
                            const float4 v1 = *reinterpret_cast<const float4 *>(ap + 4);
                            al[4] = v1.x; al[5] = v1.y; al[6] = v1.z; al[7] = v1.w;
                        }
                    }
                    if constexpr (SPLIT)
                        td_first_layer_split16<false, true, false, FULL ? 2 : 1, false, VPK, 0>(a, reinterpret_cast<const uint4 *>(lds), KB, offk, rcur, i, lane, acc, ed);
                    else
                        td_first_layer_compute16<false, true>(a, Rt, KB, offk, rcur, lane, acc, ed);
                    float part = (al[0] + al[1]) + (al[2] + al[3]);
                    if constexpr (FULL) part += (al[4] + al[5]) + (al[6] + al[7]);
                    asum += part;
                    auto flip_store = [&](int hb) {
                        float *t = TB + (hb & 1) * V16_TILE_FLOATS;
#pragma unroll
                        for (int eb = 0; eb < (FULL ? 2 : 1); ++eb)
                            *reinterpret_cast<float4 *>(t + td_tile_row16(16 * eb + lo) + 4 * g) =
                                make_float4(acc[eb][hb][0], acc[eb][hb][1], acc[eb][hb][2], acc[eb][hb][3]);
                    };
                    // (half: rows 4g + j of the first block; groups 0 / 1 and 2 / 3 sit on different halves of the banks)
                    auto flip_load = [&](int hb, float (&bv)[NS]) {
                        const float *t = TB + (hb & 1) * V16_TILE_FLOATS;
#pragma unroll
                        for (int sx = 0; sx < NS; ++sx) bv[sx] = t[td_tile_row16((FULL ? 8 : 4) * g + sx) + lo];
                    };
                    float bvb[2][NS];
                    flip_store(0);
                    flip_load(0, bvb[0]);
#pragma unroll
                    for (int hb = 0; hb < 8; ++hb) {
                        if (hb + 1 < 8) {          // read back a block ahead of its products (see the pipelined loop)
                            flip_store(hb + 1);
                            flip_load(hb + 1, bvb[(hb + 1) & 1]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int sx = 0; sx < NS; ++sx) zb[hb] = td_mfma16(bvb[hb & 1][sx], al[sx], zb[hb]);
                    }
                    }
                };
                bool full = true;
                if (SPLIT) full = __ballot(rcur.j[1] >= 0) != 0ull;        // (the fp32 first layer has no one-block path)
                if (full) chunk_body(std::true_type());
                else chunk_body(std::false_type());
            }
            const float ssum = td_sum_groups(asum);          // S[head lo] over the node's chunks, in every lane group
            constexpr float OUT_SCALE = L2 == 0 ? 1.0f : (ZPLAIN ? 1.0f / 1024.0f : 1.0f / (1024.0f * TD_Z_SCALE));
            float o0, o1;
            td_value_out16(zb, Wt_lane, o0, o1);
            o0 = fmaf(B2[nout], ssum, o0 * OUT_SCALE);
            o1 = fmaf(B2[nout + 4], ssum, o1 * OUT_SCALE);
            if (a.out) {
                a.out[(size_t)i * TD_H + nout] = o0;
                a.out[(size_t)i * TD_H + nout + 4] = o1;
            } else {
                a.h[(size_t)i * TD_H + nout] = hres0 + o0;
                a.h[(size_t)i * TD_H + nout + 4] = hres1 + o1;
            }
        }
        trace_end();
        return;
    }
    // next row of this workgroup's class with its index loads issued (dyn + SPLIT: candidates of the other class -- the ligand
    // rows inside a protein workgroup's share, one in 25 -- are dropped here, at the price of one exposed load each)
    // The loads are issued unconditionally (row 0 stands in when the wave has run out of rows): behind a branch they make the
    // compiler's wait-count bookkeeping pessimistic at the join -- vmcnt counts in order, the path without the loads has fewer
    // outstanding, and the wait for the CURRENT row's gathers was emitted as if the next row's index loads, issued a moment
    // earlier, had to land first (s_waitcnt vmcnt(9) / (8) where vmcnt(11) is enough: one exposed round trip per row).
    auto next_indexed = [&](RowIn16 &r, int64_t &cx) -> int64_t {
        const int64_t ix = next_row();
        const int64_t safe = ix >= 0 ? ix : 0;
        cx = chunk_of(safe);
        td_row_index16(a, safe, cx, lane, r);
        return ix;
    };
    auto settle = [&](int64_t ix, RowIn16 &r, int64_t &cx) -> int64_t {
        if (SPLIT && dyn)
            while (ix >= 0 && !of_class(r.xi)) ix = next_indexed(r, cx);
        return ix;
    };
    RowIn16 rin;
    int64_t ci = 0;
    int64_t i = settle(next_indexed(rin, ci), rin, ci);
    floatx4_t acc[2][8];
    float al[8], hres0 = 0.f, hres1 = 0.f;
    if (i >= 0) {
        td_row_gather16<false>(a, i, ci, lane, rin, acc);
        load_side(i, ci, al, hres0, hres1);
    }
    while (i >= 0) {
        RowIn16 rnext;
        int64_t cnext = 0;
        int64_t inext = next_indexed(rnext, cnext);
        bool more = inext >= 0;
        Edge2 ed;
        if constexpr (SPLIT) {
            float offr[8];
            {
                int dep = 0;
                asm volatile("" : "+v"(dep));
                const float4 *op = reinterpret_cast<const float4 *>(B2 + 2 * TD_H + 4 + 8 * g + dep);
                const float4 o0 = op[0], o1 = op[1];
                offr[0] = o0.x; offr[1] = o0.y; offr[2] = o0.z; offr[3] = o0.w; offr[4] = o1.x; offr[5] = o1.y; offr[6] = o1.z; offr[7] = o1.w;
            }
            // (P_i joins before the products on the default graph, as in edge_value16t_kernel -- the same bits whichever kernel the row
            // distribution setting selects; the chunked instantiation hides the P_i loads behind the products instead)
            td_first_layer_split16<false, true, CHUNKED, 2, false, VPK, 1, TdNoHook, L2H>(a, reinterpret_cast<const uint4 *>(lds), KB, offr, rin, i, lane, acc, ed);
        }
        else
            td_first_layer_compute16<false>(a, Rt, KB, offk, rin, lane, acc, ed);

        if constexpr (GATE_M) {          // e_w = sigmoid((u' . z'_e) / sigma_e + c), multiplied into alpha before S is taken
            float part[2] = {0.f, 0.f};
#pragma unroll
            for (int hb = 0; hb < 8; ++hb) {
                const float4 u4 = *reinterpret_cast<const float4 *>(a.gate_m + 16 * hb + 4 * g);
#pragma unroll
                for (int eb = 0; eb < 2; ++eb) {
                    part[eb] = fmaf(acc[eb][hb][0], u4.x, part[eb]); part[eb] = fmaf(acc[eb][hb][1], u4.y, part[eb]);
                    part[eb] = fmaf(acc[eb][hb][2], u4.z, part[eb]); part[eb] = fmaf(acc[eb][hb][3], u4.w, part[eb]);
                }
            }
            const float cm = a.gate_m[TD_H];
            float gm[2];
#pragma unroll
            for (int eb = 0; eb < 2; ++eb) gm[eb] = 1.0f / (1.0f + expf(-(td_sum_groups(part[eb]) + cm)));
            scale_edges(al, gm[0], gm[1]);
        }
        float ssum = ((al[0] + al[1]) + (al[2] + al[3])) + ((al[4] + al[5]) + (al[6] + al[7]));
        ssum = td_sum_groups(ssum);                    // S[head lo] = sum over the 32 edges, in every lane group

        // ---- Zbar[head][k] = sum_e alpha[e][head] z[e][k], one hidden block at a time: flip z^T (lane = edge) through
        //      the wave-private tile into the B layout (lane = hidden unit), 8 k-steps over the 32 edges ------------------
        // Two tiles ping-pong so that the flip of block hb + 1 is in flight while block hb feeds the MFMAs.
        floatx4_t zb[8];
        float out_scale = 1.0f;
        if constexpr (L2H) {          // see edge_value16t_kernel; two tiles ping-pong here
            uint4 z1[8], z2[8];
            td_ln_relu16_pairs_eb<!ZPLAIN>(KB, g, acc, TdLn{a.mlp.ln_c1, a.mlp.ln_c2}, z1, z2);
            uint4 aq1, aq2;
            {
                unsigned p1[4], p2[4];
                const float ax[4] = {al[0] * 1024.0f, al[2] * 1024.0f, al[4] * 1024.0f, al[6] * 1024.0f};
                const float ay[4] = {al[1] * 1024.0f, al[3] * 1024.0f, al[5] * 1024.0f, al[7] * 1024.0f};
                td_split_h2_x4(ax, ay, p1, p2);
                aq1 = make_uint4(p1[0], p1[1], p1[2], p1[3]);
                aq2 = make_uint4(p2[0], p2[1], p2[2], p2[3]);
            }
            constexpr int TWS = 20, TWP = 16 * TWS;
            auto flip_store_h = [&](int hb) {
                unsigned *TW = reinterpret_cast<unsigned *>(TB + (hb & 1) * V16_TILE_FLOATS);
                *reinterpret_cast<uint4 *>(TW + lo * TWS + 4 * g) = z1[hb];
                *reinterpret_cast<uint4 *>(TW + TWP + lo * TWS + 4 * g) = z2[hb];
            };
            auto flip_load_h = [&](int hb, uint4 (&zq)[2]) {
                const unsigned *t = reinterpret_cast<const unsigned *>(TB + (hb & 1) * V16_TILE_FLOATS) + 4 * g * TWS + lo;
                zq[0] = make_uint4(t[0], t[TWS], t[2 * TWS], t[3 * TWS]);
                zq[1] = make_uint4(t[TWP], t[TWP + TWS], t[TWP + 2 * TWS], t[TWP + 3 * TWS]);
            };
            uint4 zqb[2][2];
            flip_store_h(0);
            flip_load_h(0, zqb[0]);
#pragma unroll
            for (int hb = 0; hb < 8; ++hb) {
                if (hb + 1 < 8) {
                    flip_store_h(hb + 1);
                    flip_load_h(hb + 1, zqb[(hb + 1) & 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                zb[hb] = floatx4_t{0.f, 0.f, 0.f, 0.f};
                zb[hb] = td_mfma16h(zqb[hb & 1][1], aq1, zb[hb]);
                zb[hb] = td_mfma16h(zqb[hb & 1][0], aq2, zb[hb]);
                zb[hb] = td_mfma16h(zqb[hb & 1][0], aq1, zb[hb]);
            }
            out_scale = ZPLAIN ? 1.0f / 1024.0f : 1.0f / (1024.0f * TD_Z_SCALE);
        } else {
        auto flip_store = [&](int hb) {
            float *t = TB + (hb & 1) * V16_TILE_FLOATS;
#pragma unroll
            for (int eb = 0; eb < 2; ++eb)
                *reinterpret_cast<float4 *>(t + td_tile_row16(16 * eb + lo) + 4 * g) =
                    make_float4(acc[eb][hb][0], acc[eb][hb][1], acc[eb][hb][2], acc[eb][hb][3]);
        };
        // block hb + 1 is flipped AND read back (B[edge 8g + s][hidden 16(hb + 1) + lo]) before block hb's products are issued: the
        // reads' latency runs behind eight MFMAs instead of in front of them
        auto flip_load = [&](int hb, float (&bv)[8]) {
            const float *t = TB + (hb & 1) * V16_TILE_FLOATS;
#pragma unroll
            for (int s = 0; s < 8; ++s) bv[s] = t[td_tile_row16(8 * g + s) + lo];
        };
        float bvb[2][8];
        flip_store(0);
        flip_load(0, bvb[0]);
#pragma unroll
        for (int hb = 0; hb < 8; ++hb) {
            if (hb + 1 < 8) {
                flip_store(hb + 1);
                flip_load(hb + 1, bvb[(hb + 1) & 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            zb[hb] = floatx4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 8; ++s) zb[hb] = td_mfma16(bvb[hb & 1][s], al[s], zb[hb]);
        }

        }
        // ---- next row: gathers into the (now free) accumulators, its alpha fragment and residual ---------------------
        const int64_t icur = i;
        const float hcur0 = hres0, hcur1 = hres1;
        if (SPLIT && dyn && more) {
            inext = settle(inext, rnext, cnext);
            more = inext >= 0;
        }
        {   // (unconditional for the same reason: after the last row these are loads of row 0 that nobody reads)
            const int64_t gsafe = more ? inext : 0;
            td_row_gather16<false>(a, gsafe, more ? cnext : chunk_of(0), lane, rnext, acc);
            load_side(gsafe, more ? cnext : chunk_of(0), al, hres0, hres1);
            rin = rnext;
        }
        i = inext;

        // ---- out[n] = W2v[n, :] . Zbar[head(n), :] + b2v[n] S[head(n)];  h_i += out   (td_value_out16: outputs nout, nout + 4) -------
        float o0, o1;
        td_value_out16(zb, Wt_lane, o0, o1);
        o0 = fmaf(B2[nout], ssum, o0 * out_scale);
        o1 = fmaf(B2[nout + 4], ssum, o1 * out_scale);
        if (a.out) {
            a.out[(size_t)icur * TD_H + nout] = o0;
            a.out[(size_t)icur * TD_H + nout + 4] = o1;
        } else {
            a.h[(size_t)icur * TD_H + nout] = hcur0 + o0;
            a.h[(size_t)icur * TD_H + nout + 4] = hcur1 + o1;
        }
    }
    trace_end();
}

// ---- the x2h value pass of the default graph at THREE waves per SIMD (round 5) ------------------------------------------------------
// bf16 first layer, 32-slot rows, rows handed out through the LDS ticket (edge_row_dealing = 2: the default).  Against
// edge_value16_kernel: no software pipeline across rows (a row's gathers are hidden by the two other waves of its SIMD, as in the key
// pass) and ONE flip tile per wave (the LDS operations of a wave execute in order: the store of block hb + 1 cannot overtake the reads of
// block hb).  Same products in the same order per accumulator as the 8-wave kernel: bit-identical h.  Measured in one call each, C2 value
// pass per step: 8-wave pipelined 1.609 -> 12 waves 1.525 (10 / 11 / 12 waves: 1.676 / 1.642 / 1.606, every wave counts) -> output
// product without the Zbar exchange (td_value_out16) 1.508 -> 1.440 -> with the 16.5 KiB of scratch that freed, both source-class tables
// of the workgroup's destination class in the 48-byte form (PK = 1: three 16-byte reads per hidden block) 1.397.  156 registers,
// 48 + 64 KiB of tables + 12 x 2.75 KiB of scratch = 146.4 KiB.  (Table quads a pair ahead -- 164 registers -- changes nothing: 1.395.)
// Round 6: second layer on f16 piece pairs (L2), then the first layer too (FL = 1: 32 instead of 48 KiB of tables, 150 registers): 1.34 -> 1.27.
constexpr int V16T_WAVES = 12;
#ifndef TD_VALUET_PK
#define TD_VALUET_PK 1
#endif
constexpr int V16T_PK = TD_VALUET_PK;
#ifndef TD_VALUET_AH
#define TD_VALUET_AH -1
#endif
constexpr int V16T_AH = TD_VALUET_AH;                               // table quads one quad ahead (td_pk4_tiles)
constexpr int V16T_WAVE_FLOATS = V16_TILE_FLOATS;         // 704 floats: one flip tile
constexpr size_t V16T_LDS_BYTES =
    (size_t)(e16q_half_u4<V16T_PK>() * 4 + V16_W_FLOATS + V16T_WAVES * V16T_WAVE_FLOATS + 2 * TD_H + 4 + 32) * sizeof(float);
static_assert(V16T_LDS_BYTES <= 160 * 1024, "value pass (12 waves): LDS");

// L2: the aggregation product -- 0 fp32, 1 f16 piece pairs of z'' 2^15, 2 f16 piece pairs of z'' (see edge_key16_kernel)
// VIA (`hybrid` graphs: protein rows of one chunk, found through cptr): the launcher passes no ligand rows -- every workgroup serves the
// protein class and drops the ligand rows it meets; those are walked chunk by chunk by edge_value16_kernel<true, true> in a second launch.
// FL: 1 = the first layer on f16 piece pairs (PK = 5, TdEdgeMlp::R16h; see edge_key16_kernel)
template <int L2, bool VIA = false, int FL = 0>
__global__ __launch_bounds__(V16T_WAVES * 64) void edge_value16t_kernel(Args16 a) {
    constexpr bool ZPLAIN = L2 == 2;
    constexpr int WAVES = V16T_WAVES;
    constexpr int VPK = FL ? 5 : V16T_PK;
    constexpr int RF = e16q_half_u4<VPK>() * 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;          // lds + RF: Wt[d 8][kq 32][head 16] x 4 k (td_value_out16)
    const int lo = lane & 15, g = lane >> 4;
    float *TB = lds + RF + V16_W_FLOATS + wid * V16T_WAVE_FLOATS;            // wave-private scratch
    float *B2 = lds + RF + V16_W_FLOATS + WAVES * V16T_WAVE_FLOATS;
    const float *KB = B2 + TD_H;
    // workgroups [0, GP) serve the protein rows (class 1), [GP, gridDim.x) the ligand rows (class 0): as in edge_value16_kernel
    int my_cls = 1, GL = 0;
    const int64_t n_rows = a.count_ptr ? (int64_t)*a.count_ptr : a.count;
    {
        const int64_t G = gridDim.x, nl = a.lig_count, np = n_rows > nl ? n_rows - nl : 0;
        if (nl > 0) {
            int64_t nm = a.mixed_count ? (int64_t)*a.mixed_count - nl : (np * 3) / 10;
            nm = nm < 0 ? 0 : (nm > np ? np : nm);
            const int64_t wl = TD_ROW_COST_MIXED * nl, wp = TD_ROW_COST_PURE * (np - nm) + TD_ROW_COST_MIXED * nm;
            GL = (int)((wl * G + (wl + wp) / 2) / (wl + wp));
            const int cap = (int)G - (np > 0 ? 1 : 0);
            GL = GL < 1 ? 1 : (GL > cap ? cap : GL);
        }
        my_cls = (int)blockIdx.x >= (int)G - GL ? 0 : 1;
    }
    const int GP = gridDim.x - GL;
    if (a.trace && threadIdx.x == 0) a.trace[8 * blockIdx.x + 4] = __builtin_amdgcn_s_memrealtime();      // kernel entry
    {
        if constexpr (VPK == 5)
            td_stage_lds16(reinterpret_cast<const float4 *>(a.mlp.R16h) + (size_t)my_cls * 2 * e16q_cs_u4<5>(), reinterpret_cast<float4 *>(lds), 2 * e16q_cs_u4<5>(), tid, WAVES * 64);
        else if constexpr (V16T_PK == 4) {
            td_stage_pk4<2>(a.mlp.R16q + (size_t)my_cls * 2 * E16Q_GLOBAL_CS_U4 * 4, lds, 1, tid, WAVES * 64);
            td_stage_pk4<1>(a.mlp.R16q + (size_t)(my_cls * 2 + 1) * E16Q_GLOBAL_CS_U4 * 4, lds + e16q_cs_u4<2>() * 4, 1, tid, WAVES * 64);
        } else
            td_stage_pk4<V16T_PK>(a.mlp.R16q + (size_t)my_cls * 2 * E16Q_GLOBAL_CS_U4 * 4, lds, 2, tid, WAVES * 64);
        td_stage_lds16(reinterpret_cast<const float4 *>(a.mlp.Walt), reinterpret_cast<float4 *>(lds + RF), V16_W_FLOATS / 4, tid, WAVES * 64);
        if (tid < TD_H) B2[tid] = a.mlp.b2[tid];
        else if (tid < 2 * TD_H) B2[tid] = a.mlp.beta[tid - TD_H];
        else if (tid == 2 * TD_H) *reinterpret_cast<int *>(B2 + 2 * TD_H) = 0;
        else if (tid >= 2 * TD_H + 32 && tid < 2 * TD_H + 64) {
            const int s8 = tid - (2 * TD_H + 32);
            B2[2 * TD_H + 4 + s8] = (s8 & 7) < 5 ? a.offsets[5 * (s8 >> 3) + (s8 & 7)] : TD_FAR_CENTRE;
        }
    }
    __syncthreads();
    if (a.trace && tid == 0) a.trace[8 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    // rows: protein workgroups take single rows dealt to the XCD's workgroups, through the workgroup's LDS ticket; the ligand workgroups
    // static shares of the ligand row list
    const int32_t *list = a.rows;
    int64_t first, end, stride = 1;
    if (my_cls) td_deal16(n_rows, 1, first, end, stride, GP, (int)blockIdx.x);
    else {
        list = a.lig_rows;
        const int64_t per = (a.lig_count + GL - 1) / GL;
        first = ((int)blockIdx.x - GP) * per;
        end = first + per < a.lig_count ? first + per : a.lig_count;
    }
    int *row_ctr = reinterpret_cast<int *>(B2 + 2 * TD_H);
    auto next_row = [&]() -> int64_t {
        int n = 0;
        if (lane == 0) n = __hip_atomic_fetch_add(row_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        n = __builtin_amdgcn_readfirstlane(n);
        const int64_t itx = first + (int64_t)n * stride;
        return itx < end ? (int64_t)__builtin_amdgcn_readfirstlane((int)(list ? (int64_t)list[itx] : itx)) : -1;      // (wave-uniform, and said so)
    };
    const int nout = td_value_out_index16(lo, g);
    // (opaque: as a compile-time constant the table's offset is folded into the immediates of the 64 reads and no longer fits their 16 bits)
    int woff = RF + 4 * lane;
    asm volatile("" : "+v"(woff));
    const float *Wt_lane = lds + woff;
    for (int64_t i = next_row(); i >= 0; i = next_row()) {
        RowIn16 rin;
        const int64_t c = VIA ? (int64_t)__builtin_amdgcn_readfirstlane(a.cptr[i]) : i;          // the row of nbr / alpha
        td_row_index16(a, i, c, lane, rin);
        // a protein workgroup meets the ligand rows of its share as candidates and drops them
        if (((__builtin_amdgcn_readfirstlane(__float_as_int(rin.xi.w)) > __float_as_int(0.5f)) ? 0 : 1) != my_cls) continue;
        floatx4_t acc[2][8];
        td_row_gather16<false>(a, i, c, lane, rin, acc);
        float al[8];
        if constexpr (L2 != 0) {
            // B operand of the aggregation product, K slots (g, 2m) / (g, 2m + 1) = edges 4g + m / 16 + 4g + m (the pairing the flipped z''
            // words have): alpha[edge][head lo], two 16-byte loads
            const float *ap = a.alpha + ((size_t)c * TD_HEADS + lo) * TD_K + 4 * g;
            const float4 v0 = *reinterpret_cast<const float4 *>(ap), v1 = *reinterpret_cast<const float4 *>(ap + 16);
            al[0] = v0.x; al[1] = v1.x; al[2] = v0.y; al[3] = v1.y; al[4] = v0.z; al[5] = v1.z; al[6] = v0.w; al[7] = v1.w;
        } else {   // A operand of the aggregation product: alpha[edge 8g + s][head lo]
            const float *ap = a.alpha + ((size_t)c * TD_HEADS + lo) * TD_K + 8 * g;
            const float4 v0 = *reinterpret_cast<const float4 *>(ap), v1 = *reinterpret_cast<const float4 *>(ap + 4);
            al[0] = v0.x; al[1] = v0.y; al[2] = v0.z; al[3] = v0.w; al[4] = v1.x; al[5] = v1.y; al[6] = v1.z; al[7] = v1.w;
        }
        const float hres0 = a.h[(size_t)i * TD_H + nout], hres1 = a.h[(size_t)i * TD_H + nout + 4];
        Edge2 ed;
        float offr[8];
        {
            int dep = 0;
            asm volatile("" : "+v"(dep));
            const float4 *op = reinterpret_cast<const float4 *>(B2 + 2 * TD_H + 4 + 8 * g + dep);
            const float4 o0 = op[0], o1 = op[1];
            offr[0] = o0.x; offr[1] = o0.y; offr[2] = o0.z; offr[3] = o0.w; offr[4] = o1.x; offr[5] = o1.y; offr[6] = o1.z; offr[7] = o1.w;
        }
        float ssum = ((al[0] + al[1]) + (al[2] + al[3])) + ((al[4] + al[5]) + (al[6] + al[7]));
        ssum = td_sum_groups(ssum);                    // S[head lo] = sum over the 32 edges, in every lane group
        floatx4_t zb[8];
        constexpr float OUT_SCALE = L2 == 0 ? 1.0f : (ZPLAIN ? 1.0f / 1024.0f : 1.0f / (1024.0f * TD_Z_SCALE));
        if constexpr (L2 != 0) {
        // Zbar^T on v_mfma_f32_16x16x32_f16: zb[hb][r] = 2^10 sum_e z''[e][16hb + 4g + r] alpha[e][head lo], K = the row's 32 edges in ONE
        // instruction per piece product and hidden block.  z'' leaves the LayerNorm as two f16 pieces, the lane's two edges in one word
        // (td_ln_relu16_pairs_eb); the flip through the wave's tile moves WORDS: lane (edge lo, g) stores its four hidden units 4g .. 4g + 3 of
        // the block as one 16-byte row segment, lane (hidden lo, g) reads the words of edges 4g .. 4g + 3 in its column -- four K-slot pairs,
        // the A operand.  Tiles: [16 edge rows][20 words] per piece (the row stride of 20 keeps both the 16-byte stores and the 4-byte reads
        // conflict-free, as in the fp32 form).  alpha goes in scaled by 2^10 and z'' by 2^15 (exact) so that small weights and
        // activations keep their 22 bits above the f16 subnormal floor; the scales come off the two outputs.
        td_first_layer_split16<false, true, false, 2, false, VPK, V16T_AH, TdNoHook, true>(a, reinterpret_cast<const uint4 *>(lds), KB, offr, rin, i, lane, acc, ed);
        uint4 z1[8], z2[8];
        td_ln_relu16_pairs_eb<!ZPLAIN>(KB, g, acc, TdLn{a.mlp.ln_c1, a.mlp.ln_c2}, z1, z2);
        uint4 aq1, aq2;
        {
            unsigned p1[4], p2[4];
            const float ax[4] = {al[0] * 1024.0f, al[2] * 1024.0f, al[4] * 1024.0f, al[6] * 1024.0f};
            const float ay[4] = {al[1] * 1024.0f, al[3] * 1024.0f, al[5] * 1024.0f, al[7] * 1024.0f};
            td_split_h2_x4(ax, ay, p1, p2);
            aq1 = make_uint4(p1[0], p1[1], p1[2], p1[3]);
            aq2 = make_uint4(p2[0], p2[1], p2[2], p2[3]);
        }
        unsigned *TW = reinterpret_cast<unsigned *>(TB);
        constexpr int TWS = 20, TWP = 16 * TWS;                      // row stride and piece stride of the flip tile, words
        auto flip_store = [&](int hb) {
            *reinterpret_cast<uint4 *>(TW + lo * TWS + 4 * g) = z1[hb];
            *reinterpret_cast<uint4 *>(TW + TWP + lo * TWS + 4 * g) = z2[hb];
        };
        auto flip_load = [&](uint4 (&zq)[2]) {
            const unsigned *t = TW + 4 * g * TWS + lo;
            zq[0] = make_uint4(t[0], t[TWS], t[2 * TWS], t[3 * TWS]);
            zq[1] = make_uint4(t[TWP], t[TWP + TWS], t[TWP + 2 * TWS], t[TWP + 3 * TWS]);
        };
        uint4 zqb[2][2];
        flip_store(0);
        flip_load(zqb[0]);
#pragma unroll
        for (int hb = 0; hb < 8; ++hb) {
            if (hb + 1 < 8) {
                flip_store(hb + 1);
                flip_load(zqb[(hb + 1) & 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            zb[hb] = floatx4_t{0.f, 0.f, 0.f, 0.f};
            zb[hb] = td_mfma16h(zqb[hb & 1][1], aq1, zb[hb]);          // the small products first
            zb[hb] = td_mfma16h(zqb[hb & 1][0], aq2, zb[hb]);
            zb[hb] = td_mfma16h(zqb[hb & 1][0], aq1, zb[hb]);
        }
        } else {
        td_first_layer_split16<false, true, false, 2, false, VPK, V16T_AH>(a, reinterpret_cast<const uint4 *>(lds), KB, offr, rin, i, lane, acc, ed);
        // Zbar^T: zb[hb][r] = sum_e z[e][16hb + 4g + r] alpha[e][head lo], one hidden block at a time through the wave's flip tile
        auto flip_store = [&](int hb) {
#pragma unroll
            for (int eb = 0; eb < 2; ++eb)
                *reinterpret_cast<float4 *>(TB + td_tile_row16(16 * eb + lo) + 4 * g) =
                    make_float4(acc[eb][hb][0], acc[eb][hb][1], acc[eb][hb][2], acc[eb][hb][3]);
        };
        auto flip_load = [&](float (&bv)[8]) {
#pragma unroll
            for (int sx = 0; sx < 8; ++sx) bv[sx] = TB[td_tile_row16(8 * g + sx) + lo];
        };
        float bvb[2][8];
        flip_store(0);
        flip_load(bvb[0]);
#pragma unroll
        for (int hb = 0; hb < 8; ++hb) {
            if (hb + 1 < 8) {
                flip_store(hb + 1);
                flip_load(bvb[(hb + 1) & 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            zb[hb] = floatx4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sx = 0; sx < 8; ++sx) zb[hb] = td_mfma16(bvb[hb & 1][sx], al[sx], zb[hb]);
        }
        }
        // out[n] = W2v[n, :] . Zbar[head(n), :] + b2v[n] S[head(n)];  h_i += out   (td_value_out16: this lane's outputs nout, nout + 4)
        float o0, o1;
        td_value_out16(zb, Wt_lane, o0, o1);
        o0 = fmaf(B2[nout], ssum, o0 * OUT_SCALE);
        o1 = fmaf(B2[nout + 4], ssum, o1 * OUT_SCALE);
        if (a.out) {
            a.out[(size_t)i * TD_H + nout] = o0;
            a.out[(size_t)i * TD_H + nout + 4] = o1;
        } else {
            a.h[(size_t)i * TD_H + nout] = hres0 + o0;
            a.h[(size_t)i * TD_H + nout + 4] = hres1 + o1;
        }
    }
    if (a.trace) {
        td_trace_wave_end(a.trace, lane);
        __syncthreads();
        if (tid == 0) a.trace[8 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

// ================================================================================================ edge gate
// e_w = sigmoid(MLP(20 -> 128 -> 1)(GaussianSmearing(dist)))  (models/uni_transformer.py:312-316): the attention passes' tile
// layout with a bias instead of the node projections and a 128-wide dot product instead of the second product.  One wave per
// 32-slot row (a dst node, or a chunk of a node's in-edges on general graphs), first layer on bf16 piece triples (the 24 KiB
// K-packed table, bias, LayerNorm affine and output weights in LDS), LayerNorm + ReLU in the transposed layout, then each lane
// dots its 32 hidden units of an edge with w3 and the four lane groups add up.
constexpr int TD_GATE_WGS = 1024;   // four 4-wave workgroups per CU (110 VGPRs, 25.5 KiB LDS each): gate 0.043 -> 0.038 ms per C2 step against three
constexpr int G16_WAVES = 4;        // 4-wave workgroups, TD_GATE_WGS / 256 = four of them per CU
constexpr int G16Q_U4 = e16q_cs_u4<1>();                                // the gate's K-packed table (QA, QB, QC): 24 KiB
constexpr size_t G16_LDS_BYTES = (size_t)G16Q_U4 * 16 + (size_t)3 * TD_H * sizeof(float);

__global__ __launch_bounds__(G16_WAVES * 64) void edge_gate16_kernel(TdGate gt, const float4 *__restrict__ x4,
                                                                     const int32_t *__restrict__ nbr, int64_t N,
                                                                     const int32_t *__restrict__ rows,
                                                                     const int32_t *__restrict__ count_ptr,
                                                                     const int32_t *__restrict__ chunk_node, float *__restrict__ ew) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const uint4 *Rp = reinterpret_cast<const uint4 *>(lds);
    float *B0 = lds + G16Q_U4 * 4, *BET = B0 + TD_H, *W3 = BET + TD_H;      // bias, beta / (|gamma| M) (folded LayerNorm), output weights
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lo = lane & 15, g = lane >> 4;
    {
        td_stage_pk4<1>(gt.R16q, lds, 1, tid, G16_WAVES * 64);
        for (int t = tid; t < 3 * TD_H; t += G16_WAVES * 64) {
            const int n = t & (TD_H - 1);
            B0[t] = t < TD_H ? gt.b0[n] : (t < 2 * TD_H ? gt.beta[n] : gt.w3[n]);
        }
    }
    float offj[5];          // lane group g owns the Gaussians 5g .. 5g + 4 (K-packed products, td_pk4_tiles)
#pragma unroll
    for (int j = 0; j < 5; ++j) offj[j] = gt.offsets[5 * g + j];
    __syncthreads();
    int64_t begin, end;
    td_node_range16(N, count_ptr, begin, end);
    const float c2 = gt.coeff * TD_LOG2E;

    for (int64_t it = begin + wid; it < end; it += G16_WAVES) {
        const int64_t row = rows ? (int64_t)rows[it] : it;                       // row of nbr / ew
        const int64_t i = chunk_node ? (int64_t)chunk_node[row] : row;          // its dst node
        const float4 xi = x4[i];
        int jn[2];
        jn[0] = nbr[row * TD_K + lo];
        jn[1] = nbr[row * TD_K + 16 + lo];
        floatx4_t acc[2][8];
        uint4 bq[2][4];
        bool valid[2];
#pragma unroll
        for (int eb = 0; eb < 2; ++eb) {
            valid[eb] = jn[eb] >= 0;
            const float4 xj = x4[valid[eb] ? jn[eb] : (int)i];
            const float rx = xi.x - xj.x, ry = xi.y - xj.y, rz = xi.z - xj.z;
            const float dist = td_sqrt_d2(rx * rx + ry * ry + rz * rz);
            float gv[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const float u = dist - offj[j];
                gv[j] = __builtin_amdgcn_exp2f(c2 * (u * u));
            }
            td_pk4_bquads(gv, 0u, bq[eb]);          // (no type column: the gate's first layer is the 20 Gaussians)
#pragma unroll
            for (int hb = 0; hb < 8; ++hb) {
                const float4 b = *reinterpret_cast<const float4 *>(B0 + 16 * hb + 4 * g);
                acc[eb][hb][0] = b.x; acc[eb][hb][1] = b.y; acc[eb][hb][2] = b.z; acc[eb][hb][3] = b.w;
            }
        }
        td_pk4_tiles<1, 1, 2>(Rp + lane, lane, bq, acc);
        td_ln_relu16<2>(BET, g, acc, TdLn{gt.ln_c1, gt.ln_c2});
#pragma unroll
        for (int eb = 0; eb < 2; ++eb) {
            float part = 0.f;
#pragma unroll
            for (int hb = 0; hb < 8; ++hb) {
                const float4 w = *reinterpret_cast<const float4 *>(W3 + 16 * hb + 4 * g);
                part = fmaf(acc[eb][hb][0], w.x, part); part = fmaf(acc[eb][hb][1], w.y, part);
                part = fmaf(acc[eb][hb][2], w.z, part); part = fmaf(acc[eb][hb][3], w.w, part);
            }
            const float logit = td_sum_groups(part) + gt.b3;
            if (g == 0) ew[row * TD_K + 16 * eb + lo] = valid[eb] ? 1.0f / (1.0f + expf(-logit)) : 0.f;
        }
    }
}

// ================================================================================================ launchers
// Workgroups of a row-loop kernel: one per CU (256) when there is a row per wave for all of them.  With fewer rows the
// waves of a workgroup would queue on their SIMD's shared matrix / vector pipe while other CUs idle, so small launches
// spread over as many workgroups as there are rows (up to one per CU) and leave part of each workgroup's waves without a row.
#define TD_LDS_ONCE(fn, bytes) do { static TdLdsOnce once; int _rc = td_set_lds(once, reinterpret_cast<const void *>(fn), bytes); if (_rc != TD_OK) return _rc; } while (0)

static int grid16(int64_t count, int waves) {
    int64_t g = (count + waves - 1) / waves;
    if (g < 256) g = count < 256 ? count : 256;
    if (g > 256) g = 256;
    if (g >= 8) g = (g / 8) * 8;
    return (int)(g < 1 ? 1 : g);
}

// td_debug_wg_trace: per-workgroup (start, end, first wave end, sum of wave ends) stamps of the x2h key / value launches
static unsigned long long *g_wg_trace = nullptr;
static int g_wg_trace_slots = 0, g_wg_trace_launch[3] = {0, 0, 0};
int td_set_wg_trace(unsigned long long *buf, int slots) {
    g_wg_trace = slots > 0 ? buf : nullptr;
    g_wg_trace_slots = slots;
    g_wg_trace_launch[0] = g_wg_trace_launch[1] = g_wg_trace_launch[2] = 0;
    return TD_OK;
}
bool td_wg_trace_armed() { return g_wg_trace != nullptr; }
static unsigned long long *wg_trace_slot(int pass) {
    if (!g_wg_trace) return nullptr;
    const int n = g_wg_trace_launch[pass]++ % g_wg_trace_slots;
    return g_wg_trace + ((size_t)n * 3 + pass) * 256 * 8;
}

// cptr (general graphs): chunks of dst node i = cptr[i] .. cptr[i+1]-1 of nbr / ew / alpha; nullptr: one 32-slot row per node.
// lig_rows / lig_count / cpn_p (x2h on general graphs): the ligand rows among `rows` (all of them) and the chunks per protein row.  With one
// chunk per protein row (`hybrid`) the protein rows run as on the default graph and the ligand rows in a second, chunk-walking launch.
int td_launch_edge_key16(const TdEdgeMlp &mlp, const TdLayer &L, const float4 *x4, const int32_t *nbr, const float *ew,
                         const float *P, const float *q, const int32_t *rows, const int32_t *count_ptr, int64_t count,
                         float *alpha, hipStream_t s, bool h2x_stage, const int32_t *cptr, const int32_t *lig_rows, int64_t lig_count, int cpn_p) {
    if (count == 0) return TD_OK;
    Args16 a = {};
    a.x4 = x4; a.nbr = nbr; a.ew = ew; a.P = P; a.q = q; a.rows = rows; a.count_ptr = count_ptr; a.h = nullptr;
    a.alpha = alpha; a.x4_out = nullptr; a.count = count; a.mlp = mlp; a.offsets = L.offsets; a.coeff = L.coeff; a.p_off = 0;
    a.cptr = cptr;
    const bool h2x = h2x_stage;      // h2x key pass (unfused form): STAGE tag 1, contiguous shares, no workgroup trace
    if (!h2x) a.trace = wg_trace_slot(0);
    a.deal = h2x ? 0 : mlp.deal_rows;
#define TD_KEY_LAUNCH(WAVES, STAGE, CH, SP, BYTES)                                                            \
    do {                                                                                                      \
        TD_LDS_ONCE((edge_key16_kernel<false, WAVES, STAGE, CH, SP>), BYTES);                                 \
        edge_key16_kernel<false, WAVES, STAGE, CH, SP><<<dim3(grid16(a.count, WAVES)), dim3(WAVES * 64), BYTES, s>>>(a); \
    } while (0)
    // (GR: the kernel's GRAPH; L2V: its second-layer variant; the first-layer variant follows TdEdgeMlp::l1_f16)
#define TD_KEY_LAUNCH_X2H(GR, L2V)                                                                                            \
    do {                                                                                                                  \
        if (mlp.l1_f16) {                                                                                                 \
            TD_LDS_ONCE((edge_key16_kernel<false, K16S_WAVES, 0, GR, true, L2V, 1>), K16S_LDS_BYTES);                      \
            edge_key16_kernel<false, K16S_WAVES, 0, GR, true, L2V, 1><<<dim3(grid16(a.count, K16S_WAVES)), dim3(K16S_WAVES * 64), K16S_LDS_BYTES, s>>>(a); \
        } else {                                                                                                          \
            TD_LDS_ONCE((edge_key16_kernel<false, K16S_WAVES, 0, GR, true, L2V, 0>), K16S_LDS_BYTES);                      \
            edge_key16_kernel<false, K16S_WAVES, 0, GR, true, L2V, 0><<<dim3(grid16(a.count, K16S_WAVES)), dim3(K16S_WAVES * 64), K16S_LDS_BYTES, s>>>(a); \
        }                                                                                                                 \
    } while (0)
#define TD_KEY_LAUNCH_WALK_L2(L2V) TD_KEY_LAUNCH_X2H(1, L2V)
    // the chunk walk's key pass takes f16 logits only together with the f16 first layer: with the bf16 piece triples the kernel spilled 9
    // registers at its 168 budget and lost (C5 k = 48 / k = 64 key pass 11.16 -> 11.40 / 12.78 -> 12.89 ms per step); with the f16 tables it
    // needs 158 and wins (10.54 -> 10.15 / 12.28 -> 11.65, one call each)
#if TD_KEY_WALK_F16
#define TD_KEY_LAUNCH_WALK_F16(L2V)                                                                                           \
    do {                                                                                                                  \
        TD_LDS_ONCE((edge_key16_kernel<false, K16S_WAVES, 0, 1, true, L2V, 1>), K16S_LDS_BYTES);                          \
        edge_key16_kernel<false, K16S_WAVES, 0, 1, true, L2V, 1><<<dim3(grid16(a.count, K16S_WAVES)), dim3(K16S_WAVES * 64), K16S_LDS_BYTES, s>>>(a); \
    } while (0)
#define TD_KEY_LAUNCH_WALK()                                                        \
    do {                                                                            \
        if (!mlp.l2_f16 || !mlp.l1_f16) TD_KEY_LAUNCH_WALK_L2(0);                   \
        else if (mlp.z_plain && !TD_ZPLAIN_OFF) TD_KEY_LAUNCH_WALK_F16(2);          \
        else TD_KEY_LAUNCH_WALK_F16(1);                                             \
    } while (0)
#else
#define TD_KEY_LAUNCH_WALK() TD_KEY_LAUNCH_WALK_L2(0)
#endif
    // (the unfused h2x key pass: STAGE tag 1, fp32 logits; its first layer follows TdEdgeMlp::l1_f16 like the fused kernel's)
#define TD_KEY_LAUNCH_H2X(GR)                                                                                                 \
    do {                                                                                                                  \
        if (mlp.l1_f16) {                                                                                                 \
            TD_LDS_ONCE((edge_key16_kernel<false, K16S_WAVES, 1, GR, true, 0, 1>), K16S_LDS_BYTES);                        \
            edge_key16_kernel<false, K16S_WAVES, 1, GR, true, 0, 1><<<dim3(grid16(a.count, K16S_WAVES)), dim3(K16S_WAVES * 64), K16S_LDS_BYTES, s>>>(a); \
        } else TD_KEY_LAUNCH(K16S_WAVES, 1, GR, true, K16S_LDS_BYTES);                                                     \
    } while (0)
    if (mlp.use_split) {                      // first layer on 16-bit pieces
        if (cptr && !h2x && cpn_p == 1) {
#define TD_KEY_LAUNCH_VIA(L2V) TD_KEY_LAUNCH_X2H(2, L2V)
            if (!mlp.l2_f16) TD_KEY_LAUNCH_VIA(0);
            else if (mlp.z_plain && !TD_ZPLAIN_OFF) TD_KEY_LAUNCH_VIA(2);
            else TD_KEY_LAUNCH_VIA(1);
#undef TD_KEY_LAUNCH_VIA
            if (lig_rows && lig_count > 0) {
                a.rows = lig_rows; a.count_ptr = nullptr; a.count = lig_count; a.trace = nullptr;
                TD_KEY_LAUNCH_WALK();
            }
        } else if (cptr) { if (h2x) TD_KEY_LAUNCH_H2X(1); else TD_KEY_LAUNCH_WALK(); }
        else if (h2x) TD_KEY_LAUNCH_H2X(0);
        else {
#define TD_KEY_LAUNCH_L2(L2V) TD_KEY_LAUNCH_X2H(0, L2V)
            if (!mlp.l2_f16) TD_KEY_LAUNCH_L2(0);
            else if (mlp.z_plain && !TD_ZPLAIN_OFF) TD_KEY_LAUNCH_L2(2);
            else TD_KEY_LAUNCH_L2(1);
#undef TD_KEY_LAUNCH_L2
        }
    } else {
        if (cptr) { if (h2x) TD_KEY_LAUNCH(K16_WAVES, 1, 1, false, K16_LDS_BYTES); else TD_KEY_LAUNCH(K16_WAVES, 0, 1, false, K16_LDS_BYTES); }
        else { if (h2x) TD_KEY_LAUNCH(K16_WAVES, 1, 0, false, K16_LDS_BYTES); else TD_KEY_LAUNCH(K16_WAVES, 0, 0, false, K16_LDS_BYTES); }
    }
#undef TD_KEY_LAUNCH
#undef TD_KEY_LAUNCH_WALK
#undef TD_KEY_LAUNCH_WALK_L2
#undef TD_KEY_LAUNCH_X2H
#undef TD_KEY_LAUNCH_H2X
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

// h2x value pass (xv MLP + coordinate update) on the listed ligand rows; reads alpha written by the h2x key pass.
int td_launch_edge_xv16(const TdEdgeMlp &mlp, const TdLayer &L, const float4 *x4_in, float4 *x4_out, const int32_t *nbr,
                        const float *P, const int32_t *rows, int64_t count, const float *alpha, hipStream_t s,
                        const int32_t *cptr) {
    if (count == 0) return TD_OK;
    Args16 a = {};
    a.x4 = x4_in; a.nbr = nbr; a.ew = nullptr; a.P = P; a.q = nullptr; a.rows = rows; a.count_ptr = nullptr; a.h = nullptr;
    a.alpha = const_cast<float *>(alpha); a.x4_out = x4_out; a.count = count; a.mlp = mlp; a.offsets = L.offsets;
    a.coeff = L.coeff; a.p_off = 2 * TD_H; a.cptr = cptr;
    const dim3 grid(grid16(count, XV16_WAVES)), block(XV16_WAVES * 64);
#define TD_XV_LAUNCH(CH, SP, BYTES)                                                            \
    do {                                                                                       \
        TD_LDS_ONCE((edge_key16_kernel<true, XV16_WAVES, 1, CH, SP>), BYTES);                  \
        edge_key16_kernel<true, XV16_WAVES, 1, CH, SP><<<grid, block, BYTES, s>>>(a);          \
    } while (0)
#define TD_XV_LAUNCH_F16(CH)                                                                   \
    do {                                                                                       \
        TD_LDS_ONCE((edge_key16_kernel<true, XV16_WAVES, 1, CH, true, 0, 1>), K16S_LDS_BYTES); \
        edge_key16_kernel<true, XV16_WAVES, 1, CH, true, 0, 1><<<grid, block, K16S_LDS_BYTES, s>>>(a); \
    } while (0)
    if (mlp.use_split && mlp.l1_f16) { if (cptr) TD_XV_LAUNCH_F16(1); else TD_XV_LAUNCH_F16(0); }
    else if (mlp.use_split) { if (cptr) TD_XV_LAUNCH(1, true, K16S_LDS_BYTES); else TD_XV_LAUNCH(0, true, K16S_LDS_BYTES); }
    else { if (cptr) TD_XV_LAUNCH(1, false, K16_LDS_BYTES); else TD_XV_LAUNCH(0, false, K16_LDS_BYTES); }
#undef TD_XV_LAUNCH
#undef TD_XV_LAUNCH_F16
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

// lig_rows / lig_count: the ligand rows among `rows` (all of them: every row list of a forward pass or sampling step holds
// the ligand atoms); nullptr / 0 for a list without ligand rows.  General graphs: cptr, the chunks per protein row and the
// chunks of all ligand rows together (the class split of the bf16 kernel is sized from chunk counts).
int td_launch_edge_value16(const TdEdgeMlp &mlp, const TdLayer &L, const float4 *x4, const int32_t *nbr, const float *P,
                           const int32_t *rows, const int32_t *count_ptr, int64_t count, float *h, const float *alpha,
                           const int32_t *lig_rows, int64_t lig_count, hipStream_t s, const int32_t *cptr, int cpn_p,
                           int64_t lig_chunks, const int32_t *mixed_count, float *out, const float *gate_m) {
    if (count == 0) return TD_OK;
    Args16 a = {};
    a.x4 = x4; a.nbr = nbr; a.ew = nullptr; a.P = P; a.q = nullptr; a.rows = rows; a.count_ptr = count_ptr; a.h = h;
    a.alpha = const_cast<float *>(alpha); a.x4_out = nullptr; a.count = count; a.mlp = mlp; a.offsets = L.offsets;
    a.coeff = L.coeff; a.p_off = 2 * TD_H; a.cptr = cptr; a.cpn_p = cpn_p > 0 ? cpn_p : 1; a.lig_chunks = lig_chunks;
    a.mixed_count = mixed_count;
    a.out = out;
    a.gate_m = cptr ? nullptr : gate_m;          // (ew_net_type 'm' is accepted on the default graph only)
    int G = grid16(count, V16_WAVES);
    const dim3 block(V16_WAVES * 64);
    a.trace = wg_trace_slot(1);
    a.deal = mlp.deal_rows;
    // (the 8-wave kernel, bf16-class first layer: CH = chunk walk, GM = gate from the value vector, L2V / TdEdgeMlp::l1_f16 = the layers' variants)
#define TD_V16_LAUNCH(CH, GM, L2V)                                                                            \
    do {                                                                                                     \
        if (mlp.l1_f16) {                                                                                    \
            TD_LDS_ONCE((edge_value16_kernel<true, CH, GM, L2V, 1>), V16S_LDS_BYTES);                        \
            edge_value16_kernel<true, CH, GM, L2V, 1><<<dim3(G), block, V16S_LDS_BYTES, s>>>(a);             \
        } else {                                                                                             \
            TD_LDS_ONCE((edge_value16_kernel<true, CH, GM, L2V, 0>), V16S_LDS_BYTES);                        \
            edge_value16_kernel<true, CH, GM, L2V, 0><<<dim3(G), block, V16S_LDS_BYTES, s>>>(a);             \
        }                                                                                                    \
    } while (0)
#define TD_V16T_LAUNCH_ANY(L2V, VIAV)                                                                         \
    do {                                                                                                     \
        if (mlp.l1_f16) {                                                                                    \
            TD_LDS_ONCE((edge_value16t_kernel<L2V, VIAV, 1>), V16T_LDS_BYTES);                               \
            edge_value16t_kernel<L2V, VIAV, 1><<<dim3(Gt), dim3(V16T_WAVES * 64), V16T_LDS_BYTES, s>>>(a);   \
        } else {                                                                                             \
            TD_LDS_ONCE((edge_value16t_kernel<L2V, VIAV, 0>), V16T_LDS_BYTES);                               \
            edge_value16t_kernel<L2V, VIAV, 0><<<dim3(Gt), dim3(V16T_WAVES * 64), V16T_LDS_BYTES, s>>>(a);   \
        }                                                                                                    \
    } while (0)
#define TD_V16_LAUNCH_WALK_L2(L2V) TD_V16_LAUNCH(true, false, L2V)
#define TD_V16_LAUNCH_WALK()                                                        \
    do {                                                                            \
        if (!mlp.l2_f16) TD_V16_LAUNCH_WALK_L2(0);                                  \
        else if (mlp.z_plain && !TD_ZPLAIN_OFF) TD_V16_LAUNCH_WALK_L2(2);           \
        else TD_V16_LAUNCH_WALK_L2(1);                                              \
    } while (0)
    if (mlp.use_split) {
        a.lig_rows = lig_rows; a.lig_count = lig_rows ? lig_count : 0;
        if (G < 2 && a.lig_count > 0) G = 2;       // a workgroup for each destination class
        if (cptr && a.cpn_p == 1) {
            // `hybrid`: the protein rows (one chunk, plain k-NN) through the 12-wave kernel of the default graph, whatever edge_row_dealing says;
            // the ligand rows (several chunks) in a second, chunk-walking launch over the ligand row list
            int Gt = grid16(count, V16T_WAVES);
            a.lig_rows = nullptr; a.lig_count = 0;
#define TD_V16T_LAUNCH_VIA(L2V) TD_V16T_LAUNCH_ANY(L2V, true)
            if (!mlp.l2_f16) TD_V16T_LAUNCH_VIA(0);
            else if (mlp.z_plain && !TD_ZPLAIN_OFF) TD_V16T_LAUNCH_VIA(2);
            else TD_V16T_LAUNCH_VIA(1);
#undef TD_V16T_LAUNCH_VIA
            if (lig_rows && lig_count > 0) {
                a.rows = lig_rows; a.count_ptr = nullptr; a.count = lig_count; a.lig_rows = lig_rows; a.lig_count = lig_count;
                a.mixed_count = nullptr; a.trace = nullptr;
                G = grid16(lig_count, V16_WAVES);
                TD_V16_LAUNCH_WALK();
            }
        } else if (cptr) {
            TD_V16_LAUNCH_WALK();
        } else if (gate_m) {
            TD_V16_LAUNCH(false, true, 0);
        } else if (a.deal == 2) {          // (the default) rows through the LDS ticket: the 12-wave kernel
            int Gt = grid16(count, V16T_WAVES);
            if (Gt < 2 && a.lig_count > 0) Gt = 2;
#define TD_V16T_LAUNCH(L2V) TD_V16T_LAUNCH_ANY(L2V, false)
            if (!mlp.l2_f16) TD_V16T_LAUNCH(0);
            else if (mlp.z_plain && !TD_ZPLAIN_OFF) TD_V16T_LAUNCH(2);
            else TD_V16T_LAUNCH(1);
#undef TD_V16T_LAUNCH
        } else if (mlp.l2_f16 && mlp.z_plain && !TD_ZPLAIN_OFF) {
            TD_V16_LAUNCH(false, false, 2);
        } else if (mlp.l2_f16) {
            TD_V16_LAUNCH(false, false, 1);
        } else {
            TD_V16_LAUNCH(false, false, 0);
        }
    } else if (cptr) {
        TD_LDS_ONCE((edge_value16_kernel<false, true>), V16_LDS_BYTES);
        edge_value16_kernel<false, true><<<dim3(G), block, V16_LDS_BYTES, s>>>(a);
    } else if (gate_m) {
        TD_LDS_ONCE((edge_value16_kernel<false, false, true>), V16_LDS_BYTES);
        edge_value16_kernel<false, false, true><<<dim3(G), block, V16_LDS_BYTES, s>>>(a);
    } else {
        TD_LDS_ONCE((edge_value16_kernel<false, false>), V16_LDS_BYTES);
        edge_value16_kernel<false, false><<<dim3(G), block, V16_LDS_BYTES, s>>>(a);
    }
#undef TD_V16_LAUNCH_WALK
#undef TD_V16_LAUNCH_WALK_L2
#undef TD_V16T_LAUNCH_ANY
#undef TD_V16_LAUNCH
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

// Fused h2x stage (keys + softmax + xv + coordinate update) on the listed ligand rows.
int td_launch_edge_h2x16(const TdEdgeMlp &mlp_k, const TdEdgeMlp &mlp_v, const TdLayer &L, const float4 *x4_in, float4 *x4_out,
                         const int32_t *nbr, const float *ew, const float *P, const float *q, const int32_t *rows,
                         int64_t count, hipStream_t s, const int32_t *cptr, float *alpha) {
    if (count == 0) return TD_OK;
    ArgsH2x ar = {};
    Args16 &a = ar.a;
    a.x4 = x4_in; a.nbr = nbr; a.ew = ew; a.P = P; a.q = q; a.rows = rows; a.count_ptr = nullptr; a.h = nullptr;
    a.alpha = alpha; a.x4_out = x4_out; a.count = count; a.mlp = mlp_k; a.offsets = L.offsets; a.coeff = L.coeff; a.p_off = 0;
    a.cptr = cptr;
    ar.mlp_v = mlp_v;
    a.trace = cptr ? nullptr : wg_trace_slot(2);
    const dim3 grid(grid16(count, H2X16_WAVES)), block(H2X16_WAVES * 64);
    const bool f16_first = mlp_k.l1_f16 && mlp_v.l1_f16;          // both halves' first layers on f16 piece pairs
    if (cptr && f16_first) {           // general graphs: the chunk-walking form (16-bit first layer; the caller checks use_split)
        TD_LDS_ONCE((edge_h2x16_chunked_kernel<1>), h2x16_lds_bytes<true>());
        edge_h2x16_chunked_kernel<1><<<grid, block, h2x16_lds_bytes<true>(), s>>>(ar);
    } else if (cptr) {
        TD_LDS_ONCE((edge_h2x16_chunked_kernel<0>), h2x16_lds_bytes<true>());
        edge_h2x16_chunked_kernel<0><<<grid, block, h2x16_lds_bytes<true>(), s>>>(ar);
    } else if (mlp_k.use_split && mlp_v.use_split && f16_first) {
        TD_LDS_ONCE((edge_h2x16_kernel<true, 1>), h2x16_lds_bytes<true>());
        edge_h2x16_kernel<true, 1><<<grid, block, h2x16_lds_bytes<true>(), s>>>(ar);
    } else if (mlp_k.use_split && mlp_v.use_split) {
        TD_LDS_ONCE((edge_h2x16_kernel<true>), h2x16_lds_bytes<true>());
        edge_h2x16_kernel<true><<<grid, block, h2x16_lds_bytes<true>(), s>>>(ar);
    } else {
        TD_LDS_ONCE((edge_h2x16_kernel<false>), h2x16_lds_bytes<false>());
        edge_h2x16_kernel<false><<<grid, block, h2x16_lds_bytes<false>(), s>>>(ar);
    }
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

// ---- edge gate ------------------------------------------------------------------------------------------------------
int td_launch_gate16(const TdGate &g, const float4 *x4, const int32_t *nbr, int64_t N, const int32_t *rows,
                     const int32_t *count_ptr, float *ew, hipStream_t s, const int32_t *chunk_node) {
    if (N == 0) return TD_OK;
    TD_LDS_ONCE((edge_gate16_kernel), G16_LDS_BYTES);
    int64_t G = (N + G16_WAVES - 1) / G16_WAVES;
    if (G > TD_GATE_WGS) G = TD_GATE_WGS;              // workgroups per CU x 256
    edge_gate16_kernel<<<dim3((unsigned)G), dim3(G16_WAVES * 64), G16_LDS_BYTES, s>>>(g, x4, nbr, N, rows, count_ptr, chunk_node, ew);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

