// C ABI of libtargetdiff_hip.so, part 2 of 5: the weight blob -- LayerNorm folding, re-packing into the kernels' fragment orders,
// td_model_create / destroy / options.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "td_device.h"
#include "td_internal.h"
#include "td_api.h"

using namespace tdapi;

// ------------------------------------------------------------------------------------------ blob layout
namespace tdapi {



// The LayerNorm of an edge MLP (Linear -> LayerNorm -> ReLU -> Linear, models/common.py:60-80) folded into its two Linears at pack
// time.  With c = pre-activation minus its mean over the hidden units, sigma = sqrt(var + eps), s_n = sign(gamma_n), a_n = |gamma_n|:
//     relu(gamma_n c_n / sigma + beta_n) = (a_n / sigma) relu(s_n c_n + (beta_n / a_n) sigma)
//  * every column of the first Linear (and its bias) is centred over the hidden units and row n multiplied by s_n: the matrix product
//    IS s_n c_n -- the radial/type table, the node-side projections P_i / P_j and the bias all come from these rows, so the kernels
//    compute neither a mean nor a subtraction;
//  * beta_n / a_n replaces beta, a_n goes into column n of the second Linear, and 1 / sigma (one number per edge) multiplies the
//    second layer's per-edge result in the consumer (logit, xv, or the attention weight of the aggregation).
// Exact algebra.
// Round 5 -- the ReLU as the FMA's own output clamp.  Dividing by sigma M instead of carrying sigma along,
//     relu(s_n c_n / sigma + beta_n / a_n) / M = clamp_[0,1](s_n c_n (1 / (sigma M)) + beta_n / (a_n M)),
// holds whenever the left side never exceeds 1: |c_n| <= sqrt(hid) sigma (the c_n are centred and sigma^2 >= their mean square), so
// M = (sqrt(hid) + max_n beta_n / a_n) (1 + 2^-10) does.  v_fma_f32 has a free clamp-to-[0, 1] output modifier: ONE instruction per hidden
// value where the round-4 form took an FMA and a max, no per-edge 1 / sigma for the consumers to apply (M a_n goes into column n of the
// second Linear instead of a_n), and the kernels keep s_e = 1 / (sigma_e M) per edge only as the FMA's multiplier.
// Round 6 -- dead units.  beta_n / a_n enters M, and the kernels evaluate rsqrt(sum c^2 * M^2 / hid + eps M^2): a unit with a_n -> 0 and
// beta_n > 0 used to push M to ~1e19 and the radicand past FLT_MAX (rsqrt(inf) = 0: every activation of the edge silently collapsed to
// its bias term).  A unit whose |gamma_n| is below 2^-30 of the MLP's largest |gamma| is the constant relu(beta_n) to 1e-8 of the live
// units' scale (|c_n / sigma| <= sqrt(hid)): its constant goes into the second Linear's bias (b3 += w3[:, n] relu(beta_n)), its column of
// the second Linear is zero, and its FMA constant is -1 so that the clamp gives exactly 0 (|c_n / (sigma M)| < 1).  Its row of the first
// Linear stays: LayerNorm's mean and variance run over all hidden units.  Dead units stay out of M, so M <= sqrt(hid) + 2^30 max beta /
// max |gamma|; td_model_create refuses a model whose M^2 could still overflow the radicand (overflow_risk).
struct FoldedMlp {
    std::vector<float> w0, b0, g, b, w3, b3v;
    const float *b3;
    float ln_c1 = 1.f / 128.f, ln_c2 = 1e-5f;        // the kernels' variance constants (below)
    int dead_units = 0;
    bool overflow_risk = false;                      // M^2 * (pre-LayerNorm variance of 1e8) would leave the fp32 range
    // first_scale (round 6, late): the whole first Linear -- every column, the bias -- times 2^first_scale_exp, and ln_c2 times its
    // square: the normalised activations do not change (a power of two goes through every product, sum and the rsqrt exactly: the kernels
    // that do not care produce the same bits as without it), and the per-edge columns' largest weight sits in [2^13, 2^14) -- where the
    // f16 piece-pair table of the key pass's first layer (pack_h2_table) needs it, since its products accumulate onto the gathered node
    // projections, which therefore have to be in the same units
    int first_scale_exp = 0;
    bool first_scaled = false;
    FoldedMlp(const MlpSrc &m, int in, int hid, int out, bool first_scale = false)
        : w0((size_t)hid * in), b0(hid), g(hid), b(hid), w3((size_t)out * hid), b3v(m.b3, m.b3 + out), b3(nullptr) {
        std::vector<float> sg(hid);
        std::vector<char> dead(hid, 0);
        double bmax = 0.0, amax = 0.0;
        for (int n = 0; n < hid; ++n) amax = std::max(amax, (double)fabsf(m.g[n]));
        const double floor_a = amax * (1.0 / 1073741824.0);          // 2^-30 of the largest |gamma|; amax = 0: every unit is dead
        for (int n = 0; n < hid; ++n) {
            const double a = (double)fabsf(m.g[n]);
            sg[n] = m.g[n] < 0.f ? -1.f : 1.f;
            if (!(a > floor_a) || !(a > 1e-30)) {
                dead[n] = 1;
                ++dead_units;
                sg[n] = 1.f;
                g[n] = 0.f;
                b[n] = -1.f;
                const double r = m.b[n] > 0.f ? (double)m.b[n] : 0.0;
                for (int o = 0; o < out; ++o) b3v[o] = (float)((double)b3v[o] + (double)m.w3[(size_t)o * hid + n] * r);
                continue;
            }
            g[n] = (float)a;
            b[n] = (float)((double)m.b[n] / a);
            if ((double)b[n] > bmax) bmax = (double)b[n];
        }
        const double M = (sqrt((double)hid) + bmax) * (1.0 + 1.0 / 1024.0);
        overflow_risk = !(M * M * 1e8 < 3.0e38);
        for (int n = 0; n < hid; ++n) {
            if (dead[n]) continue;
            b[n] = (float)((double)b[n] / M);
            g[n] = (float)((double)g[n] * M);          // what column n of the second Linear carries
        }
        ln_c1 = (float)(M * M / hid);                  // 1 / (sigma M) = rsqrt(sum_n c_n^2 * ln_c1 + ln_c2)
        ln_c2 = (float)(1e-5 * M * M);
        for (int k = 0; k < in; ++k) {
            double mean = 0.0;
            for (int n = 0; n < hid; ++n) mean += (double)m.w0[(size_t)n * in + k];
            mean /= hid;
            for (int n = 0; n < hid; ++n) w0[(size_t)n * in + k] = (float)((double)sg[n] * ((double)m.w0[(size_t)n * in + k] - mean));
        }
        double mb = 0.0;
        for (int n = 0; n < hid; ++n) mb += (double)m.b0[n];
        mb /= hid;
        for (int n = 0; n < hid; ++n) b0[n] = (float)((double)sg[n] * ((double)m.b0[n] - mb));
        for (int o = 0; o < out; ++o)
            for (int n = 0; n < hid; ++n) w3[(size_t)o * hid + n] = m.w3[(size_t)o * hid + n] * g[n];
        b3 = b3v.data();
        if (first_scale) {
            const int edge_cols = in >= 2 * hid ? in - 2 * hid : in;          // [edge type | r_feat | h_i | h_j]
            float wmax = 0.f;
            for (int n = 0; n < hid; ++n)
                for (int k = 0; k < edge_cols; ++k) wmax = std::max(wmax, fabsf(w0[(size_t)n * in + k]));
            int e = (wmax > 0.f && std::isfinite(wmax)) ? 13 - ilogbf(wmax) : 0;
            e = e > 60 ? 60 : (e < -60 ? -60 : e);
            while (e > -60 && !(M * M * 1e8 * ldexp(1.0, 2 * e) < 3.0e38)) --e;       // the LayerNorm's radicand stays in range
            first_scale_exp = e;
            first_scaled = true;
            for (float &x : w0) x = ldexpf(x, e);
            for (float &x : b0) x = ldexpf(x, e);
            ln_c2 = ldexpf(ln_c2, 2 * e);
        }
    }
    MlpSrc src() const { return MlpSrc{w0.data(), b0.data(), g.data(), b.data(), w3.data(), b3}; }
};

// B fragments of a 128-deep GEMM with all 4 N tiles per lane: dst[(s*64 + lane)*4 + t] = W[32t + c][col0 + kmap(s, hi)]
size_t pack_B128(Packer &pk, const float *W, int ld, int col0) {
    size_t off = pk.alloc((size_t)TD_KSTEPS * 64 * 4);
    float *d = pk.data.data() + off;
    for (int s = 0; s < TD_KSTEPS; ++s)
        for (int lane = 0; lane < 64; ++lane)
            for (int t = 0; t < 4; ++t)
                d[((size_t)s * 64 + lane) * 4 + t] = W[(size_t)(32 * t + (lane & 31)) * ld + col0 + td_kmap(s, lane >> 5)];
    return off;
}

// bf16 round-to-nearest-even of an fp32 value, returned as the upper 16 bits
inline uint32_t bf16_rne(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint32_t r = u + 0x7fffu + ((u >> 16) & 1u);
    return r >> 16;
}
inline float bf16_to_f32(uint32_t b) {
    const uint32_t u = b << 16;
    float x;
    memcpy(&x, &u, 4);
    return x;
}
// B operand of the split node-projection GEMM, in the order node_proj_split_kernel streams it: 4 chunks per matrix =
// [half 2][k-chunk 2], each [k-step 4][piece 3][tile 2][lane 64] x 8 bf16 (two per 32-bit word, slot j of lane half hi =
// k index 16s + 8(j >> 2) + 4hi + (j & 3) with s = 4 * k-chunk + k-step; N tile = 2 * half + tile); piece p of W = bf16 of
// the residual left by pieces 0 .. p-1 (exact).  Lane-minor: a wave's 16-byte reads of one fragment are consecutive in LDS.
size_t pack_B128_split(Packer &pk, const float *W, int ld, int col0) {
    size_t off = pk.alloc((size_t)8 * 3 * 64 * 4 * 4);
    uint32_t *d = reinterpret_cast<uint32_t *>(pk.data.data() + off);
    for (int s = 0; s < 8; ++s)
        for (int lane = 0; lane < 64; ++lane)
            for (int t = 0; t < 4; ++t) {
                uint32_t pieces[3][8];
                for (int j = 0; j < 8; ++j) {
                    const int k = 16 * s + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3);
                    float r = W[(size_t)(32 * t + (lane & 31)) * ld + col0 + k];
                    for (int p = 0; p < 3; ++p) {
                        pieces[p][j] = bf16_rne(r);
                        r -= bf16_to_f32(pieces[p][j]);
                    }
                }
                for (int p = 0; p < 3; ++p)
                    for (int w = 0; w < 4; ++w)
                    {
                        const size_t chunk = (size_t)(t >> 1) * 2 + (s >> 2), ss = s & 3, tt = t & 1;
                        d[(((((chunk * 4 + ss) * 3 + p) * 2 + tt) * 64 + lane)) * 4 + w] = pieces[p][2 * w] | (pieces[p][2 * w + 1] << 16);
                    }
            }
    return off;
}


// The same 21-wide first layer (k < 20 Gaussians, k = 20 the edge-type column) for the K-PACKED products of
// td_first_layer_split16<PK>: the six bf16 piece products a_p b_q (p + q <= 4; a = this table, b = the per-edge inputs) of the 21 inputs
// are 123 (p, q, k) slot pairs -- k = 20 has b = 1.0 exactly, i.e. three -- and fit the 4 x 32 K slots of FOUR v_mfma_f32_16x16x32_bf16
// (the plain layout spends six, one per product, with 11 of every 32 K slots empty).  Lane group g owns the Gaussians 5g .. 5g+4
// (k0 .. k4).  Per lane (hidden lo, group g) and hidden block, in halves of four bf16 (slot 2w = low half of word w):
//   a1, a2, a3 = pieces 1 .. 3 of k0 .. k3;   H6 = (a1, a2, a1, a2)[k4];   H7 = (a3[k4], a1[k4], T, T')
//   t0: (a1 | a2) x (b1 | b2)     t1: (a1 | a2) x (b2 | b1)     t2: (a3 | H6) x (b1 | p1 p1 p2 p2 [k4])     t3: (a1 | H7) x (b3 | p1 p3 [k4], C)
// with the type column's three pieces in the spare slot pair: (T, T') = (t1, t2) in group 0, (t3, 0) in group 1, 0 elsewhere; C = (1, 1) /
// (1, 0) / 0.  One (dst class, source class) table: QA[hb][lane] = (a1 | a2), QB[hb][lane] = (a3 | H6), H7[hb][lane] (8 bytes),
// QC[hb][lane] = (a1 | H7): a kernel stages QA, QB and either H7 (40 bytes per lane and hidden block: t3's operand is two 8-byte
// reads, the a1 half of QA and H7) or QC (48 bytes: three 16-byte reads).  w(n, k): the folded first-layer weight, k = 20 = type column.
constexpr size_t PK4_QA = 0, PK4_QB = (size_t)8 * 64 * 4, PK4_H7 = 2 * PK4_QB, PK4_QC = PK4_H7 + (size_t)8 * 64 * 2;
constexpr size_t PK4_WORDS = PK4_QC + (size_t)8 * 64 * 4;        // 32-bit words of one (dst class, source class) table: 28 KiB
template <class F>
void pack_pk4_table(uint32_t *dst, F w) {
    for (int hb = 0; hb < 8; ++hb)
        for (int lane = 0; lane < 64; ++lane) {
            const int lo = lane & 15, g = lane >> 4, n = 16 * hb + lo;
            uint32_t a[3][6];                          // pieces of k0 .. k4 and of the type column
            for (int i = 0; i < 6; ++i) {
                float r = w(n, i < 5 ? 5 * g + i : TD_NG);
                for (int p = 0; p < 3; ++p) {
                    a[p][i] = bf16_rne(r);
                    r -= bf16_to_f32(a[p][i]);
                }
            }
            auto pair = [](uint32_t lo16, uint32_t hi16) { return lo16 | (hi16 << 16); };
            const size_t e = (size_t)hb * 64 + lane;
            uint32_t *QA = dst + PK4_QA + e * 4, *QB = dst + PK4_QB + e * 4, *H7 = dst + PK4_H7 + e * 2, *QC = dst + PK4_QC + e * 4;
            QA[0] = pair(a[0][0], a[0][1]); QA[1] = pair(a[0][2], a[0][3]); QA[2] = pair(a[1][0], a[1][1]); QA[3] = pair(a[1][2], a[1][3]);
            QB[0] = pair(a[2][0], a[2][1]); QB[1] = pair(a[2][2], a[2][3]); QB[2] = pair(a[0][4], a[1][4]); QB[3] = pair(a[0][4], a[1][4]);
            H7[0] = pair(a[2][4], a[0][4]);
            H7[1] = g == 0 ? pair(a[0][5], a[1][5]) : (g == 1 ? pair(a[2][5], 0u) : 0u);
            QC[0] = QA[0]; QC[1] = QA[1]; QC[2] = H7[0]; QC[3] = H7[1];
        }
}

// The same 21-wide first layer on f16 piece PAIRS (round 6, late; the x2h key pass): w = a1 + a2 (a1 = f16 of w, a2 = f16 of the exact
// residual: 22 bits), an input b = b1 + b2 likewise, and the three products a1 b1, a2 b1, a1 b2 of the lane group's five Gaussians are 15
// K slots of the 16 the group has in TWO v_mfma_f32_16x16x32_f16; the sixteenth carries a piece of the type column's weight (piece 1 in
// group 0, piece 2 in group 1; its input is exactly 1).  Per lane (hidden lo, group g) and hidden block, halves low first:
//   I0 = (a1_0 a1_1 | a1_2 a1_3 | a2_0 a2_1 | a2_2 a2_3)   x   (b1_0 b1_1 | b1_2 b1_3 | b1_0 b1_1 | b1_2 b1_3)
//   I1 = (a1_0 a1_1 | a1_2 a1_3 | a1_4 a1_4 | a2_4 T_g )   x   (b2_0 b2_1 | b2_2 b2_3 | b1_4 b2_4 | b1_4 1   )
// One (dst class, source class) table: I0[hb 8][lane 64], I1[hb][lane], 16 bytes each = 16 KiB.  w(n, k): the folded, SCALED weight.
constexpr size_t H2_WORDS = (size_t)2 * 8 * 64 * 4;
template <class F>
void pack_h2_table(uint32_t *dst, F w) {
    auto half_bits = [](float x) -> uint32_t {
        const _Float16 h = (_Float16)x;             // round to nearest even; subnormals kept
        uint16_t u;
        memcpy(&u, &h, 2);
        return u;
    };
    auto half_value = [](uint32_t u) -> float {
        const uint16_t v = (uint16_t)u;
        _Float16 h;
        memcpy(&h, &v, 2);
        return (float)h;
    };
    for (int hb = 0; hb < 8; ++hb)
        for (int lane = 0; lane < 64; ++lane) {
            const int lo = lane & 15, g = lane >> 4, n = 16 * hb + lo;
            uint32_t a1[6], a2[6];                     // pieces of k0 .. k4 and of the type column
            for (int i = 0; i < 6; ++i) {
                const float x = w(n, i < 5 ? 5 * g + i : TD_NG);
                a1[i] = half_bits(x);
                a2[i] = half_bits(x - half_value(a1[i]));
            }
            auto pair = [](uint32_t lo16, uint32_t hi16) { return lo16 | (hi16 << 16); };
            uint32_t *I0 = dst + ((size_t)hb * 64 + lane) * 4, *I1 = dst + (size_t)8 * 64 * 4 + ((size_t)hb * 64 + lane) * 4;
            I0[0] = pair(a1[0], a1[1]); I0[1] = pair(a1[2], a1[3]); I0[2] = pair(a2[0], a2[1]); I0[3] = pair(a2[2], a2[3]);
            I1[0] = pair(a1[0], a1[1]); I1[1] = pair(a1[2], a1[3]); I1[2] = pair(a1[4], a1[4]);
            I1[3] = pair(a2[4], g == 0 ? a1[5] : (g == 1 ? a2[5] : 0u));
        }
}

size_t pack_vec(Packer &pk, const float *v, size_t n, size_t padded) {
    size_t off = pk.alloc(padded ? padded : n);
    if (v) memcpy(pk.data.data() + off, v, n * sizeof(float));
    return off;
}

struct EdgeOff { size_t R, gamma, beta, W2, b2, Walt, R16, Walt16, R16q, R16h = 0; float ln_c1, ln_c2, w2_bound; bool z_plain; };

EdgeOff pack_edge_mlp(Packer &pk, const FoldedMlp &fm, int in_dim, int out_dim, int alt) {
    const MlpSrc m = fm.src();
    EdgeOff o;
    o.ln_c1 = fm.ln_c1; o.ln_c2 = fm.ln_c2;
    // ln_c1 = M^2 / hid: the f16 second layer takes the pieces of z'' ~ 1 / M unscaled while M <= 32 (z'' of a typical unit then sits at
    // 2^-5 or above: 20 bits over the f16 subnormal floor; beyond that the kernels scale by 2^15 first -- td_ln_relu16_pairs_*, edge16.hip)
    o.z_plain = (double)fm.ln_c1 * TD_H <= 32.0 * 32.0;
    {
        float wmax = 0.f;
        for (size_t t = 0; t < (size_t)out_dim * TD_H; ++t) wmax = std::max(wmax, fabsf(m.w3[t]));
        o.w2_bound = 8.0f * wmax;
    }
    o.Walt = 0;
    o.Walt16 = 0;
    // first layer radial / type table: [cls][slot][kstep][lane][ntile]
    o.R = pk.alloc((size_t)2 * 2 * TD_SLOT_STEPS * 64 * 4);
    float *d = pk.data.data() + o.R;
    for (int cls = 0; cls < 2; ++cls)
        for (int sl = 0; sl < 2; ++sl) {
            // edge type (models/uni_transformer.py:292-297): 0 l<-l, 1 src lig/dst prot, 2 src prot/dst lig, 3 p<-p
            const int type = cls == 0 ? (sl == 0 ? 0 : 2) : (sl == 0 ? 1 : 3);
            for (int s = 0; s < TD_SLOT_STEPS; ++s)
                for (int lane = 0; lane < 64; ++lane)
                    for (int t = 0; t < 4; ++t) {
                        const int kk = td_kmap(s, lane >> 5), n = 32 * t + (lane & 31);
                        float v = 0.f;
                        if (kk < TD_NG) v = m.w0[(size_t)n * in_dim + 4 + TD_NG * type + kk];   // r_feat, type-major
                        else if (kk == TD_NG) v = m.w0[(size_t)n * in_dim + type];              // one-hot edge type column
                        d[((((size_t)cls * 2 + sl) * TD_SLOT_STEPS + s) * 64 + lane) * 4 + t] = v;
                    }
        }
    // the same table for 16x16x4 tiles: [cls][slot][step][lane][hb], lane = (lo = hidden_local, g = k index in step)
    o.R16 = pk.alloc((size_t)2 * 2 * 6 * 64 * 8);
    {
        float *d16 = pk.data.data() + o.R16;
        for (int cls = 0; cls < 2; ++cls)
            for (int sl = 0; sl < 2; ++sl) {
                const int type = cls == 0 ? (sl == 0 ? 0 : 2) : (sl == 0 ? 1 : 3);
                for (int st = 0; st < 6; ++st)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int hb = 0; hb < 8; ++hb) {
                            const int kk = 4 * st + (lane >> 4), n = 16 * hb + (lane & 15);
                            float v = 0.f;
                            if (kk < TD_NG) v = m.w0[(size_t)n * in_dim + 4 + TD_NG * type + kk];
                            else if (kk == TD_NG) v = m.w0[(size_t)n * in_dim + type];
                            d16[((((size_t)cls * 2 + sl) * 6 + st) * 64 + lane) * 8 + hb] = v;
                        }
            }
    }
    // the same table as exact bf16 piece triples, K-packed (pack_pk4_table): [dst class][source class] x 28 KiB
    o.R16q = pk.alloc((size_t)2 * 2 * PK4_WORDS);
    for (int cls = 0; cls < 2; ++cls)
        for (int sl = 0; sl < 2; ++sl) {
            const int type = cls == 0 ? (sl == 0 ? 0 : 2) : (sl == 0 ? 1 : 3);
            pack_pk4_table(reinterpret_cast<uint32_t *>(pk.data.data() + o.R16q) + ((size_t)cls * 2 + sl) * PK4_WORDS, [&](int n, int k) {
                return k < TD_NG ? m.w0[(size_t)n * in_dim + 4 + TD_NG * type + k] : m.w0[(size_t)n * in_dim + type];
            });
        }
    // ... and, for the key MLP of an x2h stage, as f16 piece pairs (pack_h2_table): [dst class][source class] x 16 KiB
    if (fm.first_scaled) {
        o.R16h = pk.alloc((size_t)2 * 2 * H2_WORDS);
        for (int cls = 0; cls < 2; ++cls)
            for (int sl = 0; sl < 2; ++sl) {
                const int type = cls == 0 ? (sl == 0 ? 0 : 2) : (sl == 0 ? 1 : 3);
                pack_h2_table(reinterpret_cast<uint32_t *>(pk.data.data() + o.R16h) + ((size_t)cls * 2 + sl) * H2_WORDS, [&](int n, int k) {
                    return k < TD_NG ? m.w0[(size_t)n * in_dim + 4 + TD_NG * type + k] : m.w0[(size_t)n * in_dim + type];
                });
            }
    }
    o.gamma = pack_vec(pk, m.g, TD_H);
    o.beta = pack_vec(pk, m.b, TD_H);
    if (out_dim == TD_H) {
        o.W2 = pack_B128(pk, m.w3, TD_H, 0);
        o.b2 = pack_vec(pk, m.b3, TD_H);
    } else {   // xv: [16][128] -> one N tile padded 16 -> 32 columns
        o.W2 = pk.alloc((size_t)TD_KSTEPS * 64);
        float *q = pk.data.data() + o.W2;
        for (int s = 0; s < TD_KSTEPS; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int cc = lane & 31;
                q[(size_t)s * 64 + lane] = cc < out_dim ? m.w3[(size_t)cc * TD_H + td_kmap(s, lane >> 5)] : 0.f;
            }
        o.b2 = pack_vec(pk, m.b3, out_dim, TD_HEADS);
    }
    if (alt == 0) {          // h2x value MLP: A operand of the 16x16x4 xv product, W2xv16[hb][r][lane]
        o.Walt16 = pk.alloc((size_t)8 * 4 * 64);
        float *q16 = pk.data.data() + o.Walt16;
        for (int hb = 0; hb < 8; ++hb)
            for (int r = 0; r < 4; ++r)
                for (int lane = 0; lane < 64; ++lane)
                    q16[((size_t)hb * 4 + r) * 64 + lane] = m.w3[(size_t)(lane & 15) * TD_H + 16 * hb + 4 * (lane >> 4) + r];
    }
    if (alt == 1) {          // key MLP: per-head slices of W2 in the order the U_i build consumes them
        o.Walt = pk.alloc((size_t)4 * 16 * 2 * 2 * 16 * 4);
        float *q = pk.data.data() + o.Walt;
        for (int t = 0; t < 4; ++t)
            for (int r = 0; r < 16; ++r)
                for (int jq = 0; jq < 2; ++jq)
                    for (int hi = 0; hi < 2; ++hi)
                        for (int c = 0; c < 16; ++c)
                            for (int jj = 0; jj < 4; ++jj) {
                                const int n = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi;      // hidden unit of C-layout row r
                                q[((((((size_t)t * 16 + r) * 2 + jq) * 2 + hi) * 16 + c) * 4) + jj] =
                                    m.w3[(size_t)(8 * c + 4 * jq + jj) * TD_H + n];
                            }
        o.Walt16 = pk.alloc((size_t)8 * 4 * 2 * 64 * 4);
        float *q16 = pk.data.data() + o.Walt16;
        for (int hb = 0; hb < 8; ++hb)
            for (int r = 0; r < 4; ++r)
                for (int jq = 0; jq < 2; ++jq)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int jj = 0; jj < 4; ++jj) {
                            const int head = lane & 15, k = 16 * hb + 4 * (lane >> 4) + r;
                            q16[(((((size_t)hb * 4 + r) * 2 + jq) * 64 + lane) * 4) + jj] =
                                m.w3[(size_t)(8 * head + 4 * jq + jj) * TD_H + k];
                        }
    } else if (alt == 2) {   // value MLP of x2h: Wt[d][k/4][head][k%4] = W2v[8 head + d][k] (td_value_out16: a lane reads 16 bytes at 16 lane + 8192 d + 1024 hb)
        o.Walt = pk.alloc((size_t)32 * TD_H * 4);
        float *q = pk.data.data() + o.Walt;
        for (int d = 0; d < 8; ++d)
            for (int kq = 0; kq < 32; ++kq)
                for (int head = 0; head < TD_HEADS; ++head)
                    for (int kk = 0; kk < 4; ++kk)
                        q[((((size_t)d * 32 + kq) * TD_HEADS + head) * 4) + kk] = m.w3[(size_t)(8 * head + d) * TD_H + 4 * kq + kk];
    }
    return o;
}

struct NodeOff { size_t projB, projBias, qGamma, qBeta, q3B, q3Bias, projB3, q3B3; };

NodeOff pack_node_stage(Packer &pk, const MlpSrc &k, const MlpSrc &v, const MlpSrc &q, int in_dim) {
    NodeOff o;
    const int hi_col = in_dim - 2 * TD_H, hj_col = in_dim - TD_H;       // [.. | h_i | h_j]
    o.projB = pack_B128(pk, k.w0, in_dim, hi_col);
    pack_B128(pk, k.w0, in_dim, hj_col);                                // consecutive 64-float-aligned blocks
    pack_B128(pk, v.w0, in_dim, hi_col);
    pack_B128(pk, v.w0, in_dim, hj_col);
    pack_B128(pk, q.w0, TD_H, 0);
    o.projBias = pk.alloc(5 * TD_H);
    memcpy(pk.data.data() + o.projBias + 0 * TD_H, k.b0, TD_H * sizeof(float));
    memcpy(pk.data.data() + o.projBias + 2 * TD_H, v.b0, TD_H * sizeof(float));
    memcpy(pk.data.data() + o.projBias + 4 * TD_H, q.b0, TD_H * sizeof(float));
    o.qGamma = pack_vec(pk, q.g, TD_H);
    o.qBeta = pack_vec(pk, q.b, TD_H);
    o.q3B = pack_B128(pk, q.w3, TD_H, 0);
    o.q3Bias = pack_vec(pk, q.b3, TD_H);
    o.projB3 = pack_B128_split(pk, k.w0, in_dim, hi_col);      // consecutive blocks, same matrix order as projB
    pack_B128_split(pk, k.w0, in_dim, hj_col);
    pack_B128_split(pk, v.w0, in_dim, hi_col);
    pack_B128_split(pk, v.w0, in_dim, hj_col);
    pack_B128_split(pk, q.w0, TD_H, 0);
    o.q3B3 = pack_B128_split(pk, q.w3, TD_H, 0);
    return o;
}

float gaussian_coeff(const float *off) {                // models/common.py:18
    const float d = off[1] - off[0];
    return -0.5f / (d * d);
}


}  // namespace tdapi

extern "C" size_t td_model_num_weights(const td_config *cfg) {
    if (!cfg) return 0;
    const td_config &c = *cfg;
    const int H = c.hidden_dim, E = H - 1, KV = kv_in(c);
    size_t n = (size_t)E * c.protein_feat_dim + E + (size_t)E * c.ligand_num_classes + E;
    n += c.num_r_gaussian + (c.ew_net_type == 0 ? mlp_floats(c.num_r_gaussian, H, 1) : 0);
    const size_t nx = num_x2h(c), nh = num_h2x(c);
    size_t xs = 2 * mlp_floats(KV, H, H) + mlp_floats(H, H, H), hs = mlp_floats(KV, H, H) + mlp_floats(KV, H, c.n_heads) + mlp_floats(H, H, H);
    if (c.ew_net_type == 1) { xs += 4 * c.num_r_gaussian + 1; hs += 4 * c.num_r_gaussian + 1; }   // the stages' ew_net ('r')
    if (c.ew_net_type == 3) xs += H + 1;                                                            // 'm': the x2h stages' ew_net
    if (c.x2h_out_fc) xs += mlp_floats(2 * H, H, H);                                               // node_output
    n += (size_t)c.num_layers * (c.num_r_gaussian + nx * xs + nh * hs);
    n += (size_t)H * H + H + (size_t)c.ligand_num_classes * H + c.ligand_num_classes;
    return n;
}

extern "C" int td_model_create(const td_config *cfg, const float *host_weights, size_t num_weights,
                               const float *host_schedules, size_t num_schedule_floats, td_model **out) {
    if (!cfg || !host_weights || !out) { td_set_error("td_model_create: null argument"); return TD_EINVAL; }
    const td_config &c = *cfg;
    if (!config_supported(c)) {
        td_set_error("td_model_create: unsupported configuration (need hidden 128, 16 heads, 20 gaussians, edge_feat_dim 4, "
                     "knn / max_num_neighbors in 1..%d, cutoff_mode knn|hybrid|radius; got %d/%d/%d/%d, knn %d, cutoff_mode %d, "
                     "radius %g, max_num_neighbors %d)", TD_MAX_FANIN, c.hidden_dim, c.n_heads, c.num_r_gaussian,
                     c.edge_feat_dim, c.knn, c.cutoff_mode, (double)c.radius, c.max_num_neighbors);
        return TD_EINVAL;
    }
    if (num_weights != td_model_num_weights(cfg)) {
        td_set_error("td_model_create: weight blob has %zu floats, expected %zu", num_weights, td_model_num_weights(cfg));
        return TD_EINVAL;
    }
    if (host_schedules && num_schedule_floats != (size_t)7 * c.num_timesteps &&
        num_schedule_floats != (size_t)8 * c.num_timesteps && num_schedule_floats != (size_t)10 * c.num_timesteps) {
        td_set_error("td_model_create: schedule blob has %zu floats, expected %zu (sampling) or %zu (+ alphas_cumprod)",
                     num_schedule_floats, (size_t)7 * c.num_timesteps, (size_t)8 * c.num_timesteps);
        return TD_EINVAL;
    }
    const bool has_rc = host_schedules && num_schedule_floats == (size_t)10 * c.num_timesteps;
    const bool has_abar = has_rc || (host_schedules && num_schedule_floats == (size_t)8 * c.num_timesteps);
    if (c.num_blocks < 0 || c.num_blocks > 8) {
        td_set_error("td_model_create: num_blocks must be 1 .. 8 (0 = 1), got %d", c.num_blocks);
        return TD_EINVAL;
    }
    if (c.ew_net_type < 0 || c.ew_net_type > 3 || (c.x2h_out_fc != 0 && c.x2h_out_fc != 1)) {
        td_set_error("td_model_create: ew_net_type must be 0 ('global'), 1 ('r'), 2 (none) or 3 ('m'), x2h_out_fc 0 or 1; got %d / %d",
                     c.ew_net_type, c.x2h_out_fc);
        return TD_EINVAL;
    }
    if (c.num_x2h < 0 || c.num_x2h > 4 || c.num_h2x < 0 || c.num_h2x > 4) {
        td_set_error("td_model_create: num_x2h / num_h2x must be 1 .. 4 (0 = 1), got %d / %d", c.num_x2h, c.num_h2x);
        return TD_EINVAL;
    }
    if (stage_rows(c) > 1 && c.sync_twoup) {
        td_set_error("td_model_create: sync_twoup with several stages per layer is not built");
        return TD_EINVAL;
    }
    if (c.sync_twoup != 0 && c.sync_twoup != 1) { td_set_error("td_model_create: sync_twoup must be 0 or 1, got %d", c.sync_twoup); return TD_EINVAL; }
    if (c.ew_net_type != 0 && !default_graph(c)) {
        td_set_error("td_model_create: ew_net_type 'r' / 'm' / none runs on the 32-slot graphs only (knn <= 32, or radius with a cap <= 32)");
        return TD_EINVAL;
    }
    if (c.model_mean_type != 0 && c.model_mean_type != 1) {
        td_set_error("td_model_create: model_mean_type must be 0 ('C0') or 1 ('noise'), got %d", c.model_mean_type);
        return TD_EINVAL;
    }
    if (c.model_mean_type == 1 && !has_rc) {
        td_set_error("td_model_create: model_mean_type 'noise' needs the 10 schedule arrays (sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod)");
        return TD_EINVAL;
    }
    const int H = TD_H, E = H - 1, F = c.protein_feat_dim, C = c.ligand_num_classes, KV = kv_in(c), L = c.num_layers;
    Cursor cur{host_weights, num_weights};
    const float *Wp = cur.take((size_t)E * F), *bp = cur.take(E);
    const float *Wl = cur.take((size_t)E * C), *bl = cur.take(E);
    const float *goff = cur.take(TD_NG);
    // (without the global gate -- ew_net_type 'r' / none -- the blob has no edge_pred_layer; a zero MLP stands in for the packer)
    static const std::vector<float> zero_mlp(mlp_floats(TD_NG, TD_H, 1), 0.f);
    Cursor zc{zero_mlp.data(), zero_mlp.size()};
    MlpSrc gate = c.ew_net_type == 0 ? cur.mlp(TD_NG, H, 1) : zc.mlp(TD_NG, H, 1);
    const FoldedMlp fgate(gate, TD_NG, H, 1);          // LayerNorm folded into the two Linears, like the edge MLPs'
    gate = fgate.src();
    bool fold_overflow = fgate.overflow_risk;

    Packer pk;
    // ---- embeddings (+ node indicator column, models/molopt_score_model.py:336-338)
    size_t oWpT = pk.alloc((size_t)F * H), obp = pk.alloc(H), oWlT = pk.alloc((size_t)C * H), obl = pk.alloc(H);
    for (int cc = 0; cc < F; ++cc)
        for (int n = 0; n < E; ++n) pk.data[oWpT + (size_t)cc * H + n] = Wp[(size_t)n * F + cc];
    for (int cc = 0; cc < C; ++cc)
        for (int n = 0; n < E; ++n) pk.data[oWlT + (size_t)cc * H + n] = Wl[(size_t)n * C + cc];
    for (int n = 0; n < E; ++n) { pk.data[obp + n] = bp[n]; pk.data[obl + n] = bl[n]; }
    pk.data[obp + E] = 0.f;
    pk.data[obl + E] = 1.f;
    // ---- gate
    size_t oGR = pk.alloc((size_t)TD_SLOT_STEPS * 64 * 4);
    for (int s = 0; s < TD_SLOT_STEPS; ++s)
        for (int lane = 0; lane < 64; ++lane)
            for (int t = 0; t < 4; ++t) {
                const int kk = td_kmap(s, lane >> 5), n = 32 * t + (lane & 31);
                pk.data[oGR + ((size_t)s * 64 + lane) * 4 + t] = kk < TD_NG ? gate.w0[(size_t)n * TD_NG + kk] : 0.f;
            }
    size_t oGb0 = pack_vec(pk, gate.b0, H), oGg = pack_vec(pk, gate.g, H), oGb = pack_vec(pk, gate.b, H),
           oGw3 = pack_vec(pk, gate.w3, H), oGoff = pack_vec(pk, goff, TD_NG);
    const float gate_b3 = gate.b3[0], gate_coeff = gaussian_coeff(goff);
    // the same first layer as exact bf16 piece triples, K-packed like the edge MLPs' (pack_pk4_table; no type column)
    const size_t oGRp = pk.alloc(PK4_WORDS);
    pack_pk4_table(reinterpret_cast<uint32_t *>(pk.data.data() + oGRp), [&](int n, int k) { return k < TD_NG ? gate.w0[(size_t)n * TD_NG + k] : 0.f; });
    // ---- layers.  A reference layer is num_x2h x2h stages followed by num_h2x h2x stages (models/uni_transformer.py:190-206; both 1 in
    // configs/training.yml); row l * M + i of m->layers holds x2h stage i and h2x stage i of layer l (M = max of the two counts), so with
    // one stage of each kind the array is the layer list.  Blob order per layer: offsets; per x2h stage hk, hv, hq, [node_output],
    // [ew_net]; per h2x stage xk, xv, xq, [ew_net].
    struct LayerOff { NodeOff nx, nh; EdgeOff hk, hv, xk, xv; size_t off; float coeff; size_t ew_x2h = 0, ew_h2x = 0, gate_m = 0, noB = 0, nob1 = 0, nog = 0, nobeta = 0, nob2 = 0;
                      bool has_x = false, has_h = false; };
    const int NX = num_x2h(c), NH = num_h2x(c), M = stage_rows(c);
    std::vector<LayerOff> lo((size_t)L * M);
    auto gate_rows = [&](size_t &dst, const float *src) {          // [4 types][20] + bias; 'm' and none: sigmoid(40) = 1.0f
        dst = pk.alloc(4 * TD_NG + 1);
        if (src) memcpy(pk.data.data() + dst, src, (4 * TD_NG + 1) * sizeof(float));
        else pk.data[dst + 4 * TD_NG] = 40.f;
    };
    for (int l = 0; l < L && cur.ok; ++l) {
        const float *off = cur.take(TD_NG);
        if (!cur.ok) break;
        const size_t ooff = pack_vec(pk, off, TD_NG);
        for (int i = 0; i < M; ++i) { lo[(size_t)l * M + i].off = ooff; lo[(size_t)l * M + i].coeff = gaussian_coeff(off); }
        for (int i = 0; i < NX; ++i) {
            LayerOff &o = lo[(size_t)l * M + i];
            MlpSrc hk = cur.mlp(KV, H, H), hv = cur.mlp(KV, H, H), hq = cur.mlp(H, H, H);
            MlpSrc nout{};
            if (c.x2h_out_fc) nout = cur.mlp(2 * H, H, H);
            const float *ewx = c.ew_net_type == 1 ? cur.take(4 * TD_NG + 1) : nullptr;
            const float *ewm = c.ew_net_type == 3 ? cur.take(H + 1) : nullptr;
            if (!cur.ok) break;
            // the edge MLPs with their LayerNorm folded in (the query MLPs run node-side and keep theirs)
            const FoldedMlp fhk(hk, KV, H, H, true), fhv(hv, KV, H, H, true);
            hk = fhk.src(); hv = fhv.src();
            fold_overflow = fold_overflow || fhk.overflow_risk || fhv.overflow_risk;
            if (c.ew_net_type != 0) gate_rows(o.ew_x2h, ewx);
            if (ewm) {          // u'_n = sum_o w_m[o] W2v'[o][n] on the FOLDED second Linear (its columns carry |gamma_n|), c = w_m . b2v + b_m
                o.gate_m = pk.alloc(TD_H + 1);
                double cc = ewm[H];
                for (int oo = 0; oo < H; ++oo) cc += (double)ewm[oo] * (double)fhv.b3[oo];
                for (int n = 0; n < H; ++n) {
                    double u = 0.0;
                    for (int oo = 0; oo < H; ++oo) u += (double)ewm[oo] * (double)fhv.w3[(size_t)oo * H + n];
                    pk.data[o.gate_m + n] = (float)u;
                }
                pk.data[o.gate_m + H] = (float)cc;
            }
            if (c.x2h_out_fc) {
                o.noB = pack_B128(pk, nout.w0, 2 * TD_H, 0);           // the attention-output half of net.0 (cat([output, h]), :83)
                pack_B128(pk, nout.w0, 2 * TD_H, TD_H);                // the h half (consecutive blocks)
                pack_B128(pk, nout.w3, TD_H, 0);
                o.nob1 = pack_vec(pk, nout.b0, TD_H);
                o.nog = pack_vec(pk, nout.g, TD_H);
                o.nobeta = pack_vec(pk, nout.b, TD_H);
                o.nob2 = pack_vec(pk, nout.b3, TD_H);
            }
            o.nx = pack_node_stage(pk, hk, hv, hq, KV);
            o.hk = pack_edge_mlp(pk, fhk, KV, H, 1);
            o.hv = pack_edge_mlp(pk, fhv, KV, H, 2);
            o.has_x = true;
        }
        for (int j = 0; j < NH && cur.ok; ++j) {
            LayerOff &o = lo[(size_t)l * M + j];
            MlpSrc xk = cur.mlp(KV, H, H), xv = cur.mlp(KV, H, c.n_heads), xq = cur.mlp(H, H, H);
            const float *ewh = c.ew_net_type == 1 ? cur.take(4 * TD_NG + 1) : nullptr;
            if (!cur.ok) break;
            const FoldedMlp fxk(xk, KV, H, H, true), fxv(xv, KV, H, c.n_heads, true);
            xk = fxk.src(); xv = fxv.src();
            fold_overflow = fold_overflow || fxk.overflow_risk || fxv.overflow_risk;
            if (c.ew_net_type != 0) gate_rows(o.ew_h2x, ewh);
            o.nh = pack_node_stage(pk, xk, xv, xq, KV);
            o.xk = pack_edge_mlp(pk, fxk, KV, H, 1);
            o.xv = pack_edge_mlp(pk, fxv, KV, c.n_heads, 0);
            o.has_h = true;
        }
    }
    // ---- head
    const float *V0 = cur.take((size_t)H * H), *vb0 = cur.take(H), *V2 = cur.take((size_t)C * H), *vb2 = cur.take(C);
    if (!cur.ok || cur.left != 0) { td_set_error("td_model_create: weight blob layout mismatch"); return TD_EINVAL; }
    if (fold_overflow) {
        td_set_error("td_model_create: an edge MLP's LayerNorm has bias / |weight| beyond 1e15 on a unit that is not negligible "
                     "(|weight| > 2^-30 of the MLP's largest): the folded form would overflow fp32");
        return TD_EINVAL;
    }
    size_t oW0T = pk.alloc((size_t)H * H), ohb0 = pack_vec(pk, vb0, H), oW2T = pk.alloc((size_t)H * TD_MAXC),
           ohb2 = pack_vec(pk, vb2, C, TD_MAXC);
    for (int k = 0; k < H; ++k) {
        for (int n = 0; n < H; ++n) pk.data[oW0T + (size_t)k * H + n] = V0[(size_t)n * H + k];
        for (int cc = 0; cc < C; ++cc) pk.data[oW2T + (size_t)k * TD_MAXC + cc] = V2[(size_t)cc * H + k];
    }
    // ---- schedules
    const int T = c.num_timesteps;
    size_t oS = pk.alloc((size_t)10 * T);
    if (host_schedules) memcpy(pk.data.data() + oS, host_schedules, (size_t)(has_rc ? 10 : (has_abar ? 8 : 7)) * T * sizeof(float));

    td_model *m = new (std::nothrow) td_model();
    if (!m) { td_set_error("td_model_create: out of host memory"); return TD_ENOMEM; }
    m->cfg = c;
    m->blob_floats = pk.data.size();
    m->layers = new (std::nothrow) TdLayer[(size_t)L * stage_rows(c)]();
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&m->blob), m->blob_floats * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(m->blob, pk.data.data(), m->blob_floats * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess || !m->layers) {
        td_set_error("td_model_create: device upload failed: %s", hipGetErrorString(e));
        if (m->blob) (void)hipFree(m->blob);
        delete[] m->layers;
        delete m;
        return e != hipSuccess ? TD_EHIP : TD_ENOMEM;
    }
    const float *D = m->blob;
    m->emb = TdEmbed{D + oWpT, D + obp, D + oWlT, D + obl};
    m->gate = TdGate{D + oGR, D + oGb0, D + oGg, D + oGb, D + oGw3, gate_b3, D + oGoff, gate_coeff, D + oGRp, fgate.ln_c1, fgate.ln_c2, m->opt.edge_key_split != 0};
    auto edge = [&](const EdgeOff &o, bool split = false) {
        return TdEdgeMlp{D + o.R, D + o.gamma, D + o.beta, D + o.W2, D + o.b2, D + o.R16, D + o.Walt16, D + o.Walt,
                         D + o.R16q, o.R16h ? D + o.R16h : nullptr, o.ln_c1, o.ln_c2, o.w2_bound, m->opt.edge_second_layer_f16 != 0, m->opt.edge_first_layer_f16 != 0 && o.R16h != 0, o.z_plain, split && m->opt.edge_key_split != 0, m->opt.edge_row_dealing};
    };
    auto node = [&](const NodeOff &o) {
        return TdNodeStage{D + o.projB, D + o.projBias, D + o.qGamma, D + o.qBeta, D + o.q3B, D + o.q3Bias, D + o.projB3, D + o.q3B3,
                           m->opt.node_proj_split != 0, m->opt.node_proj_bpipe != 0, m->opt.node_proj_async != 0};
    };
    for (size_t l = 0; l < lo.size(); ++l) {
        TdLayer &Ly = m->layers[l];
        Ly = TdLayer{};
        Ly.offsets = D + lo[l].off; Ly.coeff = lo[l].coeff;
        if (lo[l].has_x) {
            Ly.nodeX2h = node(lo[l].nx);
            Ly.hk = edge(lo[l].hk, true); Ly.hv = edge(lo[l].hv, true);
            Ly.ew_x2h = c.ew_net_type != 0 ? D + lo[l].ew_x2h : nullptr;
            Ly.gate_m = c.ew_net_type == 3 ? D + lo[l].gate_m : nullptr;
            if (c.x2h_out_fc) Ly.nodeOut = TdNodeOut{D + lo[l].noB, D + lo[l].nob1, D + lo[l].nog, D + lo[l].nobeta, D + lo[l].nob2};
        }
        if (lo[l].has_h) {
            Ly.nodeH2x = node(lo[l].nh);
            Ly.xk = edge(lo[l].xk, true); Ly.xv = edge(lo[l].xv, true);
            Ly.ew_h2x = c.ew_net_type != 0 ? D + lo[l].ew_h2x : nullptr;
        }
    }
    m->head = TdHead{D + oW0T, D + ohb0, D + oW2T, D + ohb2};
    const float *S = D + oS;
    m->sched = TdSchedules{S, S + T, S + 2 * T, S + 3 * T, S + 4 * T, S + 5 * T, S + 6 * T, has_abar ? S + 7 * T : nullptr,
                           has_rc ? S + 8 * T : nullptr, has_rc ? S + 9 * T : nullptr};
    *out = m;
    return TD_OK;
}

extern "C" void td_model_destroy(td_model *m) {
    if (!m) return;
    if (m->blob) (void)hipFree(m->blob);
    delete[] m->layers;
    delete m;
}

extern "C" int td_model_set_option(td_model *m, const char *name, int32_t value) {
    if (!m || !name) { td_set_error("td_model_set_option: null argument"); return TD_EINVAL; }
    if (strcmp(name, "h2x_fused") == 0) m->opt.h2x_fused = value != 0;
    else if (strcmp(name, "node_proj_split") == 0) {
        m->opt.node_proj_split = value != 0;
        for (int l = 0; l < m->cfg.num_layers * stage_rows(m->cfg); ++l) m->layers[l].nodeX2h.use_split = m->layers[l].nodeH2x.use_split = value != 0;
    } else if (strcmp(name, "edge_row_dealing") == 0) {
        m->opt.edge_row_dealing = value < 0 ? 0 : (value > 2 ? 2 : value);
        for (int l = 0; l < m->cfg.num_layers * stage_rows(m->cfg); ++l)
            m->layers[l].hk.deal_rows = m->layers[l].hv.deal_rows = m->layers[l].xk.deal_rows = m->layers[l].xv.deal_rows = m->opt.edge_row_dealing;
    } else if (strcmp(name, "node_proj_bpipe") == 0) {
        m->opt.node_proj_bpipe = value != 0;
        for (int l = 0; l < m->cfg.num_layers * stage_rows(m->cfg); ++l) m->layers[l].nodeX2h.bpipe = m->layers[l].nodeH2x.bpipe = value != 0;
    } else if (strcmp(name, "node_proj_async") == 0) {
        m->opt.node_proj_async = value != 0;
        for (int l = 0; l < m->cfg.num_layers * stage_rows(m->cfg); ++l) m->layers[l].nodeX2h.async_copy = m->layers[l].nodeH2x.async_copy = value != 0;
    } else if (strcmp(name, "edge_key_split") == 0) {
        m->opt.edge_key_split = value != 0;
        m->gate.use_split = value != 0;
        for (int l = 0; l < m->cfg.num_layers * stage_rows(m->cfg); ++l) {
            m->layers[l].hk.use_split = m->layers[l].hk.R16q && value != 0;
            m->layers[l].hv.use_split = m->layers[l].hv.R16q && value != 0;
            m->layers[l].xk.use_split = m->layers[l].xk.R16q && value != 0;
            m->layers[l].xv.use_split = m->layers[l].xv.R16q && value != 0;
        }
    } else if (strcmp(name, "edge_first_layer_f16") == 0) {
        m->opt.edge_first_layer_f16 = value != 0;
        for (int l = 0; l < m->cfg.num_layers * stage_rows(m->cfg); ++l) {
            m->layers[l].hk.l1_f16 = m->layers[l].hk.R16h && value != 0;
            m->layers[l].hv.l1_f16 = m->layers[l].hv.R16h && value != 0;
            m->layers[l].xk.l1_f16 = m->layers[l].xk.R16h && value != 0;
            m->layers[l].xv.l1_f16 = m->layers[l].xv.R16h && value != 0;
        }
    } else if (strcmp(name, "edge_second_layer_f16") == 0) {
        m->opt.edge_second_layer_f16 = value != 0;
        for (int l = 0; l < m->cfg.num_layers * stage_rows(m->cfg); ++l)
            m->layers[l].hk.l2_f16 = m->layers[l].hv.l2_f16 = m->layers[l].xk.l2_f16 = m->layers[l].xv.l2_f16 = value != 0;
    } else if (strcmp(name, "session_hop_levels") == 0) {
        if (value < 1 || value > TD_HOP_LEVELS) { td_set_error("td_model_set_option: session_hop_levels must be 1..%d", TD_HOP_LEVELS); return TD_EINVAL; }
        m->opt.session_hop_levels = value;
    } else if (strcmp(name, "session_forward_reach") == 0) m->opt.session_forward_reach = value != 0;
    else if (strcmp(name, "session_step_lists") == 0) m->opt.session_step_lists = value != 0;
    else if (strcmp(name, "session_share_pockets") == 0) m->opt.session_share_pockets = value != 0;
    else { td_set_error("td_model_set_option: unknown option '%s'", name); return TD_EINVAL; }
    ++m->option_epoch;          // sessions re-capture their step graph (the captured nodes copied the old variants' arguments by value)
    return TD_OK;
}

extern "C" int td_model_get_option(const td_model *m, const char *name, int32_t *value) {
    if (!m || !name || !value) { td_set_error("td_model_get_option: null argument"); return TD_EINVAL; }
    if (strcmp(name, "h2x_fused") == 0) *value = m->opt.h2x_fused;
    else if (strcmp(name, "node_proj_split") == 0) *value = m->opt.node_proj_split;
    else if (strcmp(name, "node_proj_async") == 0) *value = m->opt.node_proj_async;
    else if (strcmp(name, "node_proj_bpipe") == 0) *value = m->opt.node_proj_bpipe;
    else if (strcmp(name, "edge_row_dealing") == 0) *value = m->opt.edge_row_dealing;
    else if (strcmp(name, "edge_key_split") == 0) *value = m->opt.edge_key_split;
    else if (strcmp(name, "edge_second_layer_f16") == 0) *value = m->opt.edge_second_layer_f16;
    else if (strcmp(name, "edge_first_layer_f16") == 0) *value = m->opt.edge_first_layer_f16;
    else if (strcmp(name, "session_hop_levels") == 0) *value = m->opt.session_hop_levels;
    else if (strcmp(name, "session_forward_reach") == 0) *value = m->opt.session_forward_reach;
    else if (strcmp(name, "session_step_lists") == 0) *value = m->opt.session_step_lists;
    else if (strcmp(name, "session_share_pockets") == 0) *value = m->opt.session_share_pockets;
    else { td_set_error("td_model_get_option: unknown option '%s'", name); return TD_EINVAL; }
    return TD_OK;
}
