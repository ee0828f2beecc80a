// C ABI of libtargetdiff_hip.so: weight re-packing, workspace carving and the per-step launch sequence.
// See include/targetdiff_hip.h for the contract of every entry point and the reference seam it replaces.
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

// ------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

void td_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *td_last_error(void) { return g_err; }
extern "C" int td_abi_version(void) { return TD_ABI_VERSION; }
#ifndef TD_BUILD_TAG
#define TD_BUILD_TAG "untagged"
#endif
extern "C" const char *td_build_tag(void) { return TD_BUILD_TAG; }

// ------------------------------------------------------------------------------------------ kernel timers
// Optional per-kernel-class HIP-event timers (bench.py's roofline leg): events are recorded on the launch
// stream around the selected classes; td_profile_end synchronises and sums hipEventElapsedTime.
namespace {
enum { PC_KNN = 0, PC_GATE, PC_NODE, PC_X2H_K, PC_X2H_V, PC_H2X_K, PC_H2X_V, PC_COMPOSE, PC_HEAD, PC_POST, PC_COUNT };
struct Profiler {
    std::atomic<unsigned> mask{0};
    std::mutex mu;                            // the event lists: launches may come from several host threads (one stream each)
    std::vector<hipEvent_t> ev[PC_COUNT];     // start/stop pairs
    std::vector<hipEvent_t> pool;
    hipEvent_t get() {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
};
Profiler g_prof;
struct ProfScope {
    int cls; hipStream_t s; bool on; hipEvent_t e0 = nullptr;
    ProfScope(int c, hipStream_t st) : cls(c), s(st), on((g_prof.mask.load(std::memory_order_relaxed) >> c) & 1u) {
        if (on) { e0 = g_prof.get(); (void)hipEventRecord(e0, s); }
    }
    ~ProfScope() {
        if (on) {
            hipEvent_t e1 = g_prof.get();
            (void)hipEventRecord(e1, s);
            std::lock_guard<std::mutex> lk(g_prof.mu);       // the pair enters the list together, whatever other threads record meanwhile
            g_prof.ev[cls].push_back(e0);
            g_prof.ev[cls].push_back(e1);
        }
    }
};
}  // namespace

extern "C" int td_profile_begin(uint32_t class_mask) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    for (int c = 0; c < PC_COUNT; ++c) {
        for (hipEvent_t e : g_prof.ev[c]) g_prof.pool.push_back(e);
        g_prof.ev[c].clear();
    }
    g_prof.mask = class_mask;
    return TD_OK;
}

extern "C" int td_profile_end(float *ms_out, int32_t *count_out, int32_t num_classes) {
    g_prof.mask = 0;
    TD_CHECK_HIP(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(g_prof.mu);
    for (int c = 0; c < PC_COUNT; ++c) {
        float total = 0.f;
        int n = 0;
        for (size_t i = 0; i + 1 < g_prof.ev[c].size(); i += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, g_prof.ev[c][i], g_prof.ev[c][i + 1]) == hipSuccess) { total += ms; ++n; }
        }
        if (c < num_classes) {
            if (ms_out) ms_out[c] = total;
            if (count_out) count_out[c] = n;
        }
        for (hipEvent_t e : g_prof.ev[c]) g_prof.pool.push_back(e);
        g_prof.ev[c].clear();
    }
    return TD_OK;
}

// ------------------------------------------------------------------------------------------ blob layout
namespace {

struct MlpSrc {          // one reference MLP (models/common.py:60-80) inside the flat blob
    const float *w0, *b0, *g, *b, *w3, *b3;
};

struct Cursor {
    const float *p;
    size_t left;
    bool ok = true;
    const float *take(size_t n) {
        if (n > left) { ok = false; return p; }
        const float *r = p;
        p += n; left -= n;
        return r;
    }
    MlpSrc mlp(int in, int hid, int out) {
        MlpSrc m;
        m.w0 = take((size_t)hid * in); m.b0 = take(hid); m.g = take(hid); m.b = take(hid);
        m.w3 = take((size_t)out * hid); m.b3 = take(out);
        return m;
    }
};

// The LayerNorm of an edge MLP (Linear -> LayerNorm -> ReLU -> Linear, models/common.py:60-80) folded into its two Linears at pack
// time.  With c = pre-activation minus its mean over the hidden units, sigma = sqrt(var + eps), s_n = sign(gamma_n), a_n = |gamma_n|:
//     relu(gamma_n c_n / sigma + beta_n) = (a_n / sigma) relu(s_n c_n + (beta_n / a_n) sigma)
//  * every column of the first Linear (and its bias) is centred over the hidden units and row n multiplied by s_n: the matrix product
//    IS s_n c_n -- the radial/type table, the node-side projections P_i / P_j and the bias all come from these rows, so the kernels
//    compute neither a mean nor a subtraction;
//  * beta_n / a_n replaces beta, a_n goes into column n of the second Linear, and 1 / sigma (one number per edge) multiplies the
//    second layer's per-edge result in the consumer (logit, xv, or the attention weight of the aggregation).
// Exact algebra.  (gamma_n = 0 is carried as a_n = 1e-20: the unit's constant relu(beta_n) survives.)
// Round 5 -- the ReLU as the FMA's own output clamp.  Dividing by sigma M instead of carrying sigma along,
//     relu(s_n c_n / sigma + beta_n / a_n) / M = clamp_[0,1](s_n c_n (1 / (sigma M)) + beta_n / (a_n M)),
// holds whenever the left side never exceeds 1: |c_n| <= sqrt(hid) sigma (the c_n are centred and sigma^2 >= their mean square), so
// M = (sqrt(hid) + max_n beta_n / a_n) (1 + 2^-10) does.  v_fma_f32 has a free clamp-to-[0, 1] output modifier: ONE instruction per hidden
// value where the round-4 form took an FMA and a max, no per-edge 1 / sigma for the consumers to apply (M a_n goes into column n of the
// second Linear instead of a_n), and the kernels keep s_e = 1 / (sigma_e M) per edge only as the FMA's multiplier.
struct FoldedMlp {
    std::vector<float> w0, b0, g, b, w3;
    const float *b3;
    float ln_c1 = 1.f / 128.f, ln_c2 = 1e-5f;        // the kernels' variance constants (below)
    FoldedMlp(const MlpSrc &m, int in, int hid, int out) : w0((size_t)hid * in), b0(hid), g(hid), b(hid), w3((size_t)out * hid), b3(m.b3) {
        std::vector<float> sg(hid);
        double bmax = 0.0;
        for (int n = 0; n < hid; ++n) {
            const float a = fabsf(m.g[n]) > 1e-20f ? fabsf(m.g[n]) : 1e-20f;
            sg[n] = m.g[n] < 0.f ? -1.f : 1.f;
            g[n] = a;
            b[n] = m.b[n] / a;
            if ((double)b[n] > bmax) bmax = (double)b[n];
        }
        const double M = (sqrt((double)hid) + bmax) * (1.0 + 1.0 / 1024.0);
        for (int n = 0; n < hid; ++n) {
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
    }
    MlpSrc src() const { return MlpSrc{w0.data(), b0.data(), g.data(), b.data(), w3.data(), b3}; }
};

size_t mlp_floats(int in, int hid, int out) { return (size_t)hid * in + 3 * (size_t)hid + (size_t)out * hid + out; }

// stages per layer (0 in the config = 1): m->layers has stage_rows(c) rows per reference layer
inline int num_x2h(const td_config &c) { return c.num_x2h > 0 ? c.num_x2h : 1; }
inline int num_h2x(const td_config &c) { return c.num_h2x > 0 ? c.num_h2x : 1; }
inline int stage_rows(const td_config &c) { return num_x2h(c) > num_h2x(c) ? num_x2h(c) : num_h2x(c); }

int kv_in(const td_config &c) { return 2 * c.hidden_dim + c.edge_feat_dim + 4 * c.num_r_gaussian; }

bool config_supported(const td_config &c) {
    const bool graph_ok = (c.cutoff_mode == TD_CUTOFF_KNN && c.knn >= 1 && c.knn <= TD_MAX_FANIN) ||
                          (c.cutoff_mode == TD_CUTOFF_HYBRID && c.knn >= 1 && c.knn <= TD_MAX_FANIN) ||
                          (c.cutoff_mode == TD_CUTOFF_RADIUS && c.radius > 0.f && c.max_num_neighbors >= 1 &&
                           c.max_num_neighbors <= TD_MAX_FANIN);
    return c.hidden_dim == TD_H && c.n_heads == TD_HEADS && graph_ok && c.num_r_gaussian == TD_NG &&
           c.edge_feat_dim == 4 && c.num_layers >= 1 && c.protein_feat_dim >= 1 && c.protein_feat_dim <= 32 &&
           c.ligand_num_classes >= 1 && c.ligand_num_classes <= TD_MAXC && c.num_timesteps >= 1;
}

// the graph every kernel's fast path is specialised for: exactly 32 in-edges per node, one MFMA tile per dst row
// (a k-NN graph with k < 32 is the 32-NN graph with the slots >= k masked, so it shares that path; so does a radius graph
// whose fan-out cap is <= 32).  The caching session additionally needs the k-NN structure (sorted lists to merge into).
bool caching_graph(const td_config &c) { return c.cutoff_mode == TD_CUTOFF_KNN && c.knn <= TD_K; }
bool default_graph(const td_config &c) {
    return caching_graph(c) || (c.cutoff_mode == TD_CUTOFF_RADIUS && c.max_num_neighbors <= TD_K);
}

// Packed-buffer builder: collects tensors into one host vector; pointers are fixed up after the upload.
struct Packer {
    std::vector<float> data;
    size_t alloc(size_t n) {
        size_t off = (data.size() + 63) & ~size_t(63);          // 256-byte alignment
        data.resize(off + n, 0.f);
        return off;
    }
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

size_t pack_vec(Packer &pk, const float *v, size_t n, size_t padded = 0) {
    size_t off = pk.alloc(padded ? padded : n);
    if (v) memcpy(pk.data.data() + off, v, n * sizeof(float));
    return off;
}

struct EdgeOff { size_t R, gamma, beta, W2, b2, Walt, R16, Walt16, R16q; float ln_c1, ln_c2; };

EdgeOff pack_edge_mlp(Packer &pk, const FoldedMlp &fm, int in_dim, int out_dim, int alt) {
    const MlpSrc m = fm.src();
    EdgeOff o;
    o.ln_c1 = fm.ln_c1; o.ln_c2 = fm.ln_c2;
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
    } else if (alt == 2) {   // value MLP of x2h: W2vK[k/4][n][k%4]
        o.Walt = pk.alloc((size_t)32 * TD_H * 4);
        float *q = pk.data.data() + o.Walt;
        for (int kq = 0; kq < 32; ++kq)
            for (int n = 0; n < TD_H; ++n)
                for (int kk = 0; kk < 4; ++kk) q[((size_t)kq * TD_H + n) * 4 + kk] = m.w3[(size_t)n * TD_H + 4 * kq + kk];
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

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

}  // namespace

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
            const FoldedMlp fhk(hk, KV, H, H), fhv(hv, KV, H, H);
            hk = fhk.src(); hv = fhv.src();
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
            const FoldedMlp fxk(xk, KV, H, H), fxv(xv, KV, H, c.n_heads);
            xk = fxk.src(); xv = fxv.src();
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
                         D + o.R16q, o.ln_c1, o.ln_c2, split && m->opt.edge_key_split != 0, m->opt.edge_row_dealing};
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
    } else if (strcmp(name, "session_hop_levels") == 0) {
        if (value < 1 || value > TD_HOP_LEVELS) { td_set_error("td_model_set_option: session_hop_levels must be 1..%d", TD_HOP_LEVELS); return TD_EINVAL; }
        m->opt.session_hop_levels = value;
    } else if (strcmp(name, "session_forward_reach") == 0) m->opt.session_forward_reach = value != 0;
    else if (strcmp(name, "session_step_lists") == 0) m->opt.session_step_lists = value != 0;
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
    else if (strcmp(name, "session_hop_levels") == 0) *value = m->opt.session_hop_levels;
    else if (strcmp(name, "session_forward_reach") == 0) *value = m->opt.session_forward_reach;
    else if (strcmp(name, "session_step_lists") == 0) *value = m->opt.session_step_lists;
    else { td_set_error("td_model_get_option: unknown option '%s'", name); return TD_EINVAL; }
    return TD_OK;
}

// ------------------------------------------------------------------------------------------ workspace
namespace {
struct Workspace {
    float4 *x4a, *x4b;
    int32_t *gid, *nbr, *lig_node, *node_ptr;
    float *ew, *P, *q, *h, *alpha;
    float *Px, *qx;          // h2x-stage projections / queries (separate from P / q: both stages project in one launch)
    size_t bytes;
};

Workspace carve(char *base, int64_t N, int64_t B, int64_t Nl) {
    Workspace w;
    size_t off = 0;
    auto take = [&](size_t n) { char *p = base ? base + off : nullptr; off += align_up(n); return p; };
    const size_t n = (size_t)(N > 0 ? N : 1), nl = (size_t)(Nl > 0 ? Nl : n);
    w.x4a = reinterpret_cast<float4 *>(take(n * sizeof(float4)));
    w.x4b = reinterpret_cast<float4 *>(take(n * sizeof(float4)));
    w.gid = reinterpret_cast<int32_t *>(take(n * sizeof(int32_t)));
    w.nbr = reinterpret_cast<int32_t *>(take(n * TD_K * sizeof(int32_t)));
    w.lig_node = reinterpret_cast<int32_t *>(take((nl + 1) * sizeof(int32_t)));
    w.node_ptr = reinterpret_cast<int32_t *>(take((size_t)(B + 1) * sizeof(int32_t)));
    w.ew = reinterpret_cast<float *>(take(n * TD_K * sizeof(float)));
    w.P = reinterpret_cast<float *>(take(n * 4 * TD_H * sizeof(float)));
    w.q = reinterpret_cast<float *>(take(n * TD_H * sizeof(float)));
    w.h = reinterpret_cast<float *>(take(n * TD_H * sizeof(float)));
    w.alpha = reinterpret_cast<float *>(take(n * TD_HEADS * TD_K * sizeof(float)));
    w.Px = reinterpret_cast<float *>(take(n * 4 * TD_H * sizeof(float)));
    w.qx = reinterpret_cast<float *>(take(n * TD_H * sizeof(float)));
    w.bytes = off;
    return w;
}

// Neighbour table and per-slot buffers of one batch.  Default graph: cptr == nullptr, one 32-slot row per node (the
// workspace's nbr / ew / alpha).  General graphs: chunk-indexed buffers of a GraphPlan, cptr[i] .. cptr[i+1]-1 = chunks of node i.
struct GraphTab {
    const int32_t *cptr;
    int32_t *nbr;
    float *ew, *alpha;
    int cpn_p;               // chunks per protein row
    int64_t NCl;             // chunks of all ligand rows together
    const int32_t *mixed;    // device count of the rows that see both source classes (a session's dirty rows), or nullptr
};
GraphTab default_tab(Workspace &w) { return GraphTab{nullptr, w.nbr, w.ew, w.alpha, 1, 0, nullptr}; }

// lig / Nl (x2h passes): the ligand rows of the batch, all of them among `rows`
int key_pass(const TdEdgeMlp &mlp, const TdLayer &L, const float4 *x4, const GraphTab &gt, const float *ew, const int32_t *nbr,
             const float *P, const float *q, const int32_t *rows, const int32_t *count_ptr, int64_t count, float *alpha, hipStream_t s,
             const int32_t *lig = nullptr, int64_t Nl = 0, bool h2x_stage = false) {
    return td_launch_edge_key16(mlp, L, x4, nbr, ew, P, q, rows, count_ptr, count, alpha, s, h2x_stage, gt.cptr, lig, Nl, gt.cpn_p);
}
// lig / Nl: the ligand rows of the batch (all of them are among `rows`: every row list of a step contains the ligand atoms)
int value_pass(const TdEdgeMlp &mlp, const TdLayer &L, const float4 *x4, const GraphTab &gt, const int32_t *nbr, const float *P,
               const int32_t *rows, const int32_t *count_ptr, int64_t count, float *h, const float *alpha, const int32_t *lig,
               int64_t Nl, hipStream_t s, float *out = nullptr) {
    return td_launch_edge_value16(mlp, L, x4, nbr, P, rows, count_ptr, count, h, alpha, lig, Nl, s, gt.cptr, gt.cpn_p,
                                  lig ? gt.NCl : 0, gt.mixed, out, L.gate_m);
}

// h2x stage, projections in one launch: src-side (k_j, v_j) of the nodes a ligand atom can see -- `hop_rows` (the
// ligand atoms and their neighbours, fixed for the step) when the caller has the list, every node otherwise -- plus, as
// a second segment, the dst-side projections and queries of the ligand atoms
int h2x_project(const TdLayer &L, Workspace &w, float *h, int64_t N, int64_t Nl, float *P, float *q,
                const int32_t *hop_rows, const int32_t *hop_count, hipStream_t s) {
    ProfScope ps(PC_NODE, s);
    if (hop_rows) return td_launch_node_proj(L.nodeH2x, h, N, hop_rows, 0x0a, P, q, s, hop_count, w.lig_node, Nl, 0x15);
    return td_launch_node_proj(L.nodeH2x, h, N, nullptr, 0x0a, P, q, s, nullptr, w.lig_node, Nl, 0x15);
}

// attention over the ligand atoms' edges and the coordinate update xc -> xn.  Rows of several chunks (general graphs) take the
// two-launch form (keys + softmax over all chunks of a row -> alpha in memory -> xv); one 32-slot row per node: fused.
int h2x_attend(const td_model *m, const TdLayer &L, Workspace &w, const GraphTab &gt, int64_t Nl, float4 *xc, float4 *xn, float *P,
               float *q, hipStream_t s) {
    int rc;
    if (m->opt.h2x_fused && (!gt.cptr || (L.xk.use_split && L.xv.use_split))) {
        // (general graphs: the chunk-walking fused form exists for the bf16 first layer; alpha holds the logits between its two sweeps)
        ProfScope ps(PC_H2X_K, s);
        return td_launch_edge_h2x16(L.xk, L.xv, L, xc, xn, gt.nbr, gt.ew, P, q, w.lig_node, Nl, s, gt.cptr, gt.alpha);
    }
    {
        ProfScope ps(PC_H2X_K, s);
        if ((rc = key_pass(L.xk, L, xc, gt, gt.ew, gt.nbr, P, q, w.lig_node, nullptr, Nl, gt.alpha, s, nullptr, 0, true)) != TD_OK) return rc;
    }
    ProfScope ps(PC_H2X_V, s);
    return td_launch_edge_xv16(L.xv, L, xc, xn, gt.nbr, P, w.lig_node, Nl, gt.alpha, s, gt.cptr);
}

// L x (node_proj, x2h, node_proj, h2x) on a composed batch whose graph (gt.nbr) and edge gate (gt.ew) are in place.  h is
// updated in place; returns the buffer holding the final coordinates through *x_final.  init_xn: x4b is not yet a copy of x4a.
// hop_rows (optional, sampling session): the ligand atoms and their neighbours -- the only rows whose h2x-stage projections and
// last-layer features are ever read when just the ligand outputs are consumed.
// Sampling session, layer 1: rows outside the ligand's one-hop forward reach keep the protein-only graph's layer-1
// output (`hs`), so the attention passes run on `rows` only and `rest` is restored from the cache afterwards.
struct FwdReach {
    const int32_t *rows, *rest, *counts;     // counts[0] = |rows|, counts[1] = |rest|
    const float *hs;
};

int run_backbone_stages(const td_model *m, Workspace &w, const GraphTab &gt, float *h, int64_t N, int64_t Nl, int fix_x,
                        float4 **x_final, hipStream_t s, bool init_xn);

int run_backbone(const td_model *m, Workspace &w, const GraphTab &gt, float *h, int64_t N, int64_t Nl, int fix_x,
                 float4 **x_final, hipStream_t s, bool init_xn, bool layer0_x2h_done = false,
                 const int32_t *hop_rows = nullptr, const int32_t *hop_count = nullptr, int hop_levels = 0,
                 const FwdReach *fwd = nullptr) {
    int rc;
    if (stage_rows(m->cfg) > 1 || m->cfg.num_x2h > 1 || m->cfg.num_h2x > 1) return run_backbone_stages(m, w, gt, h, N, Nl, fix_x, x_final, s, init_xn);
    const int Lc = m->cfg.num_layers;
    // row list of receptive-field level k (1-based); nullptr = every row
    auto level_rows = [&](int k) -> const int32_t * { return (hop_rows && k <= hop_levels) ? hop_rows + (size_t)(k - 1) * N : nullptr; };
    auto level_count = [&](int k) -> const int32_t * { return (hop_rows && k <= hop_levels) ? hop_count + (k - 1) : nullptr; };
    float4 *xc = w.x4a, *xn = w.x4b;
    const bool do_h2x = !fix_x && Nl > 0;
    if (do_h2x && init_xn) TD_CHECK_HIP(hipMemcpyAsync(xn, xc, (size_t)N * sizeof(float4), hipMemcpyDeviceToDevice, s));
    // Sampling session: only ligand outputs are consumed, so the layer e from the end updates receptive-field level e + 1
    // only, and its projections are needed on level e + 2 (those rows and their neighbours).
    auto proj_rows = [&](int l) -> const int32_t * { return l > 0 ? level_rows(Lc - 1 - l + 2) : nullptr; };
    auto proj_count = [&](int l) -> const int32_t * { return l > 0 ? level_count(Lc - 1 - l + 2) : nullptr; };
    bool proj_done = false;        // this layer's x2h-stage projections rode in the previous layer's paired launch
    const bool sync = m->cfg.sync_twoup != 0;
    for (int l = 0; l < Lc; ++l) {
        const TdLayer &L = m->layers[l];
        if (!(l == 0 && layer0_x2h_done)) {
            const int e = Lc - 1 - l;
            const int32_t *rws = l > 0 ? level_rows(e + 1) : nullptr, *cnt = l > 0 ? level_count(e + 1) : nullptr;
            const bool use_fwd = fwd && l == 1 && !rws;
            if (use_fwd) { rws = fwd->rows; cnt = fwd->counts; }
            if (sync && do_h2x) {
                // sync_twoup: the h2x stage reads the layer's input features, i.e. the h this stage's projections are taken from: both
                // stages' projections in one launch, before the value pass overwrites h
                ProfScope ps(PC_NODE, s);
                if ((rc = td_launch_node_proj_pair(L.nodeH2x, level_rows(1), level_count(1), w.lig_node, Nl, w.Px, w.qx, L.nodeX2h,
                                                   proj_rows(l), proj_count(l), w.P, w.q, h, N, s)) != TD_OK) return rc;
            } else if (!proj_done) {
                ProfScope ps(PC_NODE, s);
                if ((rc = td_launch_node_proj(L.nodeX2h, h, N, proj_rows(l), 0x1f, w.P, w.q, s, proj_count(l))) != TD_OK) return rc;
            }
            proj_done = false;
            if (L.ew_x2h) {          // ew_net_type 'r' / none: this stage's own gate from the layer's coordinates (no caching: every row)
                ProfScope ps(PC_GATE, s);
                if ((rc = td_launch_layer_gate(L.ew_x2h, L.offsets, L.coeff, xc, gt.nbr, N, gt.ew, s)) != TD_OK) return rc;
            }
            { ProfScope ps(PC_X2H_K, s); if ((rc = key_pass(L.hk, L, xc, gt, gt.ew, gt.nbr, w.P, w.q, rws, cnt, N, gt.alpha, s, w.lig_node, Nl)) != TD_OK) return rc; }
            // x2h_out_fc: the attention output goes to w.q (the queries are spent once the key pass is done) without the residual, and
            // node_output([output | h]) + h follows on every row
            float *att_out = L.nodeOut.B ? w.q : nullptr;
            { ProfScope ps(PC_X2H_V, s); if ((rc = value_pass(L.hv, L, xc, gt, gt.nbr, w.P, rws, cnt, N, h, gt.alpha, w.lig_node, Nl, s, att_out)) != TD_OK) return rc; }
            if (att_out) {
                ProfScope ps(PC_NODE, s);
                if ((rc = td_launch_node_output(L.nodeOut, att_out, h, N, s)) != TD_OK) return rc;
            }
            if (use_fwd && (rc = td_launch_restore_rows(fwd->rest, fwd->counts + 1, N, fwd->hs, h, s)) != TD_OK) return rc;
        }
        if (!do_h2x) continue;
        if (sync) {
            // (projections taken at the top of the layer)
        } else if (l + 1 < Lc) {
            // the h2x stage of this layer and the x2h stage of the next project the same h: one launch
            ProfScope ps(PC_NODE, s);
            if ((rc = td_launch_node_proj_pair(L.nodeH2x, level_rows(1), level_count(1), w.lig_node, Nl, w.Px, w.qx,
                                               m->layers[l + 1].nodeX2h, proj_rows(l + 1), proj_count(l + 1), w.P, w.q, h, N,
                                               s)) != TD_OK) return rc;
            proj_done = true;
        } else {
            if ((rc = h2x_project(L, w, h, N, Nl, w.Px, w.qx, level_rows(1), level_count(1), s)) != TD_OK) return rc;
        }
        if (L.ew_h2x) {              // the h2x stage's own gate, from the same (not yet updated) coordinates
            ProfScope ps(PC_GATE, s);
            if ((rc = td_launch_layer_gate(L.ew_h2x, L.offsets, L.coeff, xc, gt.nbr, N, gt.ew, s)) != TD_OK) return rc;
        }
        if ((rc = h2x_attend(m, L, w, gt, Nl, xc, xn, w.Px, w.qx, s)) != TD_OK) return rc;
        float4 *t = xc; xc = xn; xn = t;
    }
    *x_final = xc;
    return TD_OK;
}

// The same on a model with several stages per layer (num_x2h / num_h2x != 1; models/uni_transformer.py:190-206): the x2h stages of a layer one
// after the other on the layer's start coordinates, then its h2x stages, each on the coordinates the previous one left and all on the last
// x2h stage's features.  Every stage takes its own projections (nothing is fused across stages); every row, no caching.
int run_backbone_stages(const td_model *m, Workspace &w, const GraphTab &gt, float *h, int64_t N, int64_t Nl, int fix_x,
                        float4 **x_final, hipStream_t s, bool init_xn) {
    int rc;
    const int Lc = m->cfg.num_layers, NX = num_x2h(m->cfg), NH = num_h2x(m->cfg), M = stage_rows(m->cfg);
    float4 *xc = w.x4a, *xn = w.x4b;
    const bool do_h2x = !fix_x && Nl > 0;
    if (do_h2x && init_xn) TD_CHECK_HIP(hipMemcpyAsync(xn, xc, (size_t)N * sizeof(float4), hipMemcpyDeviceToDevice, s));
    for (int l = 0; l < Lc; ++l) {
        for (int i = 0; i < NX; ++i) {
            const TdLayer &L = m->layers[(size_t)l * M + i];
            { ProfScope ps(PC_NODE, s); if ((rc = td_launch_node_proj(L.nodeX2h, h, N, nullptr, 0x1f, w.P, w.q, s, nullptr)) != TD_OK) return rc; }
            if (L.ew_x2h) { ProfScope ps(PC_GATE, s); if ((rc = td_launch_layer_gate(L.ew_x2h, L.offsets, L.coeff, xc, gt.nbr, N, gt.ew, s)) != TD_OK) return rc; }
            { ProfScope ps(PC_X2H_K, s); if ((rc = key_pass(L.hk, L, xc, gt, gt.ew, gt.nbr, w.P, w.q, nullptr, nullptr, N, gt.alpha, s, w.lig_node, Nl)) != TD_OK) return rc; }
            float *att_out = L.nodeOut.B ? w.q : nullptr;
            { ProfScope ps(PC_X2H_V, s); if ((rc = value_pass(L.hv, L, xc, gt, gt.nbr, w.P, nullptr, nullptr, N, h, gt.alpha, w.lig_node, Nl, s, att_out)) != TD_OK) return rc; }
            if (att_out) { ProfScope ps(PC_NODE, s); if ((rc = td_launch_node_output(L.nodeOut, att_out, h, N, s)) != TD_OK) return rc; }
        }
        if (!do_h2x) continue;
        for (int j = 0; j < NH; ++j) {
            const TdLayer &L = m->layers[(size_t)l * M + j];
            if ((rc = h2x_project(L, w, h, N, Nl, w.Px, w.qx, nullptr, nullptr, s)) != TD_OK) return rc;
            if (L.ew_h2x) { ProfScope ps(PC_GATE, s); if ((rc = td_launch_layer_gate(L.ew_h2x, L.offsets, L.coeff, xc, gt.nbr, N, gt.ew, s)) != TD_OK) return rc; }
            if ((rc = h2x_attend(m, L, w, gt, Nl, xc, xn, w.Px, w.qx, s)) != TD_OK) return rc;
            float4 *t = xc; xc = xn; xn = t;
        }
    }
    *x_final = xc;
    return TD_OK;
}

inline int num_blocks(const td_config &c) { return c.num_blocks > 1 ? c.num_blocks : 1; }

// graph + edge gate of a composed batch on the default graph (32-slot rows: k-NN with k <= 32, radius with cap <= 32)
int build_default_graph(const td_model *m, Workspace &w, int64_t N, int max_graph_nodes, hipStream_t s) {
    int rc;
    {
        ProfScope ps(PC_KNN, s);
        if (m->cfg.cutoff_mode == TD_CUTOFF_RADIUS)
            rc = td_launch_radius32(w.x4a, w.node_ptr, w.gid, N, m->cfg.radius, m->cfg.max_num_neighbors, w.nbr, s);
        else
            rc = td_launch_knn(w.x4a, w.node_ptr, w.gid, N, max_graph_nodes, w.nbr, s, m->cfg.knn);
        if (rc != TD_OK) return rc;
    }
    if (m->cfg.ew_net_type != 0) return TD_OK;          // 'r' / none: every stage computes its own gate (run_backbone)
    ProfScope ps(PC_GATE, s);
    return td_launch_gate(m->gate, w.x4a, w.nbr, N, nullptr, nullptr, w.ew, s);
}

// ------------------------------------------------------------------------------------------ general graphs
// Layout of the chunked neighbour table of one batch (graph.hip "general graphs") plus the chunk-indexed buffers.  The
// layout depends only on the per-graph atom counts, so a sampling session builds it once; a stateless call builds and
// frees one per call (one host round trip for the counts -- these are the non-default graph modes).
struct GraphPlan {
    int mode = 0, k = 0, cpn_p = 1;
    float radius = 0.f;
    int64_t N = 0, Np = 0, Nl = 0, B = 0, NC = 0, NCl = 0;
    char *block = nullptr;
    int32_t *cptr = nullptr, *chunk_node = nullptr, *lig_chunks = nullptr, *cnbr = nullptr, *prot_node = nullptr, *pptr = nullptr, *meta = nullptr;
    float *ew = nullptr, *alpha = nullptr;
};

void free_async_or_sync(void *p, hipStream_t s);
// td_debug_fail_alloc: the n-th stream-ordered allocation from now fails (fault injection for the error paths)
std::atomic<int> g_fail_alloc{0};
hipError_t td_malloc_async(void **p, size_t bytes, hipStream_t s) {
    int n = g_fail_alloc.load(std::memory_order_relaxed);
    while (n > 0 && !g_fail_alloc.compare_exchange_weak(n, n - 1)) {}
    if (n == 1) { *p = nullptr; return hipErrorOutOfMemory; }
    return hipMallocAsync(p, bytes, s);
}
// stream-ordered scratch of one call: every block taken so far is given back on every exit path (a failing second or third
// allocation used to leak the earlier ones)
struct AsyncScratch {
    hipStream_t s;
    std::vector<void *> blocks;
    explicit AsyncScratch(hipStream_t st) : s(st) {}
    AsyncScratch(const AsyncScratch &) = delete;
    AsyncScratch &operator=(const AsyncScratch &) = delete;
    ~AsyncScratch() { for (void *b : blocks) free_async_or_sync(b, s); }
    template <class T>
    int take(T **out, size_t bytes, const char *who) {
        void *p = nullptr;
        hipError_t e = td_malloc_async(&p, bytes ? bytes : 4, s);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            td_set_error("%s: hipMallocAsync(%zu) failed: %s", who, bytes, hipGetErrorString(e));
            return TD_ENOMEM;
        }
        blocks.push_back(p);
        *out = static_cast<T *>(p);
        return TD_OK;
    }
};
void plan_destroy(GraphPlan &p, hipStream_t s) {
    free_async_or_sync(p.block, s);
    p.block = nullptr;
}

// host_pptr / host_lptr: [B+1] prefix offsets of the protein / ligand atoms (host copies)
int plan_create(const td_config &c, const int32_t *host_pptr, const int32_t *host_lptr, int64_t B, hipStream_t s, GraphPlan *out) {
    GraphPlan p;
    p.mode = c.cutoff_mode;
    p.k = c.cutoff_mode == TD_CUTOFF_RADIUS ? c.max_num_neighbors : c.knn;
    p.radius = c.radius;
    p.cpn_p = (p.k + TD_K - 1) / TD_K;
    p.B = B;
    p.Np = host_pptr[B]; p.Nl = host_lptr[B]; p.N = p.Np + p.Nl;
    std::vector<int32_t> meta((size_t)3 * (B + 1));
    int32_t *g_cbase = meta.data(), *g_cl = g_cbase + (B + 1), *g_lbase = g_cl + (B + 1);
    int64_t nc = 0, ncl = 0;
    for (int64_t g = 0; g < B; ++g) {
        const int np = host_pptr[g + 1] - host_pptr[g], nl = host_lptr[g + 1] - host_lptr[g];
        int cl = p.cpn_p;
        if (p.mode == TD_CUTOFF_HYBRID) { cl = (nl - 1 + p.k + TD_K - 1) / TD_K; if (cl < 1) cl = 1; }
        g_cbase[g] = (int32_t)nc; g_cl[g] = cl; g_lbase[g] = (int32_t)ncl;
        nc += (int64_t)np * p.cpn_p + (int64_t)nl * cl;
        ncl += (int64_t)nl * cl;
    }
    g_cbase[B] = (int32_t)nc; g_cl[B] = 0; g_lbase[B] = (int32_t)ncl;
    if (nc > 0x7fffffff / TD_K) { td_set_error("graph plan: %lld chunks overflow the 32-bit slot index", (long long)nc); return TD_EINVAL; }
    p.NC = nc; p.NCl = ncl;
    size_t off = 0;
    auto reserve = [&](size_t n) { size_t o = off; off += align_up(n ? n : 4); return o; };
    const size_t o_cptr = reserve((size_t)(p.N + 1) * 4), o_cn = reserve((size_t)nc * 4), o_lc = reserve((size_t)ncl * 4),
                 o_nbr = reserve((size_t)nc * TD_K * 4), o_ew = reserve((size_t)nc * TD_K * 4),
                 o_al = reserve((size_t)nc * TD_HEADS * TD_K * 4), o_pn = reserve((size_t)p.Np * 4),
                 o_pp = reserve((size_t)(B + 1) * 4), o_meta = reserve(meta.size() * 4);
    hipError_t e = td_malloc_async(reinterpret_cast<void **>(&p.block), off, s);
    if (e != hipSuccess) { td_set_error("graph plan: hipMallocAsync(%zu) failed: %s", off, hipGetErrorString(e)); return TD_ENOMEM; }
    char *b = p.block;
    p.cptr = reinterpret_cast<int32_t *>(b + o_cptr); p.chunk_node = reinterpret_cast<int32_t *>(b + o_cn);
    p.lig_chunks = reinterpret_cast<int32_t *>(b + o_lc); p.cnbr = reinterpret_cast<int32_t *>(b + o_nbr);
    p.ew = reinterpret_cast<float *>(b + o_ew); p.alpha = reinterpret_cast<float *>(b + o_al);
    p.prot_node = reinterpret_cast<int32_t *>(b + o_pn); p.pptr = reinterpret_cast<int32_t *>(b + o_pp);
    p.meta = reinterpret_cast<int32_t *>(b + o_meta);
    // the small per-graph tables: synchronous copies (the source vectors die with this frame)
    e = hipMemcpyAsync(b + o_meta, meta.data(), meta.size() * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(p.pptr, host_pptr, (size_t)(B + 1) * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        td_set_error("graph plan: copying the per-graph tables failed: %s", hipGetErrorString(e));
        plan_destroy(p, s);
        return TD_EHIP;
    }
    *out = p;
    return TD_OK;
}

// cptr / chunk_node / lig_chunks from the per-graph tables (needs node_ptr and gid of the composed batch)
int plan_layout(GraphPlan &p, const int32_t *node_ptr, const int32_t *gid, hipStream_t s) {
    const int32_t *meta = p.meta;
    return td_launch_layout(node_ptr, p.pptr, gid, meta, meta + (p.B + 1), meta + 2 * (p.B + 1), p.cpn_p, p.N, p.cptr,
                            p.chunk_node, p.lig_chunks, (int32_t)p.NC, s);
}

GraphTab plan_tab(const GraphPlan &p) { return GraphTab{p.cptr, p.cnbr, p.ew, p.alpha, p.cpn_p, p.NCl, nullptr}; }

// graph + edge gate of a composed batch on a general graph (chunked table of the plan)
int build_general_graph(const td_model *m, GraphPlan &p, Workspace &w, int64_t N, int64_t Nl, int max_graph_nodes, hipStream_t s) {
    int rc;
    {
        ProfScope ps(PC_KNN, s);
        if ((rc = td_launch_graph_general(p.mode, w.x4a, w.node_ptr, p.pptr, w.gid, p.prot_node, p.Np, w.lig_node, Nl, N, p.k,
                                          p.radius, max_graph_nodes, p.cptr, p.cnbr, p.NC, s)) != TD_OK) return rc;
    }
    ProfScope ps(PC_GATE, s);
    return td_launch_gate(m->gate, w.x4a, p.cnbr, p.NC, nullptr, nullptr, p.ew, s, p.chunk_node);
}

// Plan of a composed batch given as (mask_ligand, node_ptr) -- the refine_net seam: per-graph protein / ligand counts and
// the protein row list are derived on the host (one round trip; compose_context order = protein rows first is required).
int plan_from_mask(const td_config &c, const uint8_t *d_mask, const int32_t *d_node_ptr, int64_t N, int64_t B, hipStream_t s,
                   GraphPlan *out, int64_t *nl_out) {
    std::vector<uint8_t> mask((size_t)N);
    std::vector<int32_t> nptr((size_t)B + 1);
    TD_CHECK_HIP(hipMemcpyAsync(mask.data(), d_mask, (size_t)N, hipMemcpyDeviceToHost, s));
    TD_CHECK_HIP(hipMemcpyAsync(nptr.data(), d_node_ptr, (size_t)(B + 1) * 4, hipMemcpyDeviceToHost, s));
    TD_CHECK_HIP(hipStreamSynchronize(s));
    std::vector<int32_t> hp((size_t)B + 1, 0), hl((size_t)B + 1, 0), prot;
    prot.reserve((size_t)N);
    for (int64_t g = 0; g < B; ++g) {
        int np = 0, nl = 0;
        for (int i = nptr[g]; i < nptr[g + 1]; ++i) {
            if (mask[(size_t)i]) ++nl;
            else {
                if (nl) { td_set_error("general graphs need compose_context order (protein rows first inside every graph)"); return TD_EINVAL; }
                ++np;
                prot.push_back(i);
            }
        }
        hp[g + 1] = hp[g] + np; hl[g + 1] = hl[g] + nl;
    }
    int rc = plan_create(c, hp.data(), hl.data(), B, s, out);
    if (rc != TD_OK) return rc;
    if (!prot.empty()) {
        hipError_t e = hipMemcpyAsync(out->prot_node, prot.data(), prot.size() * 4, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) {
            td_set_error("graph plan: copying the protein row list failed: %s", hipGetErrorString(e));
            plan_destroy(*out, s);
            return TD_EHIP;
        }
    }
    *nl_out = hl[B];
    return TD_OK;
}

// host copies of two [B+1] device arrays (one synchronisation)
int fetch_ptrs(const int32_t *d_a, const int32_t *d_b, int64_t B, std::vector<int32_t> &a, std::vector<int32_t> &b, hipStream_t s) {
    a.resize((size_t)B + 1); b.resize((size_t)B + 1);
    TD_CHECK_HIP(hipMemcpyAsync(a.data(), d_a, (size_t)(B + 1) * 4, hipMemcpyDeviceToHost, s));
    TD_CHECK_HIP(hipMemcpyAsync(b.data(), d_b, (size_t)(B + 1) * 4, hipMemcpyDeviceToHost, s));
    TD_CHECK_HIP(hipStreamSynchronize(s));
    return TD_OK;
}
}  // namespace

extern "C" size_t td_workspace_bytes(const td_model *m, int64_t N, int64_t B, int64_t N_l) {
    (void)m;
    return carve(nullptr, N, B, N_l).bytes;
}

// ------------------------------------------------------------------------------------------ entry points
extern "C" int td_graph_ptr(const int64_t *d_batch, int64_t N, int64_t B, int32_t *d_ptr, void *stream) {
    if (!d_ptr || (N > 0 && !d_batch) || N < 0 || B < 0) { td_set_error("td_graph_ptr: bad argument"); return TD_EINVAL; }
    return td_launch_graph_ptr(d_batch, N, B, d_ptr, static_cast<hipStream_t>(stream));
}

extern "C" int td_knn(const float *d_x, const int32_t *d_node_ptr, int64_t N, int64_t B, int32_t k,
                      int32_t max_graph_nodes, int32_t *d_out_nbr, void *stream) {
    if (k < 1 || k > TD_MAX_FANIN) { td_set_error("td_knn: k must be in 1..%d (got %d)", TD_MAX_FANIN, k); return TD_EINVAL; }
    if (N < 0 || B < 0 || (N > 0 && (!d_x || !d_node_ptr || !d_out_nbr))) { td_set_error("td_knn: bad argument"); return TD_EINVAL; }
    if (N == 0) return TD_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // scratch: float4 coordinates + graph ids (stream-ordered allocation keeps the call self-contained)
    AsyncScratch scratch(s);
    float4 *x4 = nullptr;
    int32_t *gid = nullptr;
    int rc;
    if ((rc = scratch.take(&x4, (size_t)N * sizeof(float4), "td_knn")) != TD_OK) return rc;
    if ((rc = scratch.take(&gid, (size_t)N * sizeof(int32_t), "td_knn")) != TD_OK) return rc;
    TD_CHECK_HIP(hipMemsetAsync(gid, 0, (size_t)N * sizeof(int32_t), s));
    // the ligand flag (.w) is irrelevant for the search: pack with an all-zero mask (gid is zero-filled scratch)
    rc = td_launch_pack_x(d_x, reinterpret_cast<const uint8_t *>(gid), N, x4, s);
    if (rc == TD_OK) rc = td_launch_node_gid(d_node_ptr, N, B, gid, s);
    if (rc == TD_OK && k == TD_K) rc = td_launch_knn(x4, d_node_ptr, gid, N, max_graph_nodes, d_out_nbr, s);
    else if (rc == TD_OK) {
        // any other k: through the chunked table of the general-graph path (every node counts as "protein": one row kind)
        td_config c = {};
        c.cutoff_mode = TD_CUTOFF_KNN; c.knn = k;
        std::vector<int32_t> hp((size_t)B + 1), hl((size_t)B + 1, 0);
        hipError_t e = hipMemcpyAsync(hp.data(), d_node_ptr, (size_t)(B + 1) * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        GraphPlan p;
        if (e != hipSuccess) { td_set_error("td_knn: %s", hipGetErrorString(e)); rc = TD_EHIP; }
        else rc = plan_create(c, hp.data(), hl.data(), B, s, &p);
        if (rc == TD_OK) {
            rc = plan_layout(p, d_node_ptr, gid, s);
            if (rc == TD_OK) rc = td_launch_graph_general(TD_CUTOFF_KNN, x4, d_node_ptr, p.pptr, gid, nullptr, 0, nullptr, 0, N, k, 0.f,
                                                          max_graph_nodes, p.cptr, p.cnbr, p.NC, s);
            if (rc == TD_OK) rc = td_launch_slots_to_dense(p.cptr, p.cnbr, N, k, d_out_nbr, s);
            plan_destroy(p, s);
        }
    }
    return rc;
}

extern "C" int td_graph_build(const td_model *m, const float *d_x, const uint8_t *d_mask_ligand, const int32_t *d_node_ptr,
                              int64_t N, int64_t B, int32_t max_graph_nodes, int32_t *d_out_nbr, int32_t width, void *stream) {
    if (!m || N < 0 || B < 0 || width < 1 || (N > 0 && (!d_x || !d_mask_ligand || !d_node_ptr || !d_out_nbr))) {
        td_set_error("td_graph_build: bad argument");
        return TD_EINVAL;
    }
    if (N == 0) return TD_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    AsyncScratch scratch(s);
    float4 *x4 = nullptr;
    int32_t *gid = nullptr, *lig = nullptr;
    int rc;
    if ((rc = scratch.take(&x4, (size_t)N * sizeof(float4), "td_graph_build")) != TD_OK) return rc;
    if ((rc = scratch.take(&gid, (size_t)N * sizeof(int32_t), "td_graph_build")) != TD_OK) return rc;
    if ((rc = scratch.take(&lig, (size_t)(N + 1) * sizeof(int32_t), "td_graph_build")) != TD_OK) return rc;
    rc = td_launch_pack_x(d_x, d_mask_ligand, N, x4, s);
    if (rc == TD_OK) rc = td_launch_node_gid(d_node_ptr, N, B, gid, s);
    if (rc == TD_OK) rc = td_launch_ligand_list(d_mask_ligand, N, lig, lig + N, s);
    GraphPlan p;
    int64_t nl = 0;
    if (rc == TD_OK) rc = plan_from_mask(m->cfg, d_mask_ligand, d_node_ptr, N, B, s, &p, &nl);
    if (rc == TD_OK) {
        rc = plan_layout(p, d_node_ptr, gid, s);
        if (rc == TD_OK) rc = td_launch_graph_general(p.mode, x4, d_node_ptr, p.pptr, gid, p.prot_node, p.Np, lig, nl, N, p.k, p.radius,
                                                      max_graph_nodes, p.cptr, p.cnbr, p.NC, s);
        if (rc == TD_OK) rc = td_launch_slots_to_dense(p.cptr, p.cnbr, N, width, d_out_nbr, s);
        plan_destroy(p, s);
    }
    return rc;
}

extern "C" int td_refine_forward(const td_model *m, const float *d_h, const float *d_x, const uint8_t *d_mask_ligand,
                                 const int32_t *d_node_ptr, int64_t N, int64_t B, int32_t fix_x,
                                 int32_t max_graph_nodes, float *d_out_h, float *d_out_x, int32_t *d_out_nbr,
                                 float *d_out_ew, void *d_workspace, size_t workspace_bytes, void *stream) {
    if (!m || N < 0 || B < 0) { td_set_error("td_refine_forward: bad argument"); return TD_EINVAL; }
    if (N == 0) return TD_OK;
    if (!d_h || !d_x || !d_mask_ligand || !d_node_ptr || !d_out_h || !d_out_x || !d_workspace) {
        td_set_error("td_refine_forward: null pointer");
        return TD_EINVAL;
    }
    Workspace w = carve(static_cast<char *>(d_workspace), N, B, 0);
    if (w.bytes > workspace_bytes) {
        td_set_error("td_refine_forward: workspace has %zu bytes, need %zu", workspace_bytes, w.bytes);
        return TD_ENOMEM;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc;
    TD_CHECK_HIP(hipMemcpyAsync(w.node_ptr, d_node_ptr, (size_t)(B + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    if ((rc = td_launch_node_gid(w.node_ptr, N, B, w.gid, s)) != TD_OK) return rc;
    if ((rc = td_launch_pack_x(d_x, d_mask_ligand, N, w.x4a, s)) != TD_OK) return rc;
    // ligand row list (h2x destinations; the x2h value pass serves the ligand rows from it); its length is needed for the
    // launch shapes -> one small D2H (this entry point mirrors the refine_net seam; the sampler path uses td_model_forward,
    // which knows N_l on the host).
    int32_t nl = 0;
    {
        if ((rc = td_launch_ligand_list(d_mask_ligand, N, w.lig_node, w.lig_node + N, s)) != TD_OK) return rc;
        TD_CHECK_HIP(hipMemcpyAsync(&nl, w.lig_node + N, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        TD_CHECK_HIP(hipStreamSynchronize(s));
    }
    if (d_out_h != d_h) TD_CHECK_HIP(hipMemcpyAsync(d_out_h, d_h, (size_t)N * TD_H * sizeof(float), hipMemcpyDeviceToDevice, s));
    float4 *xf = nullptr;
    if (!default_graph(m->cfg)) {
        if (d_out_nbr || d_out_ew) { td_set_error("td_refine_forward: graph / gate outputs exist for the k = 32 kNN graph only (use td_graph_build)"); return TD_EINVAL; }
        GraphPlan p;
        int64_t nl_all = 0;
        if ((rc = plan_from_mask(m->cfg, d_mask_ligand, w.node_ptr, N, B, s, &p, &nl_all)) != TD_OK) return rc;
        rc = plan_layout(p, w.node_ptr, w.gid, s);
        // every block: graph + gate from the current coordinates (in w.x4a), then the layer stack (models/uni_transformer.py:306-323)
        for (int blk = 0; rc == TD_OK && blk < num_blocks(m->cfg); ++blk) {
            rc = build_general_graph(m, p, w, N, nl_all, max_graph_nodes, s);
            if (rc == TD_OK) rc = run_backbone(m, w, plan_tab(p), d_out_h, N, nl_all, fix_x, &xf, s, true);
            if (rc == TD_OK && xf == w.x4b) std::swap(w.x4a, w.x4b);
        }
        if (rc == TD_OK) rc = td_launch_unpack_x(xf, N, d_out_x, s);
        plan_destroy(p, s);
        return rc;
    }
    for (int blk = 0; blk < num_blocks(m->cfg); ++blk) {
        if ((rc = build_default_graph(m, w, N, max_graph_nodes, s)) != TD_OK) return rc;
        if ((rc = run_backbone(m, w, default_tab(w), d_out_h, N, nl, fix_x, &xf, s, true)) != TD_OK) return rc;
        if (xf == w.x4b) std::swap(w.x4a, w.x4b);
    }
    if ((rc = td_launch_unpack_x(xf, N, d_out_x, s)) != TD_OK) return rc;
    if (d_out_nbr) TD_CHECK_HIP(hipMemcpyAsync(d_out_nbr, w.nbr, (size_t)N * TD_K * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    if (d_out_ew) TD_CHECK_HIP(hipMemcpyAsync(d_out_ew, w.ew, (size_t)N * TD_K * sizeof(float), hipMemcpyDeviceToDevice, s));
    return TD_OK;
}

extern "C" int td_model_forward(const td_model *m, const float *d_protein_pos, const float *d_protein_v,
                                const int32_t *d_protein_ptr, int64_t N_p, const float *d_ligand_pos,
                                const int64_t *d_ligand_v, const int32_t *d_ligand_ptr, int64_t N_l, int64_t B,
                                int32_t fix_x, int32_t max_graph_nodes, float *d_pred_ligand_pos,
                                float *d_pred_ligand_v, float *d_final_ligand_h, float *d_final_h, void *d_workspace,
                                size_t workspace_bytes, const float *d_ligand_graph_bias, void *stream) {
    if (!m || N_p < 0 || N_l < 0 || B < 0) { td_set_error("td_model_forward: bad argument"); return TD_EINVAL; }
    const int64_t N = N_p + N_l;
    if (N == 0) return TD_OK;
    if (!d_protein_ptr || !d_ligand_ptr || !d_workspace || (N_p > 0 && (!d_protein_pos || !d_protein_v)) ||
        (N_l > 0 && (!d_ligand_pos || !d_ligand_v || !d_pred_ligand_pos || !d_pred_ligand_v))) {
        td_set_error("td_model_forward: null pointer");
        return TD_EINVAL;
    }
    Workspace w = carve(static_cast<char *>(d_workspace), N, B, N_l);
    if (w.bytes > workspace_bytes) {
        td_set_error("td_model_forward: workspace has %zu bytes, need %zu", workspace_bytes, w.bytes);
        return TD_ENOMEM;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc;
    float *h = d_final_h ? d_final_h : w.h;
    if (!default_graph(m->cfg)) {
        std::vector<int32_t> hp, hl;
        if ((rc = fetch_ptrs(d_protein_ptr, d_ligand_ptr, B, hp, hl, s)) != TD_OK) return rc;
        GraphPlan p;
        if ((rc = plan_create(m->cfg, hp.data(), hl.data(), B, s, &p)) != TD_OK) return rc;
        float4 *xg = nullptr;
        {
            ProfScope ps(PC_COMPOSE, s);
            rc = td_launch_compose(m, d_protein_pos, d_protein_v, d_protein_ptr, N_p, d_ligand_pos, d_ligand_v, d_ligand_ptr, N_l, B,
                                   h, w.x4a, w.node_ptr, w.gid, w.lig_node, p.prot_node, s, d_ligand_graph_bias);
        }
        if (rc == TD_OK) rc = plan_layout(p, w.node_ptr, w.gid, s);
        for (int blk = 0; rc == TD_OK && blk < num_blocks(m->cfg); ++blk) {
            rc = build_general_graph(m, p, w, N, N_l, max_graph_nodes, s);
            if (rc == TD_OK) rc = run_backbone(m, w, plan_tab(p), h, N, N_l, fix_x, &xg, s, true);
            if (rc == TD_OK && xg == w.x4b) std::swap(w.x4a, w.x4b);
        }
        if (rc == TD_OK) {
            ProfScope ps(PC_HEAD, s);
            rc = td_launch_head(m->head, h, xg, w.lig_node, N_l, m->cfg.ligand_num_classes, d_pred_ligand_pos, d_pred_ligand_v,
                                d_final_ligand_h, s);
        }
        plan_destroy(p, s);
        return rc;
    }
    {
        ProfScope ps(PC_COMPOSE, s);
        if ((rc = td_launch_compose(m, d_protein_pos, d_protein_v, d_protein_ptr, N_p, d_ligand_pos, d_ligand_v,
                                    d_ligand_ptr, N_l, B, h, w.x4a, w.node_ptr, w.gid, w.lig_node, nullptr, s, d_ligand_graph_bias)) != TD_OK)
            return rc;
    }
    float4 *xf = nullptr;
    for (int blk = 0; blk < num_blocks(m->cfg); ++blk) {
        if ((rc = build_default_graph(m, w, N, max_graph_nodes, s)) != TD_OK) return rc;
        if ((rc = run_backbone(m, w, default_tab(w), h, N, N_l, fix_x, &xf, s, true)) != TD_OK) return rc;
        if (xf == w.x4b) std::swap(w.x4a, w.x4b);
    }
    ProfScope ps(PC_HEAD, s);
    return td_launch_head(m->head, h, xf, w.lig_node, N_l, m->cfg.ligand_num_classes, d_pred_ligand_pos,
                          d_pred_ligand_v, d_final_ligand_h, s);
}

extern "C" int td_posterior_step(const td_model *m, const int32_t *d_t, const int32_t *d_ligand_ptr, int64_t N_l,
                                 int64_t B, const float *d_ligand_pos, const int64_t *d_ligand_v,
                                 const float *d_pred_pos, const float *d_pred_v, const float *d_noise,
                                 const float *d_uniform, float *d_pos_next, int64_t *d_v_next, float *d_log_v0,
                                 float *d_log_post, void *stream) {
    if (!m || N_l < 0 || B < 0) { td_set_error("td_posterior_step: bad argument"); return TD_EINVAL; }
    if (N_l == 0) return TD_OK;
    if (!d_t || !d_ligand_ptr || !d_ligand_pos || !d_ligand_v || !d_pred_pos || !d_pred_v || !d_noise || !d_uniform ||
        !d_pos_next || !d_v_next) {
        td_set_error("td_posterior_step: null pointer");
        return TD_EINVAL;
    }
    ProfScope ps(PC_POST, static_cast<hipStream_t>(stream));
    return td_launch_posterior(m->sched, m->cfg.num_timesteps, d_t, d_ligand_ptr, N_l, B, m->cfg.ligand_num_classes,
                               d_ligand_pos, d_ligand_v, d_pred_pos, d_pred_v, d_noise, d_uniform, d_pos_next,
                               d_v_next, d_log_v0, d_log_post, static_cast<hipStream_t>(stream), m->cfg.model_mean_type);
}

// ------------------------------------------------------------------------------------------ standalone EGNN refine net
struct td_egnn {
    int num_layers;
    float *blob;
    TdEgnnLayer *layers;     // host array
};

namespace {
constexpr int EGNN_EDGE_IN = 2 * TD_H + 1 + 4;      // [h_i | h_j | d^2 | one_hot(type)]  (models/egnn.py:22, num_r_gaussian = 1)
size_t egnn_layer_floats() {
    return (size_t)TD_H * EGNN_EDGE_IN + TD_H + (size_t)TD_H * TD_H + TD_H + TD_H + 1 + (size_t)TD_H * TD_H + TD_H + TD_H +
           (size_t)TD_H * 2 * TD_H + TD_H + (size_t)TD_H * TD_H + TD_H;
}
// 128 x 128 weight (row-major [out][in]) as A fragments of the 16x16x4 product: [ot][hb][lane] x 4 r
size_t pack_A16(Packer &pk, const float *W) {
    size_t off = pk.alloc((size_t)8 * 8 * 64 * 4);
    float *d = pk.data.data() + off;
    for (int ot = 0; ot < 8; ++ot)
        for (int hb = 0; hb < 8; ++hb)
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 4; ++r)
                    d[(((size_t)ot * 8 + hb) * 64 + lane) * 4 + r] = W[(size_t)(16 * ot + (lane & 15)) * TD_H + 16 * hb + 4 * (lane >> 4) + r];
    return off;
}
}  // namespace

extern "C" size_t td_egnn_num_weights(int32_t num_layers) { return num_layers > 0 ? (size_t)num_layers * egnn_layer_floats() : 0; }

extern "C" int td_egnn_create(int32_t num_layers, int32_t hidden_dim, int32_t edge_feat_dim, int32_t knn,
                              const float *host_weights, size_t num_weights, td_egnn **out) {
    if (!host_weights || !out || num_layers <= 0) { td_set_error("td_egnn_create: bad argument"); return TD_EINVAL; }
    if (hidden_dim != TD_H || edge_feat_dim != 4 || knn != TD_K) {
        td_set_error("td_egnn_create: unsupported configuration (need hidden 128, edge_feat_dim 4, knn 32; got %d/%d/%d)",
                     hidden_dim, edge_feat_dim, knn);
        return TD_EINVAL;
    }
    if (num_weights != td_egnn_num_weights(num_layers)) {
        td_set_error("td_egnn_create: weight blob has %zu floats, expected %zu", num_weights, td_egnn_num_weights(num_layers));
        return TD_EINVAL;
    }
    struct Off { size_t projB, projBias, W2f, Wxf, vec, nodeB, nb1, nb2; };
    std::vector<Off> off((size_t)num_layers);
    Packer pk;
    Cursor cur{host_weights, num_weights};
    for (int l = 0; l < num_layers; ++l) {
        const float *W1 = cur.take((size_t)TD_H * EGNN_EDGE_IN), *b1 = cur.take(TD_H);
        const float *W2 = cur.take((size_t)TD_H * TD_H), *b2 = cur.take(TD_H);
        const float *winf = cur.take(TD_H), *binf = cur.take(1);
        const float *Wx = cur.take((size_t)TD_H * TD_H), *bx = cur.take(TD_H), *wx2 = cur.take(TD_H);
        const float *Wn1 = cur.take((size_t)TD_H * 2 * TD_H), *bn1 = cur.take(TD_H);
        const float *Wn2 = cur.take((size_t)TD_H * TD_H), *bn2 = cur.take(TD_H);
        Off &o = off[(size_t)l];
        o.projB = pack_B128(pk, W1, EGNN_EDGE_IN, 0);          // h_i columns (dst)
        pack_B128(pk, W1, EGNN_EDGE_IN, TD_H);                 // h_j columns (src): consecutive block
        o.projBias = pk.alloc(5 * TD_H);
        memcpy(pk.data.data() + o.projBias, b1, TD_H * sizeof(float));
        o.W2f = pack_A16(pk, W2);
        o.Wxf = pack_A16(pk, Wx);
        o.vec = pk.alloc(128 + 512 + 128 + 132 + 128 + 128);
        float *v = pk.data.data() + o.vec;
        for (int n = 0; n < TD_H; ++n) {
            v[n] = W1[(size_t)n * EGNN_EDGE_IN + 2 * TD_H];                                    // d^2 column
            for (int t = 0; t < 4; ++t) v[128 + t * TD_H + n] = W1[(size_t)n * EGNN_EDGE_IN + 2 * TD_H + 1 + t];
            v[640 + n] = b2[n];
            v[768 + n] = winf[n];
            v[900 + n] = bx[n];
            v[1028 + n] = wx2[n];
        }
        v[768 + 128] = binf[0];
        o.nodeB = pack_B128(pk, Wn1, 2 * TD_H, 0);             // mi half of node_mlp.net.0  (cat([mi, h]), models/egnn.py:56)
        pack_B128(pk, Wn1, 2 * TD_H, TD_H);                    // h half
        pack_B128(pk, Wn2, TD_H, 0);
        o.nb1 = pack_vec(pk, bn1, TD_H);
        o.nb2 = pack_vec(pk, bn2, TD_H);
    }
    if (!cur.ok || cur.left != 0) { td_set_error("td_egnn_create: weight blob layout mismatch"); return TD_EINVAL; }
    td_egnn *m = new (std::nothrow) td_egnn();
    if (!m) { td_set_error("td_egnn_create: out of host memory"); return TD_ENOMEM; }
    m->num_layers = num_layers;
    m->layers = new (std::nothrow) TdEgnnLayer[(size_t)num_layers];
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&m->blob), pk.data.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(m->blob, pk.data.data(), pk.data.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess || !m->layers) {
        td_set_error("td_egnn_create: device upload failed: %s", hipGetErrorString(e));
        if (m->blob) (void)hipFree(m->blob);
        delete[] m->layers;
        delete m;
        return TD_EHIP;
    }
    const float *D = m->blob;
    for (int l = 0; l < num_layers; ++l) {
        const Off &o = off[(size_t)l];
        TdEgnnLayer &L = m->layers[l];
        L.proj = TdNodeStage{D + o.projB, D + o.projBias, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, false, false, false};
        L.W2f = D + o.W2f; L.Wxf = D + o.Wxf; L.vec = D + o.vec; L.nodeB = D + o.nodeB; L.nb1 = D + o.nb1; L.nb2 = D + o.nb2;
    }
    *out = m;
    return TD_OK;
}

extern "C" void td_egnn_destroy(td_egnn *m) {
    if (!m) return;
    if (m->blob) (void)hipFree(m->blob);
    delete[] m->layers;
    delete m;
}

namespace {
struct EgnnWs { float4 *x4a, *x4b; int32_t *gid, *nbr; float *P, *mi; size_t bytes; };
EgnnWs egnn_carve(char *base, int64_t N) {
    EgnnWs w;
    size_t off = 0;
    auto take = [&](size_t n) { char *p = base ? base + off : nullptr; off += align_up(n); return p; };
    const size_t n = (size_t)(N > 0 ? N : 1);
    w.x4a = reinterpret_cast<float4 *>(take(n * sizeof(float4)));
    w.x4b = reinterpret_cast<float4 *>(take(n * sizeof(float4)));
    w.gid = reinterpret_cast<int32_t *>(take(n * sizeof(int32_t)));
    w.nbr = reinterpret_cast<int32_t *>(take(n * TD_K * sizeof(int32_t)));
    w.P = reinterpret_cast<float *>(take(n * 4 * TD_H * sizeof(float)));
    w.mi = reinterpret_cast<float *>(take(n * TD_H * sizeof(float)));
    w.bytes = off;
    return w;
}
}  // namespace

extern "C" size_t td_egnn_workspace_bytes(int64_t N) { return egnn_carve(nullptr, N).bytes; }

// EGNN.forward (models/egnn.py:121-133): per layer a fresh kNN graph on the current coordinates, then one EnBaseLayer.
extern "C" int td_egnn_forward(const td_egnn *m, const float *d_h, const float *d_x, const uint8_t *d_mask_ligand,
                               const int32_t *d_node_ptr, int64_t N, int64_t B, int32_t max_graph_nodes, float *d_out_h,
                               float *d_out_x, float *d_all_h, float *d_all_x, void *d_workspace, size_t workspace_bytes,
                               void *stream) {
    if (!m || N < 0 || B < 0) { td_set_error("td_egnn_forward: bad argument"); return TD_EINVAL; }
    if (N == 0) return TD_OK;
    if (!d_h || !d_x || !d_mask_ligand || !d_node_ptr || !d_out_h || !d_out_x || !d_workspace) {
        td_set_error("td_egnn_forward: null pointer");
        return TD_EINVAL;
    }
    EgnnWs w = egnn_carve(static_cast<char *>(d_workspace), N);
    if (w.bytes > workspace_bytes) {
        td_set_error("td_egnn_forward: workspace has %zu bytes, need %zu", workspace_bytes, w.bytes);
        return TD_ENOMEM;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc;
    if ((rc = td_launch_node_gid(d_node_ptr, N, B, w.gid, s)) != TD_OK) return rc;
    if ((rc = td_launch_pack_x(d_x, d_mask_ligand, N, w.x4a, s)) != TD_OK) return rc;
    TD_CHECK_HIP(hipMemcpyAsync(w.x4b, w.x4a, (size_t)N * sizeof(float4), hipMemcpyDeviceToDevice, s));
    if (d_out_h != d_h) TD_CHECK_HIP(hipMemcpyAsync(d_out_h, d_h, (size_t)N * TD_H * sizeof(float), hipMemcpyDeviceToDevice, s));
    float4 *xc = w.x4a, *xn = w.x4b;
    for (int l = 0; l < m->num_layers; ++l) {
        const TdEgnnLayer &L = m->layers[l];
        if ((rc = td_launch_knn(xc, d_node_ptr, w.gid, N, max_graph_nodes, w.nbr, s)) != TD_OK) return rc;
        if ((rc = td_launch_node_proj(L.proj, d_out_h, N, nullptr, 0x03, w.P, w.P, s)) != TD_OK) return rc;
        if ((rc = td_launch_egnn_edge(L, xc, xn, w.nbr, w.P, w.mi, N, s)) != TD_OK) return rc;
        if ((rc = td_launch_egnn_node(L, w.mi, d_out_h, N, s)) != TD_OK) return rc;
        float4 *t = xc; xc = xn; xn = t;
        // keep the protein rows of the (now stale) buffer in sync is not needed: only ligand rows ever change and the
        // edge kernel rewrites every ligand row of its output buffer
        if (d_all_x && (rc = td_launch_unpack_x(xc, N, d_all_x + (size_t)l * N * 3, s)) != TD_OK) return rc;
        if (d_all_h) TD_CHECK_HIP(hipMemcpyAsync(d_all_h + (size_t)l * N * TD_H, d_out_h, (size_t)N * TD_H * sizeof(float),
                                                 hipMemcpyDeviceToDevice, s));
    }
    return td_launch_unpack_x(xc, N, d_out_x, s);
}

// ---- the other forward consumers: likelihood estimation (scripts/likelihood_est_diffusion.py) and return_all
extern "C" int td_perturb(const td_model *m, const int32_t *d_t, const int32_t *d_ligand_ptr, int64_t N_l, int64_t B,
                          const float *d_ligand_pos, const int64_t *d_ligand_v, const float *d_noise, const float *d_uniform,
                          float *d_pos_t, int64_t *d_v_t, void *stream) {
    if (!m || N_l < 0 || B < 0) { td_set_error("td_perturb: bad argument"); return TD_EINVAL; }
    if (!m->sched.abar) { td_set_error("td_perturb: the model was created without alphas_cumprod (8 schedule arrays)"); return TD_EINVAL; }
    if (N_l == 0) return TD_OK;
    if (!d_t || !d_ligand_ptr || !d_ligand_pos || !d_ligand_v || !d_noise || !d_uniform || !d_pos_t || !d_v_t) {
        td_set_error("td_perturb: null pointer");
        return TD_EINVAL;
    }
    return td_launch_perturb(m->sched, m->cfg.num_timesteps, d_t, d_ligand_ptr, N_l, B, m->cfg.ligand_num_classes, d_ligand_pos,
                             d_ligand_v, d_noise, d_uniform, d_pos_t, d_v_t, static_cast<hipStream_t>(stream));
}

extern "C" int td_likelihood_terms(const td_model *m, const int32_t *d_t, const int32_t *d_ligand_ptr, int64_t N_l, int64_t B,
                                   const float *d_pos_0, const float *d_pos_t, const int64_t *d_v_0, const int64_t *d_v_t,
                                   const float *d_pred_pos, const float *d_pred_v, float *d_kl_pos, float *d_kl_v,
                                   void *stream) {
    if (!m || N_l < 0 || B < 0) { td_set_error("td_likelihood_terms: bad argument"); return TD_EINVAL; }
    if (B == 0) return TD_OK;
    if (!d_t || !d_ligand_ptr || !d_kl_pos || !d_kl_v ||
        (N_l > 0 && (!d_pos_0 || !d_pos_t || !d_v_0 || !d_v_t || !d_pred_pos || !d_pred_v))) {
        td_set_error("td_likelihood_terms: null pointer");
        return TD_EINVAL;
    }
    return td_launch_likelihood_terms(m->sched, m->cfg.num_timesteps, d_t, d_ligand_ptr, B, m->cfg.ligand_num_classes, d_pos_0,
                                      d_pos_t, d_v_0, d_v_t, d_pred_pos, d_pred_v, d_kl_pos, d_kl_v,
                                      static_cast<hipStream_t>(stream));
}

extern "C" int td_likelihood_prior(const td_model *m, const int32_t *d_ligand_ptr, int64_t N_l, int64_t B,
                                   const float *d_pos_0, const int64_t *d_v_index, float *d_kl_pos, float *d_kl_v,
                                   void *stream) {
    if (!m || N_l < 0 || B < 0) { td_set_error("td_likelihood_prior: bad argument"); return TD_EINVAL; }
    if (!m->sched.abar) { td_set_error("td_likelihood_prior: the model was created without alphas_cumprod (8 schedule arrays)"); return TD_EINVAL; }
    if (B == 0) return TD_OK;
    if (!d_ligand_ptr || !d_kl_pos || !d_kl_v || (N_l > 0 && (!d_pos_0 || !d_v_index))) {
        td_set_error("td_likelihood_prior: null pointer");
        return TD_EINVAL;
    }
    return td_launch_likelihood_prior(m->sched, m->cfg.num_timesteps, d_ligand_ptr, B, m->cfg.ligand_num_classes, d_pos_0,
                                      d_v_index, d_kl_pos, d_kl_v, static_cast<hipStream_t>(stream));
}

extern "C" int td_embed_ligand(const td_model *m, const int64_t *d_ligand_v, int64_t N_l, float *d_h, void *stream) {
    if (!m || N_l < 0 || (N_l > 0 && (!d_ligand_v || !d_h))) { td_set_error("td_embed_ligand: bad argument"); return TD_EINVAL; }
    return td_launch_embed_ligand(m->emb, m->cfg.ligand_num_classes, d_ligand_v, N_l, d_h, static_cast<hipStream_t>(stream));
}

extern "C" int td_v_inference(const td_model *m, const float *d_h, int64_t n, float *d_logits, void *stream) {
    if (!m || n < 0 || (n > 0 && (!d_h || !d_logits))) { td_set_error("td_v_inference: bad argument"); return TD_EINVAL; }
    return td_launch_head(m->head, d_h, nullptr, nullptr, n, m->cfg.ligand_num_classes, nullptr, d_logits, nullptr,
                          static_cast<hipStream_t>(stream));
}

extern "C" int td_center_pos(float *d_protein_pos, const int32_t *d_protein_ptr, float *d_ligand_pos,
                             const int32_t *d_ligand_ptr, int64_t B, float *d_offset, int32_t compute_offset,
                             int32_t sign, void *stream) {
    if (B < 0 || !d_offset || !d_protein_ptr || !d_ligand_ptr || (compute_offset && !d_protein_pos)) {
        td_set_error("td_center_pos: bad argument");
        return TD_EINVAL;
    }
    return td_launch_center(d_protein_pos, d_protein_ptr, d_ligand_pos, d_ligand_ptr, B, d_offset, compute_offset,
                            sign, static_cast<hipStream_t>(stream));
}

extern "C" int td_debug_node_stage(const td_model *m, int32_t layer, int32_t stage, const float *d_h, int64_t N,
                                   float *d_P, float *d_q, void *stream) {
    if (!m || layer < 0 || layer >= m->cfg.num_layers || (stage != 0 && stage != 1) || N < 0 || (N > 0 && (!d_h || !d_P || !d_q))) {
        td_set_error("td_debug_node_stage: bad argument");
        return TD_EINVAL;
    }
    if (stage_rows(m->cfg) != 1) {       // m->layers holds stage_rows rows per reference layer: this hook addresses whole layers only
        td_set_error("td_debug_node_stage: models with several x2h / h2x stages per layer are not addressed by this hook");
        return TD_EINVAL;
    }
    const TdLayer &L = m->layers[layer];
    return td_launch_node_proj(stage == 0 ? L.nodeX2h : L.nodeH2x, d_h, N, nullptr, 0x1f, d_P, d_q, static_cast<hipStream_t>(stream));
}

extern "C" int td_debug_fail_alloc(int32_t nth) {
    g_fail_alloc.store(nth > 0 ? nth : 0);
    return TD_OK;
}

extern "C" int td_debug_wg_trace(uint64_t *d_buf, int32_t slots) {
    return td_set_wg_trace(reinterpret_cast<unsigned long long *>(d_buf), d_buf ? slots : 0);
}

extern "C" int td_debug_reductions(const float *d_in64, float *d_out6x64, void *stream) {
    return td_launch_reductions(d_in64, d_out6x64, static_cast<hipStream_t>(stream));
}

// ------------------------------------------------------------------------------------------ sampling session
// State of one ScorePosNet3D.sample_diffusion call (models/molopt_score_model.py:633-703).  Everything that depends
// only on the protein is loop-invariant there (protein_pos, protein_v, batch_protein are passed unchanged to every
// forward, :652-661; protein coordinates are never updated, models/uni_transformer.py:206) and is computed once:
// embeddings, protein-only sorted neighbour lists, and -- for protein atoms that no ligand atom displaces from
// their k-NN row ("clean" rows, 80-90 % of them) -- the edge gate row and the layer-0 x2h output.
// Three kinds:
//   CACHING, default graph    k-NN with k <= 32: one 32-slot row per node (the workspace's nbr / ew / alpha)
//   CACHING, general graph    k-NN with 32 < k <= 64 and `hybrid` (protein rows are plain k-NN rows there too): the same
//                             machinery on the chunked table of a GraphPlan -- 64 static keys per protein row, cached gate
//                             chunks, chunk-aware row lists, the chunk-loop edge kernels
//   PLAIN                     radius graphs (rows in index order: no sorted lists to merge into) and graphs too large for
//                             the row-list kernel's LDS flags: every step is a stateless forward on the session's layout
struct td_session {
    const td_model *m;
    int64_t N, Np, Nl, B;
    int max_graph_nodes;
    char *block;
    hipStream_t last_stream;     // stream of the latest call: the block is freed in its order
    Workspace w;                 // per-step buffers (x4a/x4b, gid, nbr, lig_node, node_ptr, ew, P, q, h, alpha)
    int32_t *prot_node, *pptr, *lptr, *snbr, *dirty_rows, *dirty_count, *hop_rows, *hop_count, *dirty_chunks;
    int hop_levels;
    unsigned long long *skeys;
    float *ews, *h0, *h1s, *h2s, *P0, *q0;
    uint8_t *flags2;
    int32_t *fwd_rows, *fwd_rest, *fwd_counts;
    bool use_fwd;
    uint8_t *clean;
    int graph_nodes_max;         // exact size of the largest graph -- sizes the LDS flags of td_launch_step_lists
    bool caching;                // static-protein caching + receptive-field pruning (false: PLAIN)
    bool chunked;                // general graph: the neighbour table lives in `plan`
    GraphPlan plan;
    // td_session_step: the denoiser's outputs of the step, and the step as a captured graph
    float *pred_pos, *pred_v;
    hipGraph_t graph;
    hipGraphExec_t graph_exec;
    td_step_io graph_io;         // the arguments the graph was captured with
    unsigned graph_epoch = 0;    // ... and the model's option epoch at that time
    int eager_steps;             // steps issued launch by launch so far (the first one also does the one-time kernel set-up)
    bool graph_failed, last_step_graph;
};

namespace {
constexpr int TD_STEP_LISTS_MAX_NODES = 12288;       // LDS flags of step_lists_kernel: 4 bytes per node of a graph, 48 KiB

GraphTab session_tab(td_session *S) {
    GraphTab gt = S->chunked ? plan_tab(S->plan) : default_tab(S->w);
    if (S->caching) gt.mixed = S->dirty_count;
    return gt;
}

// stream-ordered free with a fallback: the stream may have been destroyed by the caller in the meantime (a C-ABI user with its own
// hipStreamCreate / hipStreamDestroy) -- then synchronise the device and free synchronously instead of leaking the block
void free_async_or_sync(void *p, hipStream_t s) {
    if (!p) return;
    if (hipFreeAsync(p, s) == hipSuccess) return;
    (void)hipGetLastError();
    (void)hipDeviceSynchronize();
    (void)hipFree(p);
}

void session_drop_graph(td_session *S) {
    if (S->graph_exec) (void)hipGraphExecDestroy(S->graph_exec);
    if (S->graph) (void)hipGraphDestroy(S->graph);
    S->graph_exec = nullptr;
    S->graph = nullptr;
}

void session_free(td_session *S, hipStream_t s) {
    if (!S) return;
    session_drop_graph(S);
    if (S->chunked) plan_destroy(S->plan, s);
    free_async_or_sync(S->block, s);
    delete S;
}
}  // namespace

extern "C" int td_session_create(const td_model *m, const float *d_protein_pos, const float *d_protein_v,
                                 const int32_t *d_protein_ptr, int64_t N_p, const int32_t *d_ligand_ptr, int64_t N_l,
                                 int64_t B, int32_t max_graph_nodes, void *stream, td_session **out) {
    if (!m || !out || N_p <= 0 || N_l <= 0 || B <= 0 || !d_protein_pos || !d_protein_v || !d_protein_ptr || !d_ligand_ptr) {
        td_set_error("td_session_create: bad argument");
        return TD_EINVAL;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t N = N_p + N_l;
    td_session *S = new (std::nothrow) td_session();
    if (!S) { td_set_error("td_session_create: out of host memory"); return TD_ENOMEM; }
    S->m = m; S->N = N; S->Np = N_p; S->Nl = N_l; S->B = B; S->max_graph_nodes = max_graph_nodes; S->last_stream = s;
    S->chunked = !default_graph(m->cfg);
    int rc;
    // per-graph atom counts: the exact size of the largest graph, and the chunk layout of a general graph
    std::vector<int32_t> hp, hl;
    if ((rc = fetch_ptrs(d_protein_ptr, d_ligand_ptr, B, hp, hl, s)) != TD_OK) { delete S; return rc; }
    int gmax = 0;
    for (int64_t g = 0; g < B; ++g) gmax = std::max(gmax, (hp[g + 1] - hp[g]) + (hl[g + 1] - hl[g]));
    S->graph_nodes_max = gmax;
    const bool knn_like = m->cfg.cutoff_mode == TD_CUTOFF_KNN || m->cfg.cutoff_mode == TD_CUTOFF_HYBRID;
    S->caching = knn_like && (!S->chunked || gmax <= TD_STEP_LISTS_MAX_NODES) && num_blocks(m->cfg) == 1 && m->cfg.ew_net_type == 0 &&
                 !m->cfg.x2h_out_fc && !m->cfg.sync_twoup && stage_rows(m->cfg) == 1;      // (the static-protein tables hold the global gate's rows and plain x2h outputs;
                                                                 // a cached layer 0 skips the projections sync_twoup takes at its top)
    if (S->chunked && (rc = plan_create(m->cfg, hp.data(), hl.data(), B, s, &S->plan)) != TD_OK) { delete S; return rc; }
    const int64_t NC = S->chunked ? S->plan.NC : N;          // 32-slot rows of the neighbour table
    const int KS = S->chunked ? 64 : TD_K;                   // static keys kept per protein row

    // ---- one device block: [workspace | session-static buffers]
    const size_t ws_bytes = carve(nullptr, N, B, N_l).bytes;
    size_t off = ws_bytes;
    auto reserve = [&](size_t n) { size_t o = off; off += align_up(n ? n : 4); return o; };
    const size_t n = (size_t)N, nc = (size_t)NC;
    const bool C = S->caching;
    const size_t o_prot = reserve((size_t)N_p * 4), o_pptr = reserve((size_t)(B + 1) * 4), o_lptr = reserve((size_t)(B + 1) * 4),
                 o_h0 = reserve(n * TD_H * 4), o_tmp_lpos = reserve((size_t)N_l * 12), o_tmp_lv = reserve((size_t)N_l * 8),
                 o_snbr = reserve(C ? nc * TD_K * 4 : 0), o_skeys = reserve(C ? n * KS * 8 : 0), o_ews = reserve(C ? nc * TD_K * 4 : 0),
                 o_h1s = reserve(C ? n * TD_H * 4 : 0), o_h2s = reserve(C ? n * TD_H * 4 : 0), o_f2 = reserve(C ? n : 0),
                 o_frows = reserve(C ? n * 4 : 0), o_frest = reserve(C ? n * 4 : 0), o_fcnt = reserve(256),
                 o_P0 = reserve(C ? n * 4 * TD_H * 4 : 0), o_q0 = reserve(C ? n * TD_H * 4 : 0), o_clean = reserve(C ? n : 0),
                 o_dirty = reserve(C ? n * 4 : 0), o_dcnt = reserve(256), o_hop = reserve(C ? n * 4 * TD_HOP_LEVELS : 0),
                 o_hcnt = reserve(256), o_dchunks = reserve(C && S->chunked ? nc * 4 : 0),
                 o_ppos = reserve((size_t)N_l * 3 * 4), o_pv = reserve((size_t)N_l * TD_MAXC * 4);
    hipError_t e = td_malloc_async(reinterpret_cast<void **>(&S->block), off, s);
    if (e != hipSuccess) {
        td_set_error("td_session_create: hipMallocAsync(%zu) failed: %s", off, hipGetErrorString(e));
        S->block = nullptr;
        session_free(S, s);
        return TD_ENOMEM;
    }
    char *b = S->block;
    S->w = carve(b, N, B, N_l);
    S->prot_node = reinterpret_cast<int32_t *>(b + o_prot);
    S->pptr = reinterpret_cast<int32_t *>(b + o_pptr);
    S->lptr = reinterpret_cast<int32_t *>(b + o_lptr);
    S->snbr = reinterpret_cast<int32_t *>(b + o_snbr);
    S->skeys = reinterpret_cast<unsigned long long *>(b + o_skeys);
    S->ews = reinterpret_cast<float *>(b + o_ews);
    S->h0 = reinterpret_cast<float *>(b + o_h0);
    S->h1s = reinterpret_cast<float *>(b + o_h1s);
    S->h2s = reinterpret_cast<float *>(b + o_h2s);
    S->flags2 = reinterpret_cast<uint8_t *>(b + o_f2);
    S->fwd_rows = reinterpret_cast<int32_t *>(b + o_frows);
    S->fwd_rest = reinterpret_cast<int32_t *>(b + o_frest);
    S->fwd_counts = reinterpret_cast<int32_t *>(b + o_fcnt);
    S->use_fwd = C && m->opt.session_forward_reach && m->cfg.num_layers >= 2;
    S->P0 = reinterpret_cast<float *>(b + o_P0);
    S->q0 = reinterpret_cast<float *>(b + o_q0);
    S->clean = reinterpret_cast<uint8_t *>(b + o_clean);
    S->dirty_rows = reinterpret_cast<int32_t *>(b + o_dirty);
    S->dirty_count = reinterpret_cast<int32_t *>(b + o_dcnt);       // [0] rows, [1] chunks (general graphs)
    S->hop_rows = reinterpret_cast<int32_t *>(b + o_hop);
    S->hop_count = reinterpret_cast<int32_t *>(b + o_hcnt);
    S->dirty_chunks = reinterpret_cast<int32_t *>(b + o_dchunks);
    S->pred_pos = reinterpret_cast<float *>(b + o_ppos);
    S->pred_v = reinterpret_cast<float *>(b + o_pv);
    {
        // receptive-field levels tracked per step (each prunes one more layer from the end)
        int lv = m->opt.session_hop_levels;
        if (lv < 1) lv = 1;
        if (lv > TD_HOP_LEVELS) lv = TD_HOP_LEVELS;
        if (lv > m->cfg.num_layers) lv = m->cfg.num_layers;
        S->hop_levels = lv;
    }
    float *tmp_lpos = reinterpret_cast<float *>(b + o_tmp_lpos);
    int64_t *tmp_lv = reinterpret_cast<int64_t *>(b + o_tmp_lv);
    Workspace &w = S->w;
    auto fail = [&](int r) { session_free(S, s); return r; };
#define TD_TRY(expr) do { if ((rc = (expr)) != TD_OK) return fail(rc); } while (0)
#define TD_TRY_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { td_set_error("%s failed: %s", #expr, hipGetErrorString(_e)); return fail(TD_EHIP); } } while (0)
    TD_TRY_HIP(hipMemcpyAsync(S->pptr, d_protein_ptr, (size_t)(B + 1) * 4, hipMemcpyDeviceToDevice, s));
    TD_TRY_HIP(hipMemcpyAsync(S->lptr, d_ligand_ptr, (size_t)(B + 1) * 4, hipMemcpyDeviceToDevice, s));
    TD_TRY_HIP(hipMemsetAsync(tmp_lpos, 0, (size_t)N_l * 12, s));
    TD_TRY_HIP(hipMemsetAsync(tmp_lv, 0, (size_t)N_l * 8, s));
    // embeddings + packed order (ligand rows are placeholders until the first step), protein row list
    int32_t *prot_out = S->chunked ? S->plan.prot_node : S->prot_node;
    TD_TRY(td_launch_compose(m, d_protein_pos, d_protein_v, S->pptr, N_p, tmp_lpos, tmp_lv, S->lptr, N_l, B, S->h0, w.x4a,
                             w.node_ptr, w.gid, w.lig_node, prot_out, s));
    if (S->chunked) {
        S->prot_node = S->plan.prot_node;
        TD_TRY(plan_layout(S->plan, w.node_ptr, w.gid, s));
    }
    TD_TRY_HIP(hipMemcpyAsync(w.x4b, w.x4a, n * sizeof(float4), hipMemcpyDeviceToDevice, s));
    if (!S->caching) {
        *out = S;
        return TD_OK;
    }
    // ---- protein-only graph, its gate rows, layer-0 projections / queries, layer-0 x2h output
    GraphTab gt = session_tab(S);                  // the step's table (alpha is scratch here)
    GraphTab st = gt;                              // the static table
    st.nbr = S->snbr; st.ew = S->ews;
    TD_TRY_HIP(hipMemsetAsync(S->snbr, 0xff, nc * TD_K * 4, s));
    if (S->chunked) {
        TD_TRY(td_launch_knn_general_static(w.x4a, w.node_ptr, S->pptr, w.gid, S->prot_node, N_p, m->cfg.knn, gmax, S->plan.cptr,
                                            S->snbr, S->skeys, s));
        TD_TRY(td_launch_gate(m->gate, w.x4a, S->snbr, NC, nullptr, nullptr, S->ews, s, S->plan.chunk_node));
        // the step's table: ligand rows start as all pads; hybrid: their ligand half never changes
        TD_TRY_HIP(hipMemsetAsync(S->plan.cnbr, 0xff, nc * TD_K * 4, s));
        TD_TRY_HIP(hipMemsetAsync(S->plan.ew, 0, nc * TD_K * 4, s));
        if (m->cfg.cutoff_mode == TD_CUTOFF_HYBRID)
            TD_TRY(td_launch_hybrid_ligand_half(w.node_ptr, S->pptr, w.gid, w.lig_node, N_l, S->plan.cptr, S->plan.cnbr, s));
    } else {
        TD_TRY(td_launch_knn_static(w.x4a, w.node_ptr, w.gid, S->prot_node, N_p, max_graph_nodes, S->snbr, S->skeys, s, m->cfg.knn));
        TD_TRY(td_launch_gate(m->gate, w.x4a, S->snbr, N_p, S->prot_node, nullptr, S->ews, s));
    }
    const TdLayer &L0 = m->layers[0];
    TD_TRY(td_launch_node_proj(L0.nodeX2h, S->h0, N, nullptr, 0x1f, S->P0, S->q0, s));
    TD_TRY(key_pass(L0.hk, L0, w.x4a, st, st.ew, st.nbr, S->P0, S->q0, S->prot_node, nullptr, N_p, gt.alpha, s));
    TD_TRY_HIP(hipMemcpyAsync(S->h1s, S->h0, n * TD_H * 4, hipMemcpyDeviceToDevice, s));
    TD_TRY(value_pass(L0.hv, L0, w.x4a, st, st.nbr, S->P0, S->prot_node, nullptr, N_p, S->h1s, gt.alpha, nullptr, 0, s));
    if (S->use_fwd) {      // layer-1 x2h output of the protein-only graph (valid wherever the ligand is two hops away)
        const TdLayer &L1 = m->layers[1];
        TD_TRY(td_launch_node_proj(L1.nodeX2h, S->h1s, N, nullptr, 0x1f, w.P, w.q, s));
        TD_TRY(key_pass(L1.hk, L1, w.x4a, st, st.ew, st.nbr, w.P, w.q, S->prot_node, nullptr, N_p, gt.alpha, s));
        TD_TRY_HIP(hipMemcpyAsync(S->h2s, S->h1s, n * TD_H * 4, hipMemcpyDeviceToDevice, s));
        TD_TRY(value_pass(L1.hv, L1, w.x4a, st, st.nbr, w.P, S->prot_node, nullptr, N_p, S->h2s, gt.alpha, nullptr, 0, s));
    }
#undef TD_TRY
#undef TD_TRY_HIP
    *out = S;
    return TD_OK;
}

extern "C" void td_session_destroy(td_session *S) {
    if (!S) return;
    session_free(S, S->last_stream);
}

extern "C" int td_session_forward(td_session *S, const float *d_ligand_pos, const int64_t *d_ligand_v,
                                  float *d_pred_ligand_pos, float *d_pred_ligand_v, float *d_final_ligand_h,
                                  const float *d_ligand_graph_bias, void *stream) {
    if (!S || !d_ligand_pos || !d_ligand_v || !d_pred_ligand_pos || !d_pred_ligand_v) {
        td_set_error("td_session_forward: null pointer");
        return TD_EINVAL;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    S->last_stream = s;
    const td_model *m = S->m;
    Workspace &w = S->w;
    const int64_t N = S->N, Nl = S->Nl, Np = S->Np;
    const GraphTab gt = session_tab(S);
    int rc;
    if (!S->caching) {
        {
            ProfScope ps(PC_COMPOSE, s);
            TD_CHECK_HIP(hipMemcpyAsync(w.h, S->h0, (size_t)N * TD_H * sizeof(float), hipMemcpyDeviceToDevice, s));
            if ((rc = td_launch_ligand_update(m, d_ligand_pos, d_ligand_v, w.lig_node, Nl, w.h, w.x4a, s, nullptr, d_ligand_graph_bias, w.gid)) != TD_OK) return rc;
        }
        float4 *xg = nullptr;
        Workspace wb = w;             // (a block that ends in x4b hands its coordinates to the next one by swapping the two buffers)
        rc = TD_OK;
        for (int blk = 0; rc == TD_OK && blk < num_blocks(m->cfg); ++blk) {
            if (S->chunked) rc = build_general_graph(m, S->plan, wb, N, Nl, S->max_graph_nodes, s);
            else rc = build_default_graph(m, wb, N, S->max_graph_nodes, s);          // 32-slot rows
            if (rc == TD_OK) rc = run_backbone(m, wb, gt, wb.h, N, Nl, 0, &xg, s, true);
            if (rc == TD_OK && xg == wb.x4b) std::swap(wb.x4a, wb.x4b);
        }
        if (rc != TD_OK) return rc;
        ProfScope ps(PC_HEAD, s);
        return td_launch_head(m->head, w.h, xg, w.lig_node, Nl, m->cfg.ligand_num_classes, d_pred_ligand_pos, d_pred_ligand_v,
                              d_final_ligand_h, s);
    }
    {
        // first kernel of the step: also zeroes the row-list counters and the ligand rows' forward-reach flags
        ProfScope ps(PC_COMPOSE, s);
        TdStepReset rs;
        rs.c0 = S->dirty_count; rs.n0 = 2;
        rs.c1 = S->fwd_counts; rs.n1 = 2;
        rs.c2 = S->hop_count; rs.n2 = TD_HOP_LEVELS;
        rs.flags2 = S->use_fwd ? S->flags2 : nullptr;
        if ((rc = td_launch_ligand_update(m, d_ligand_pos, d_ligand_v, w.lig_node, Nl, w.h, w.x4a, s, &rs, d_ligand_graph_bias, w.gid)) != TD_OK) return rc;
    }
    bool lists_done = false;
    {
        ProfScope ps(PC_KNN, s);
        const TdStepLists lists{S->dirty_rows, S->dirty_count, S->use_fwd ? S->fwd_rows : nullptr, S->fwd_rest, S->fwd_counts,
                                S->hop_rows, S->hop_count, S->hop_levels, S->chunked ? S->dirty_chunks : nullptr,
                                S->chunked ? S->dirty_count + 1 : nullptr};
        if (S->chunked) {
            const GraphPlan &p = S->plan;
            if ((rc = td_launch_knn_merge_general(w.x4a, w.node_ptr, S->pptr, w.gid, S->prot_node, Np, S->skeys, S->snbr, S->h0,
                                                  S->h1s, S->ews, p.cptr, p.cpn_p, p.cnbr, w.h, p.ew, S->clean,
                                                  S->use_fwd ? S->flags2 : nullptr, s, m->cfg.knn)) != TD_OK) return rc;
            if ((rc = td_launch_ligand_rows_general(p.mode, w.x4a, w.node_ptr, S->pptr, w.gid, w.lig_node, Nl, p.k, S->graph_nodes_max,
                                                    p.cptr, p.cnbr, s)) != TD_OK) return rc;
            // every row list of the step in one launch (the session is CACHING only when every graph fits its LDS flags)
            if ((rc = td_launch_step_lists(S->clean, w.x4a, p.cnbr, w.node_ptr, N, S->B, S->graph_nodes_max, lists, s, p.cptr,
                                           p.chunk_node)) != TD_OK) {
                if (rc == TD_EINVAL) td_set_error("td_session_forward: a graph of %d nodes does not fit the row-list kernel", S->graph_nodes_max);
                return rc;
            }
            lists_done = true;
        } else {
            // the protein rows' merge and the ligand rows' full search: independent, one launch
            if ((rc = td_launch_knn_merge(w.x4a, w.node_ptr, S->pptr, w.gid, S->prot_node, Np, S->skeys, S->snbr, S->h0,
                                          S->h1s, S->ews, w.nbr, w.h, w.ew, S->clean, S->use_fwd ? S->flags2 : nullptr, s, m->cfg.knn,
                                          w.lig_node, Nl, S->max_graph_nodes)) != TD_OK) return rc;
            // every row list of the step (dirty rows, forward reach, receptive-field levels) in one launch, one workgroup per graph;
            // graphs too large for its LDS flags take the separate kernels
            rc = (m->opt.session_step_lists && S->graph_nodes_max > 0) ? td_launch_step_lists(S->clean, w.x4a, w.nbr, w.node_ptr, N, S->B, S->graph_nodes_max, lists, s)
                                        : TD_EINVAL;
            lists_done = rc == TD_OK;
            if (rc != TD_OK && rc != TD_EINVAL) return rc;
            if (!lists_done) {
                if ((rc = td_launch_compact_dirty(S->clean, w.x4a, N, S->dirty_rows, S->dirty_count, s)) != TD_OK) return rc;
                // S->clean gets its second life as the receptive-field flags below: the forward-reach compaction (its last reader)
                // clears it; without the forward reach a memset does
                if (S->use_fwd) {
                    if ((rc = td_launch_forward_reach(S->clean, w.x4a, w.nbr, N, S->flags2, S->fwd_rows, S->fwd_rest, S->fwd_counts,
                                                      S->clean, s)) != TD_OK) return rc;
                } else {
                    TD_CHECK_HIP(hipMemsetAsync(S->clean, 0, (size_t)N, s));
                }
            }
        }
    }
    {
        ProfScope ps(PC_GATE, s);
        if (S->chunked)
            rc = td_launch_gate(m->gate, w.x4a, gt.nbr, S->plan.NC, S->dirty_chunks, S->dirty_count + 1, gt.ew, s, S->plan.chunk_node);
        else
            rc = td_launch_gate(m->gate, w.x4a, gt.nbr, N, S->dirty_rows, S->dirty_count, gt.ew, s);
        if (rc != TD_OK) return rc;
    }
    const TdLayer &L0 = m->layers[0];
    {   // layer 0, x2h: only ligand rows need new projections, only dirty rows need the attention passes
        { ProfScope ps(PC_NODE, s); if ((rc = td_launch_node_proj(L0.nodeX2h, w.h, Nl, w.lig_node, 0x1f, S->P0, S->q0, s)) != TD_OK) return rc; }
        { ProfScope ps(PC_X2H_K, s); if ((rc = key_pass(L0.hk, L0, w.x4a, gt, gt.ew, gt.nbr, S->P0, S->q0, S->dirty_rows, S->dirty_count, N, gt.alpha, s, w.lig_node, Nl)) != TD_OK) return rc; }
        { ProfScope ps(PC_X2H_V, s); if ((rc = value_pass(L0.hv, L0, w.x4a, gt, gt.nbr, S->P0, S->dirty_rows, S->dirty_count, N, w.h, gt.alpha, w.lig_node, Nl, s)) != TD_OK) return rc; }
    }
    // rows the last layer still has to update (S->clean is free again after the dirty-row compaction: reuse as flags)
    if (!lists_done && (rc = td_launch_hop_levels(w.lig_node, Nl, w.nbr, N, S->clean, S->hop_rows, S->hop_count, S->hop_levels, s, true)) != TD_OK) return rc;
    float4 *xf = nullptr;
    const FwdReach fwd{S->fwd_rows, S->fwd_rest, S->fwd_counts, S->h2s};
    if ((rc = run_backbone(m, w, gt, w.h, N, Nl, 0, &xf, s, false, true, S->hop_rows, S->hop_count, S->hop_levels,
                           S->use_fwd ? &fwd : nullptr)) != TD_OK) return rc;
    ProfScope ps(PC_HEAD, s);
    return td_launch_head(m->head, w.h, xf, w.lig_node, Nl, m->cfg.ligand_num_classes, d_pred_ligand_pos,
                          d_pred_ligand_v, d_final_ligand_h, s);
}

namespace {
// the launches of one step, in order (eagerly or into a capturing stream)
int session_step_issue(td_session *S, const td_step_io &io, hipStream_t s) {
    const td_model *m = S->m;
    int rc = td_session_forward(S, io.d_ligand_pos, io.d_ligand_v, S->pred_pos, S->pred_v, nullptr, io.d_ligand_graph_bias, s);
    if (rc != TD_OK) return rc;
    ProfScope ps(PC_POST, s);
    return td_launch_posterior_step(m->sched, m->cfg.num_timesteps, io.d_step, io.d_t_all, io.num_steps, S->lptr, S->Nl, S->B,
                                    m->cfg.ligand_num_classes, io.d_ligand_pos, io.d_ligand_v, S->pred_pos, S->pred_v, io.d_noise,
                                    io.d_uniform, io.d_pos_traj, io.d_v_traj, io.d_v0_traj, io.d_vt_traj, io.pos_only, s,
                                    m->cfg.model_mean_type);
}
}  // namespace

extern "C" int td_session_step(td_session *S, const td_step_io *io, int32_t use_graph, void *stream) {
    if (!S || !io || !io->d_step || !io->d_t_all || io->num_steps < 1 || !io->d_ligand_pos || !io->d_ligand_v || !io->d_noise ||
        !io->d_uniform || !io->d_pos_traj || !io->d_v_traj) {
        td_set_error("td_session_step: bad argument");
        return TD_EINVAL;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    S->last_step_graph = false;
    // measurement hooks put events / trace pointers into the launch sequence: those steps are issued launch by launch
    const bool graph_ok = use_graph && !S->graph_failed && g_prof.mask == 0 && !td_wg_trace_armed();
    if (graph_ok && S->graph_exec && (memcmp(&S->graph_io, io, sizeof(td_step_io)) != 0 || S->graph_epoch != S->m->option_epoch)) session_drop_graph(S);
    if (graph_ok && !S->graph_exec && S->eager_steps > 0) {
        // capture the launch sequence of a step (nothing executes while the stream captures), instantiate it once
        hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
        if (e == hipSuccess) {
            const int rc = session_step_issue(S, *io, s);
            hipGraph_t g = nullptr;
            e = hipStreamEndCapture(s, &g);
            if (rc == TD_OK && e == hipSuccess && g) {
                e = hipGraphInstantiate(&S->graph_exec, g, nullptr, nullptr, 0);
                if (e == hipSuccess) { S->graph = g; S->graph_io = *io; S->graph_epoch = S->m->option_epoch; }
                else { (void)hipGraphDestroy(g); S->graph_exec = nullptr; }
            } else if (g) {
                (void)hipGraphDestroy(g);
            }
        }
        if (!S->graph_exec) {       // not fatal: this session keeps issuing its steps launch by launch
            (void)hipGetLastError();
            S->graph_failed = true;
        }
    }
    if (graph_ok && S->graph_exec) {
        S->last_stream = s;
        TD_CHECK_HIP(hipGraphLaunch(S->graph_exec, s));
        S->last_step_graph = true;
        return TD_OK;
    }
    const int rc = session_step_issue(S, *io, s);
    if (rc == TD_OK) ++S->eager_steps;
    return rc;
}

extern "C" int td_session_step_graph(const td_session *S) { return S && S->last_step_graph ? 1 : 0; }

extern "C" int td_session_row_counts(td_session *S, int32_t *host_counts, int32_t n_counts, void *stream) {
    if (!S || !host_counts || n_counts < 2) { td_set_error("td_session_row_counts: bad argument"); return TD_EINVAL; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    host_counts[0] = (int32_t)S->N;
    if (!S->caching) {            // every layer runs on every row
        host_counts[1] = (int32_t)S->N;
        for (int k = 2; k < n_counts; ++k) host_counts[k] = -1;
        return TD_OK;
    }
    TD_CHECK_HIP(hipMemcpyAsync(host_counts + 1, S->dirty_count, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    for (int k = 2; k < n_counts; ++k) host_counts[k] = -1;
    const int room = n_counts - 2 < TD_HOP_LEVELS ? n_counts - 2 : TD_HOP_LEVELS;
    const int lv = S->hop_levels < room ? S->hop_levels : room;
    if (lv > 0) TD_CHECK_HIP(hipMemcpyAsync(host_counts + 2, S->hop_count, sizeof(int32_t) * (size_t)lv, hipMemcpyDeviceToHost, s));
    if (n_counts > 2 + TD_HOP_LEVELS && S->use_fwd)
        TD_CHECK_HIP(hipMemcpyAsync(host_counts + 2 + TD_HOP_LEVELS, S->fwd_counts, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    TD_CHECK_HIP(hipStreamSynchronize(s));
    return TD_OK;
}
