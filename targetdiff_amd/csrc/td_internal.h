// Internal declarations shared by the translation units of libtargetdiff_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>

#include "../../include/targetdiff_hip.h"

// ---- compile-time shape of the live configuration (configs/training.yml:9-42) ----------------------
constexpr int TD_H = 128;        // hidden_dim
constexpr int TD_HEADS = 16;     // n_heads
constexpr int TD_DH = 8;         // head dim
constexpr int TD_K = 32;         // knn fan-in: one node's in-edges = one 32-row MFMA tile
constexpr int TD_NG = 20;        // num_r_gaussian
constexpr int TD_SLOTK = 24;     // K columns per source-class slot of the first-layer edge GEMM: 20 g + 1 + 3 pad
constexpr int TD_SLOT_STEPS = TD_SLOTK / 2;   // 12 MFMA k-steps per slot
constexpr int TD_KSTEPS = TD_H / 2;           // 64 k-steps (32x32x2) for a 128-deep contraction
constexpr int TD_MAXC = 16;      // ligand classes padded
constexpr int TD_SMALL_BATCH_ROWS = 16384;   // below this many rows a launch is shaped for latency (more, smaller workgroups)

void td_set_error(const char *fmt, ...);
#define TD_CHECK_HIP(expr)                                                                       \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            td_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return TD_EHIP;                                                                      \
        }                                                                                        \
    } while (0)

// Dynamic-LDS opt-in of a kernel, once per device (not per process: a second GPU used from the same process needs its
// own hipFuncSetAttribute) and safe against concurrent first launches.
struct TdLdsOnce {
    std::atomic<unsigned long long> mask{0};
};
inline int td_set_lds(TdLdsOnce &once, const void *fn, size_t bytes) {
    int dev = 0;
    TD_CHECK_HIP(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (once.mask.load(std::memory_order_acquire) & bit) return TD_OK;
    TD_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    once.mask.fetch_or(bit, std::memory_order_release);
    return TD_OK;
}

// ---- packed weights (device pointers) ---------------------------------------------------------------
// One edge MLP (hk/hv/xk/xv: Linear(340,128) -> LN -> ReLU -> Linear(128,out)), re-packed:
//  * the 340-wide first Linear is split into node-side projections (proj_*), a per-(dst class, src class)
//    radial/type table R and a bias;  * the second Linear is stored as per-wave MFMA B fragments.
struct TdEdgeMlp {
    const float *R;        // [2 dst class][2 slot][12 kstep][64 lane][4 ntile]  first-layer radial+type B fragments
    // (every table below comes from the MLP with its LayerNorm folded into the two Linears: FoldedMlp, pack.cpp)
    const float *gamma;    // [128] |LayerNorm weight| x the folded scale M (0 for a dead unit): already inside the second Linear's columns; not read by the kernels
    const float *beta;     // [128] LayerNorm bias / (|LayerNorm weight| M): z'' = clamp_[0,1](centred pre-activation / (sigma M) + beta)
    const float *W2;       // out=128: [64 kstep][64 lane][4 ntile];  out=16 (xv): [64 kstep][64 lane] (cols >= 16 zero)
    const float *b2;       // [128] or [16]
    const float *R16;      // [2 dst class][2 slot][6 kstep][64 lane][8 hidden block]  radial/type table for 16x16x4 tiles
    const float *Walt16;   // key MLPs: Wq16[hb][r][jq][lane][4] = W2[8 lo + 4jq + jj][16hb + 4g + r]
    const float *Walt;     // key MLPs: Wq[t][r][jq][hi][c<16][4] = W2[8c+4jq+jj][32t+erow(r,hi)];  hv: Wt[d][k/4][head][4] = W2v[8 head + d][k]
    const float *R16q;     // the radial/type table as exact bf16 piece triples, K-packed for four v_mfma_f32_16x16x32_bf16 per tile
                           // (pack_pk4_table, pack.cpp): [2 dst class][2 slot] x {QA, QB, H7, QC}[8 hidden block][64 lanes]
    const float *R16h;     // attention MLPs (hk, hv, xk, xv): the same table as f16 piece pairs for two v_mfma_f32_16x16x32_f16 per tile (pack_h2_table); the MLP's whole
                           // first layer, node projections included, is in units of 2^FoldedMlp::first_scale_exp
    float ln_c1, ln_c2;    // folded LayerNorm (FoldedMlp, pack.cpp): 1 / (sigma M) = rsqrt(sum_n c_n^2 * ln_c1 + ln_c2); z'' = clamp_[0,1](c_n / (sigma M) + beta_n)
    float w2_bound;        // key MLPs: 8 max |W2'| (folded second Linear): |U_i[n][head]| = |sum_d W2'[8 head + d][n] q_i[8 head + d]| <= w2_bound max |q_i|
                           // (the f16 logits product scales the query by a power of two from this bound, edge16.hip)
    bool l2_f16;           // x2h passes: logits (rows of one chunk) / alpha^T z (every graph) on v_mfma_f32_16x16x32_f16 with f16 piece pairs (model option "edge_second_layer_f16")
    bool l1_f16;           // attention kernels (x2h passes, h2x stage): the radial / type first layer on f16 piece pairs (R16h; model option "edge_first_layer_f16"); false: exact bf16 piece triples
    bool z_plain;          // f16 second layer: the folded scale M is at most TD_Z_PLAIN_MAX_M, take the f16 pieces of z'' itself (edge16.hip, td_ln_relu16_pairs_*)
    bool use_split;        // run the first layer on the piece triples where a kernel has that variant (model option "edge_key_split")
    int deal_rows;         // x2h passes: rows dealt round-robin inside an XCD's range (model option "edge_row_dealing": 0 contiguous
                           // shares, 1 dealt, 2 dealt + the workgroup's rows handed to its waves through an LDS counter)
};

// Node-side weights of one stage (x2h or h2x): 4 projections (k_i,k_j,v_i,v_j) + the query MLP.
struct TdNodeStage {
    const float *projB;    // [5 mat][64 kstep][64 lane][4 ntile]   mats: k_i, k_j, v_i, v_j, q.net.0
    const float *projBias; // [5][128]  (k_i: b0 of k MLP, k_j: 0, v_i: b0 of v MLP, v_j: 0, q: b0)
    const float *qGamma, *qBeta;   // [128]
    const float *q3B;      // [64 kstep][64 lane][4 ntile]  q.net.3
    const float *q3Bias;   // [128]
    const float *projB3;   // optional: the same 5 matrices as bf16 piece triples [mat][8 kstep][3 piece][64 lane][4 ntile] x 8 bf16
    const float *q3B3;     // optional: q.net.3 likewise (nullptr: the fp32 path is the only one)
    bool use_split;        // run the GEMMs on the exact 3-way bf16 operand split (model option "node_proj_split")
    bool bpipe;            // split kernel: B fragments of the next k-step read ahead of the current k-step's MFMAs ("node_proj_bpipe")
    bool async_copy;         // split kernel: B chunks by inline-asm global_load_lds + explicit wait (model option "node_proj_async", default)
};

// node_output MLP of an x2h stage with out_fc (models/uni_transformer.py:39-40): Linear(256, 128) -> LayerNorm -> ReLU -> Linear(128, 128)
struct TdNodeOut {
    const float *B;        // 3 x B fragments (pack_B128): net.0[:, 0:128] (attention output), net.0[:, 128:256] (h), net.3
    const float *b1, *gamma, *beta, *b2;
};

struct TdLayer {
    TdNodeStage nodeX2h, nodeH2x;
    TdEdgeMlp hk, hv, xk, xv;
    const float *ew_x2h, *ew_h2x;   // ew_net_type 'r' / 'm' / none: [4 types][20] + bias of the stage's gate (nullptr: the global gate; 'm' and none:
                                    // zero weights and a bias of 40, i.e. e_w = 1 -- 'm' gates in the value pass instead)
    const float *gate_m;            // ew_net_type 'm': [128] folded W2v^T w_m, then w_m . b2v + b_m (nullptr: off)
    TdNodeOut nodeOut;              // x2h_out_fc (B == nullptr: off)
    const float *offsets;  // [20] Gaussian centres of this layer (models/common.py:15)
    float coeff;           // -0.5 / (offset[1]-offset[0])^2
};

struct TdGate {            // edge_pred_layer MLP(20 -> 128 -> 1) (models/uni_transformer.py:236-237,312-316)
    const float *R;        // [12 kstep][64 lane][4 ntile]
    const float *b0, *gamma, *beta, *w3;   // [128] each, LayerNorm folded like the edge MLPs' (FoldedMlp): beta = bias / (|weight| M), w3 carries |weight| M
    float b3;
    const float *offsets;  // [20]
    float coeff;
    const float *R16q;     // the 20 x 128 first layer as exact bf16 piece triples, K-packed (pack_pk4_table, pack.cpp)
    float ln_c1, ln_c2;    // as in TdEdgeMlp
    bool use_split;        // model option "edge_key_split"
};

struct TdEmbed {
    const float *WpT;      // [protein_feat_dim][128]  (column 127 zero)
    const float *bp;       // [128] (entry 127 = 0: node indicator of protein atoms)
    const float *WlT;      // [classes][128]
    const float *bl;       // [128] (entry 127 = 1: node indicator of ligand atoms)
};

struct TdHead {            // v_inference (models/molopt_score_model.py:307-311)
    const float *W0T;      // [128 in][128 out]
    const float *b0;       // [128]
    const float *W2T;      // [128 in][16 out] (classes padded)
    const float *b2;       // [16]
};

// One EnBaseLayer of the standalone EGNN refine net (models/egnn.py:10-35), packed by td_egnn_create
struct TdEgnnLayer {
    TdNodeStage proj;      // node_proj_kernel, mask 0x03: mat 0 = edge_mlp.net.0[:, 0:128] (+ bias), mat 1 = [:, 128:256]
    const float *W2f;      // edge_mlp.net.2 as 16x16x4 A fragments [ot][hb][lane] x 4 r
    const float *Wxf;      // x_mlp.0, same layout
    const float *vec;      // [w_d 128 | W_t 4 x 128 | b2 128 | w_inf 128, b_inf, pad 3 | b_x 128 | w_x2 128]
    const float *nodeB;    // 3 x B fragments (pack_B128): node_mlp.net.0[:, 0:128] (mi), [:, 128:256] (h), node_mlp.net.2
    const float *nb1, *nb2;
};

struct TdSchedules {       // [T] each
    const float *c0, *ct, *logvar, *log_a, *log_1ma, *log_ca, *log_1mca;
    const float *abar;     // alphas_cumprod of the position schedule; nullptr when the model was created without it
    const float *rc, *rm1; // sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod (model_mean_type 'noise'); nullptr without them
};

// Per-model switches (td_model_set_option; defaults = the shipped configuration).  They live in the model, i.e. per device
// and per handle -- nothing is read from the environment.
struct TdOptions {
    int h2x_fused = 1;             // one launch for the h2x stage's key + value halves (0: two launches, alpha through memory)
    int node_proj_split = 1;       // node-side GEMMs on exact bf16 x 3 operand pieces with fp32 accumulation (0: fp32 MFMA)
    int edge_key_split = 1;        // attention passes: radial/type first layer on exact bf16 x 3 pieces (0: fp32 MFMA)
    int edge_first_layer_f16 = 1;  // x2h passes and h2x stage: the 21-wide radial / type first layer on f16 piece pairs (two products per tile; 0: the exact bf16 piece triples, four)
    int edge_second_layer_f16 = 1; // x2h passes: logits / alpha^T z on f16 piece pairs (0: fp32 MFMA products; the chunk-walking key pass takes the f16 logits only beside the f16 first layer)
    int session_hop_levels = 4;    // receptive-field levels a sampling session tracks (1 .. 4)
    int session_forward_reach = 1; // layer 1 of a session runs on the ligand's one-hop forward reach only
    int edge_row_dealing = 2;      // x2h key / value passes: units of rows dealt round-robin to an XCD's workgroups (0: one
                                   // contiguous share per workgroup; 1: a fixed row sequence per wave; 2: the workgroup's rows
                                   // handed to its waves one at a time through an LDS counter; the value pass's protein rows on a
                                   // general graph with one chunk per protein row always take 2: the 12-wave kernel has no other form)
    int node_proj_bpipe = 0;       // split node GEMMs: register double-buffering of the B fragments (LDS reads ahead of the MFMAs; measured
                                   // slower at C2: 0.848 vs 0.817 ms per step, profiles/r03b_*; kept as a switch)
    int node_proj_async = 1;    // split node GEMMs: B chunks by inline-asm global_load_lds + an explicit wait per round (0: the builtin, which the compiler serialises)
    int session_share_pockets = 1; // a session's static tables (protein-only k-NN keys / lists, cached gate rows, embeddings, layer-0 / layer-1 outputs of the
                                   // protein-only graph) once per distinct pocket of the batch instead of once per graph (default graph; 0: per graph)
    int session_step_lists = 1;    // a step's row lists from one launch (a workgroup per graph; 0: the separate kernels, which
                                   // graphs too large for its LDS flags use anyway)
};

struct td_model {
    td_config cfg;
    TdOptions opt;
    float *blob;           // one device allocation holding every packed tensor
    size_t blob_floats;
    TdEmbed emb;
    TdGate gate;
    TdLayer *layers;       // host array [num_layers]
    TdHead head;
    TdSchedules sched;
    unsigned option_epoch = 0;   // bumped by td_model_set_option: sessions drop a step graph captured under older options
};

// ---- kernel launchers (each defined next to its kernels) --------------------------------------------
// graph.hip
int td_launch_graph_ptr(const int64_t *batch, int64_t N, int64_t B, int32_t *ptr, hipStream_t s);
int td_launch_node_gid(const int32_t *node_ptr, int64_t N, int64_t B, int32_t *gid, hipStream_t s);
int td_launch_pack_x(const float *x3, const uint8_t *mask, int64_t N, float4 *x4, hipStream_t s);
int td_launch_unpack_x(const float4 *x4, int64_t N, float *x3, hipStream_t s);
// k <= TD_K neighbours per 32-slot row (slots >= k are -1): every fan-in up to 32 runs on the 32-slot fast path
int td_launch_knn(const float4 *x4, const int32_t *node_ptr, const int32_t *gid, int64_t N, int max_graph_nodes,
                  int32_t *nbr, hipStream_t s, int k = TD_K);
int td_launch_compose(const td_model *m, const float *ppos, const float *pv, const int32_t *pptr, int64_t Np,
                      const float *lpos, const int64_t *lv, const int32_t *lptr, int64_t Nl, int64_t B,
                      float *h, float4 *x4, int32_t *node_ptr, int32_t *gid, int32_t *lig_node, int32_t *prot_node,
                      hipStream_t s, const float *gbias = nullptr);
int td_launch_knn_rows(const float4 *x4, const int32_t *node_ptr, const int32_t *gid, const int32_t *rows, int64_t count,
                       int max_graph_nodes, int32_t *nbr, hipStream_t s, int k = TD_K);
int td_launch_knn_static(const float4 *x4, const int32_t *node_ptr, const int32_t *gid, const int32_t *rows, int64_t count,
                         int max_graph_nodes, int32_t *nbr, unsigned long long *skeys, hipStream_t s, int k = TD_K);
int td_launch_knn_merge(const float4 *x4, const int32_t *node_ptr, const int32_t *pptr, const int32_t *gid,
                        const int32_t *prot_rows, int64_t Np, const unsigned long long *skeys, const int32_t *snbr,
                        const float *h0, const float *h1s, const float *ews, int32_t *nbr, float *h, float *ew,
                        uint8_t *clean, uint8_t *flags2, hipStream_t s, int k = TD_K, const int32_t *lig_rows = nullptr, int64_t Nl = 0,
                        int max_graph_nodes = 0, const int32_t *cgraph = nullptr, const int32_t *cbase = nullptr);
// per-pocket sharing of a session's static tables (graph.hip, session.cpp)
int td_launch_pocket_hash(const float *ppos, const float *pv, const int32_t *pptr, int64_t B, int F, unsigned long long *out, hipStream_t s);
int td_launch_pocket_verify(const float *ppos, const float *pv, const int32_t *pptr, int64_t B, int F, const int32_t *cand, int32_t *flags,
                            hipStream_t s);
int td_launch_compact_rows(const void *src, const int32_t *rows, int64_t n_rows, int row_bytes, void *dst, hipStream_t s);
int td_launch_compact_dirty(const uint8_t *clean, const float4 *x4, int64_t N, int32_t *rows, int32_t *count,
                            hipStream_t s);
int td_launch_forward_reach(const uint8_t *clean, const float4 *x4, const int32_t *nbr, int64_t N, uint8_t *flags2,
                            int32_t *rows_on, int32_t *rows_off, int32_t *counts2, uint8_t *clean_to_zero, hipStream_t s);
int td_launch_restore_rows(const int32_t *rows, const int32_t *count_ptr, int64_t max_rows, const float *hs, float *h,
                           hipStream_t s, const int32_t *gid = nullptr, const int32_t *node_ptr = nullptr, const int32_t *cbase = nullptr);
constexpr int TD_HOP_LEVELS = 4;
int td_launch_hop_levels(const int32_t *lig_node, int64_t Nl, const int32_t *nbr, int64_t N, uint8_t *flags,
                         int32_t *rows, int32_t *counts, int levels, hipStream_t s, bool zeroed = false);
// edge gate on 16 x 16 tiles with the bf16 first layer (edge16.hip); td_launch_gate (gate.hip) dispatches to it
int td_launch_gate16(const TdGate &g, const float4 *x4, const int32_t *nbr, int64_t N, const int32_t *rows,
                     const int32_t *count_ptr, float *ew, hipStream_t s, const int32_t *chunk_node);
// all row lists of a sampling step in one launch (graph.hip); TD_EINVAL when a graph does not fit the LDS flag arrays
struct TdStepLists {
    int32_t *dirty_rows, *dirty_count;
    int32_t *reach_rows, *rest_rows, *reach_counts;
    int32_t *level_rows, *level_counts;
    int levels;
    int32_t *dirty_chunks = nullptr, *dirty_chunk_count = nullptr;     // general graphs: the chunks of the dirty rows
};
// cptr / chunk_node (general graphs): chunk-indexed neighbour table
int td_launch_step_lists(const uint8_t *clean, const float4 *x4, const int32_t *nbr, const int32_t *node_ptr, int64_t N,
                         int64_t B, int max_nodes, const TdStepLists &out, hipStream_t s, const int32_t *cptr = nullptr,
                         const int32_t *chunk_node = nullptr);
// bookkeeping a session step resets in its first kernel: up to three counter arrays and the ligand rows' forward-reach flags
struct TdStepReset {
    int32_t *c0 = nullptr, *c1 = nullptr, *c2 = nullptr;
    int n0 = 0, n1 = 0, n2 = 0;
    uint8_t *flags2 = nullptr;
};
int td_launch_ligand_update(const td_model *m, const float *lpos, const int64_t *lv, const int32_t *lig_node, int64_t Nl,
                            float *h, float4 *x4, hipStream_t s, const TdStepReset *reset = nullptr, const float *gbias = nullptr,
                            const int32_t *gid = nullptr);
int td_launch_ligand_list(const uint8_t *mask, int64_t N, int32_t *lig_node, int32_t *count, hipStream_t s);
// node.hip
// rows: optional list of node ids (N = its length); mat_mask bits 0..3 = [k_i, k_j, v_i, v_j] projections, bit 4 = query MLP
int td_launch_node_proj(const TdNodeStage &st, const float *h, int64_t N, const int32_t *rows, unsigned mat_mask,
                        float *P, float *q, hipStream_t s, const int32_t *count_ptr = nullptr,
                        const int32_t *rows2 = nullptr, int64_t N2 = 0, unsigned mask2 = 0);
int td_launch_node_proj_pair(const TdNodeStage &hx, const int32_t *hop_rows, const int32_t *hop_count,
                             const int32_t *lig_rows, int64_t Nl, float *Px, float *qx, const TdNodeStage &nx,
                             const int32_t *rows, const int32_t *count_ptr, float *P, float *q, const float *h, int64_t N,
                             hipStream_t s);
int td_launch_node_output(const TdNodeOut &no, const float *out, float *h, int64_t N, hipStream_t s);
int td_launch_layer_gate(const float *w, const float *offsets, float coeff, const float4 *x4, const int32_t *nbr, int64_t N, float *ew,
                         hipStream_t s);
// gate.hip -- rows: optional row list; chunk_node: dst node of every row of nbr / ew on general graphs (nullptr: row == node)
int td_launch_gate(const TdGate &g, const float4 *x4, const int32_t *nbr, int64_t N, const int32_t *rows,
                   const int32_t *count_ptr, float *ew, hipStream_t s, const int32_t *chunk_node = nullptr);
// edge16.hip.  cptr (general graphs): the in-edges of dst node i are the chunks cptr[i] .. cptr[i+1]-1 (32 slots each) of
// nbr / ew / alpha; nullptr on the default graph (one 32-slot row per node, chunk == node)
int td_launch_edge_key16(const TdEdgeMlp &mlp, const TdLayer &L, const float4 *x4, const int32_t *nbr, const float *ew,
                         const float *P, const float *q, const int32_t *rows, const int32_t *count_ptr, int64_t count,
                         float *alpha, hipStream_t s, bool h2x_stage, const int32_t *cptr = nullptr, const int32_t *lig_rows = nullptr,
                         int64_t lig_count = 0, int cpn_p = 0);
int td_launch_edge_h2x16(const TdEdgeMlp &mlp_k, const TdEdgeMlp &mlp_v, const TdLayer &L, const float4 *x4_in, float4 *x4_out,
                         const int32_t *nbr, const float *ew, const float *P, const float *q, const int32_t *rows,
                         int64_t count, hipStream_t s, const int32_t *cptr = nullptr, float *alpha = nullptr);
int td_launch_edge_xv16(const TdEdgeMlp &mlp, const TdLayer &L, const float4 *x4_in, float4 *x4_out, const int32_t *nbr,
                        const float *P, const int32_t *rows, int64_t count, const float *alpha, hipStream_t s,
                        const int32_t *cptr = nullptr);
int td_launch_edge_value16(const TdEdgeMlp &mlp, const TdLayer &L, const float4 *x4, const int32_t *nbr, const float *P,
                           const int32_t *rows, const int32_t *count_ptr, int64_t count, float *h, const float *alpha, const int32_t *lig_rows, int64_t lig_count,
                           hipStream_t s, const int32_t *cptr = nullptr, int cpn_p = 1, int64_t lig_chunks = 0,
                           const int32_t *mixed_count = nullptr, float *out = nullptr, const float *gate_m = nullptr);
int td_set_wg_trace(unsigned long long *buf, int slots);
bool td_wg_trace_armed();
// graph.hip, general graphs
int td_launch_layout(const int32_t *node_ptr, const int32_t *pptr, const int32_t *gid, const int32_t *g_cbase,
                     const int32_t *g_cl, const int32_t *g_lbase, int cpn_p, int64_t N, int32_t *cptr, int32_t *chunk_node,
                     int32_t *lig_chunks, int32_t total_chunks, hipStream_t s);
int td_launch_graph_general(int mode, const float4 *x4, const int32_t *node_ptr, const int32_t *pptr, const int32_t *gid,
                            const int32_t *prot_node, int64_t Np, const int32_t *lig_node, int64_t Nl, int64_t N, int k,
                            float radius, int max_graph_nodes, const int32_t *cptr, int32_t *cnbr, int64_t NC, hipStream_t s);
int td_launch_knn_general_static(const float4 *x4, const int32_t *node_ptr, const int32_t *pptr, const int32_t *gid,
                                 const int32_t *prot_rows, int64_t Np, int k, int max_graph_nodes, const int32_t *cptr,
                                 int32_t *snbr, unsigned long long *skeys, hipStream_t s);
int td_launch_ligand_rows_general(int mode, const float4 *x4, const int32_t *node_ptr, const int32_t *pptr, const int32_t *gid,
                                  const int32_t *lig_node, int64_t Nl, int k, int max_graph_nodes, const int32_t *cptr,
                                  int32_t *cnbr, hipStream_t s);
int td_launch_hybrid_ligand_half(const int32_t *node_ptr, const int32_t *pptr, const int32_t *gid, const int32_t *lig_node,
                                 int64_t Nl, const int32_t *cptr, int32_t *cnbr, hipStream_t s);
int td_launch_knn_merge_general(const float4 *x4, const int32_t *node_ptr, const int32_t *pptr, const int32_t *gid,
                                const int32_t *prot_rows, int64_t Np, const unsigned long long *skeys, const int32_t *snbr,
                                const float *h0, const float *h1s, const float *ews, const int32_t *cptr, int cpn, int32_t *nbr,
                                float *h, float *ew, uint8_t *clean, uint8_t *flags2, hipStream_t s, int k);
int td_launch_radius32(const float4 *x4, const int32_t *node_ptr, const int32_t *gid, int64_t N, float radius, int cap,
                       int32_t *nbr, hipStream_t s);
int td_launch_slots_to_dense(const int32_t *cptr, const int32_t *cnbr, int64_t N, int width, int32_t *out, hipStream_t s);
// misc.hip
int td_launch_head(const TdHead &hd, const float *h, const float4 *x4, const int32_t *lig_node, int64_t Nl,
                   int classes, float *pred_pos, float *pred_v, float *lig_h, hipStream_t s);
int td_launch_posterior(const TdSchedules &sc, int T, const int32_t *t, const int32_t *lptr, int64_t Nl, int64_t B,
                        int classes, const float *pos, const int64_t *v, const float *pred_pos,
                        const float *pred_v, const float *noise, const float *uni, float *pos_next,
                        int64_t *v_next, float *log_v0, float *log_post, hipStream_t s, int mean_type = 0);
int td_launch_posterior_step(const TdSchedules &sc, int T, int32_t *step, const int32_t *t_all, int num_steps, const int32_t *lptr,
                             int64_t Nl, int64_t B, int classes, float *pos, int64_t *v, const float *pred_pos, const float *pred_v,
                             const float *noise, const float *uni, float *pos_traj, int64_t *v_traj, float *v0_traj, float *vt_traj,
                             int pos_only, hipStream_t s, int mean_type = 0);
// egnn.hip / node.hip
int td_launch_egnn_edge(const TdEgnnLayer &L, const float4 *x4, float4 *x4_out, const int32_t *nbr, const float *P, float *mi,
                        int64_t N, hipStream_t s);
int td_launch_egnn_node(const TdEgnnLayer &L, const float *mi, float *h, int64_t N, hipStream_t s);
// likelihood.hip
int td_launch_perturb(const TdSchedules &sc, int T, const int32_t *t, const int32_t *lptr, int64_t Nl, int64_t B, int classes,
                      const float *pos, const int64_t *v, const float *noise, const float *uni, float *pos_t, int64_t *v_t,
                      hipStream_t s);
int td_launch_likelihood_terms(const TdSchedules &sc, int T, const int32_t *t, const int32_t *lptr, int64_t B, int classes,
                               const float *x0, const float *xt, const int64_t *v0, const int64_t *vt, const float *pred_pos,
                               const float *pred_v, float *kl_pos, float *kl_v, hipStream_t s);
int td_launch_likelihood_prior(const TdSchedules &sc, int T, const int32_t *lptr, int64_t B, int classes, const float *x0,
                               const int64_t *vidx, float *kl_pos, float *kl_v, hipStream_t s);
int td_launch_embed_ligand(const TdEmbed &emb, int classes, const int64_t *lv, int64_t Nl, float *h, hipStream_t s);
int td_launch_center(float *ppos, const int32_t *pptr, float *lpos, const int32_t *lptr, int64_t B, float *offset,
                     int compute, int sign, hipStream_t s);
int td_launch_reductions(const float *in, float *out, hipStream_t s);
