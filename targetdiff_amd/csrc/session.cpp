// C ABI of libtargetdiff_hip.so, part 5 of 5: the sampling session (one ScorePosNet3D.sample_diffusion call): static-protein caching,
// the per-step launch sequence and its hipGraph capture.
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <vector>

#include "td_device.h"
#include "td_internal.h"
#include "td_api.h"

using namespace tdapi;

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
    // Per-pocket sharing of the static tables (CACHING, default graph; model option session_share_pockets): all samples of a pocket carry the
    // same protein block (scripts/sample_diffusion.py:42 replicates one pocket n_data times), so skeys / snbr / ews / h0 / h1s / h2s are kept
    // ONCE per distinct block -- `static_rows` compact rows, the canonical graphs' protein atoms back to back -- and graph g reads the rows of
    // cgraph[g], the first graph of the batch with the same block, from cbase[g] on.  P0 / q0 (the layer-0 projections the attention passes
    // gather through the neighbour index) stay per node.  nullptr: tables indexed by node.
    int32_t *cgraph = nullptr, *cbase = nullptr;
    int64_t static_rows = 0;
    int pocket_groups = 0;
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

namespace tdapi {
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
    // the previous replay may still be in flight on the stream it was launched on (an option change between two steps lands here):
    // whether destroying an executing graph is deferred depends on the runtime version, so wait for it.  Cold path.
    if (S->graph_exec) (void)hipStreamSynchronize(S->last_stream);
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
}  // namespace tdapi

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
    // ---- per-pocket sharing of the static tables: which graphs carry the same protein block (bit for bit)?
    const bool share = S->caching && !S->chunked && m->opt.session_share_pockets != 0;
    std::vector<int32_t> cgraph_h, cbase_h, canon_rows_h;
    int64_t Nc = N;                                          // rows of the static tables (compact: the canonical graphs' protein atoms)
    if (share) {
        const int F = m->cfg.protein_feat_dim;
        char *tg = nullptr;
        const size_t tg_bytes = align_up((size_t)B * 8) + 2 * align_up((size_t)B * 4);
        hipError_t te = td_malloc_async(reinterpret_cast<void **>(&tg), tg_bytes, s);
        if (te != hipSuccess) { td_set_error("td_session_create: hipMallocAsync(%zu) failed: %s", tg_bytes, hipGetErrorString(te)); delete S; return TD_ENOMEM; }
        unsigned long long *d_hash = reinterpret_cast<unsigned long long *>(tg);
        int32_t *d_cand = reinterpret_cast<int32_t *>(tg + align_up((size_t)B * 8)), *d_flag = reinterpret_cast<int32_t *>(tg + align_up((size_t)B * 8) + align_up((size_t)B * 4));
        std::vector<unsigned long long> hash((size_t)B);
        std::vector<int32_t> flag((size_t)B, 0);
        cgraph_h.assign((size_t)B, 0); cbase_h.assign((size_t)B, 0);
        auto bail = [&](int r) { free_async_or_sync(tg, s); delete S; return r; };
        if ((rc = td_launch_pocket_hash(d_protein_pos, d_protein_v, d_protein_ptr, B, F, d_hash, s)) != TD_OK) return bail(rc);
        if (hipMemcpyAsync(hash.data(), d_hash, (size_t)B * 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
            td_set_error("td_session_create: reading the pocket hashes failed"); return bail(TD_EHIP);
        }
        {   // candidate = the first graph of the batch with the same (atom count, hash): an ordered map, B log B for any mix of pockets
            std::map<std::pair<int32_t, unsigned long long>, int32_t> first;
            for (int64_t g = 0; g < B; ++g)
                cgraph_h[(size_t)g] = first.emplace(std::make_pair(hp[g + 1] - hp[g], hash[(size_t)g]), (int32_t)g).first->second;
        }
        hipError_t e2 = hipMemcpyAsync(d_cand, cgraph_h.data(), (size_t)B * 4, hipMemcpyHostToDevice, s);
        if (e2 == hipSuccess) e2 = hipMemsetAsync(d_flag, 0, (size_t)B * 4, s);
        if (e2 != hipSuccess) { td_set_error("td_session_create: pocket grouping failed: %s", hipGetErrorString(e2)); return bail(TD_EHIP); }
        if ((rc = td_launch_pocket_verify(d_protein_pos, d_protein_v, d_protein_ptr, B, F, d_cand, d_flag, s)) != TD_OK) return bail(rc);
        if (hipMemcpyAsync(flag.data(), d_flag, (size_t)B * 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
            td_set_error("td_session_create: reading the pocket comparison failed"); return bail(TD_EHIP);
        }
        free_async_or_sync(tg, s);
        Nc = 0;
        for (int64_t g = 0; g < B; ++g) {
            if (flag[(size_t)g]) cgraph_h[(size_t)g] = (int32_t)g;          // same hash, different block: keeps its own tables
            const int32_t c = cgraph_h[(size_t)g];
            if (c == (int32_t)g) {
                cbase_h[(size_t)g] = (int32_t)Nc;
                const int32_t n0 = hp[g] + hl[g], np = hp[g + 1] - hp[g];          // the graph's nodes start at node_ptr[g]; protein atoms first
                for (int32_t a = 0; a < np; ++a) canon_rows_h.push_back(n0 + a);
                Nc += np;
                ++S->pocket_groups;
            } else {
                cbase_h[(size_t)g] = cbase_h[(size_t)c];
            }
        }
        S->static_rows = Nc;
    }

    // ---- one device block: [workspace | session-static buffers]
    const size_t ws_bytes = carve(nullptr, N, B, N_l).bytes;
    size_t off = ws_bytes;
    auto reserve = [&](size_t n) { size_t o = off; off += align_up(n ? n : 4); return o; };
    const size_t n = (size_t)N, nc = (size_t)NC;
    const bool C = S->caching;
    const size_t ns = share ? (size_t)Nc : n, ncs = share ? (size_t)Nc : nc;          // rows of the static tables
    const size_t o_prot = reserve((size_t)N_p * 4), o_pptr = reserve((size_t)(B + 1) * 4), o_lptr = reserve((size_t)(B + 1) * 4),
                 o_h0 = reserve(ns * TD_H * 4), o_tmp_lpos = reserve((size_t)N_l * 12), o_tmp_lv = reserve((size_t)N_l * 8),
                 o_snbr = reserve(C ? ncs * TD_K * 4 : 0), o_skeys = reserve(C ? ns * KS * 8 : 0), o_ews = reserve(C ? ncs * TD_K * 4 : 0),
                 o_h1s = reserve(C ? ns * TD_H * 4 : 0), o_h2s = reserve(C ? ns * TD_H * 4 : 0), o_f2 = reserve(C ? n : 0),
                 o_frows = reserve(C ? n * 4 : 0), o_frest = reserve(C ? n * 4 : 0), o_fcnt = reserve(256),
                 o_P0 = reserve(C ? n * 4 * TD_H * 4 : 0), o_q0 = reserve(C ? n * TD_H * 4 : 0), o_clean = reserve(C ? n : 0),
                 o_dirty = reserve(C ? n * 4 : 0), o_dcnt = reserve(256), o_hop = reserve(C ? n * 4 * TD_HOP_LEVELS : 0),
                 o_hcnt = reserve(256), o_dchunks = reserve(C && S->chunked ? nc * 4 : 0),
                 o_ppos = reserve((size_t)N_l * 3 * 4), o_pv = reserve((size_t)N_l * TD_MAXC * 4),
                 o_cg = reserve(share ? (size_t)B * 4 : 0), o_cb = reserve(share ? (size_t)B * 4 : 0), o_crows = reserve(share ? (size_t)Nc * 4 : 0);
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
    int32_t *canon_rows = nullptr;
    if (share) {
        S->cgraph = reinterpret_cast<int32_t *>(b + o_cg);
        S->cbase = reinterpret_cast<int32_t *>(b + o_cb);
        canon_rows = reinterpret_cast<int32_t *>(b + o_crows);
    }
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
    // Sharing: the set-up below runs on the canonical graphs' protein rows only, into full-size (node-indexed) scratch tables that live in one
    // stream-ordered block until their canonical rows have been copied into the compact shared tables
    char *scratch = nullptr;
    float *h0_f = S->h0, *h1_f = S->h1s, *h2_f = S->h2s, *ews_f = S->ews;
    int32_t *snbr_f = S->snbr;
    unsigned long long *skeys_f = S->skeys;
    const int32_t *set_rows = S->prot_node;          // (assigned below for chunked plans)
    int64_t set_count = N_p;
    auto fail2 = [&](int r) { free_async_or_sync(scratch, s); return fail(r); };
    if (share) {
        const size_t b_keys = align_up(n * TD_K * 8), b_nbr = align_up(n * TD_K * 4), b_h = align_up(n * TD_H * 4);
        const size_t sb = b_keys + 2 * b_nbr + 3 * b_h;
        hipError_t se = td_malloc_async(reinterpret_cast<void **>(&scratch), sb, s);
        if (se != hipSuccess) { td_set_error("td_session_create: hipMallocAsync(%zu) failed: %s", sb, hipGetErrorString(se)); scratch = nullptr; return fail(TD_ENOMEM); }
        skeys_f = reinterpret_cast<unsigned long long *>(scratch);
        snbr_f = reinterpret_cast<int32_t *>(scratch + b_keys);
        ews_f = reinterpret_cast<float *>(scratch + b_keys + b_nbr);
        h0_f = reinterpret_cast<float *>(scratch + b_keys + 2 * b_nbr);
        h1_f = reinterpret_cast<float *>(scratch + b_keys + 2 * b_nbr + b_h);
        h2_f = reinterpret_cast<float *>(scratch + b_keys + 2 * b_nbr + 2 * b_h);
        hipError_t ue = hipMemcpyAsync(S->cgraph, cgraph_h.data(), (size_t)B * 4, hipMemcpyHostToDevice, s);
        if (ue == hipSuccess) ue = hipMemcpyAsync(S->cbase, cbase_h.data(), (size_t)B * 4, hipMemcpyHostToDevice, s);
        if (ue == hipSuccess) ue = hipMemcpyAsync(canon_rows, canon_rows_h.data(), (size_t)Nc * 4, hipMemcpyHostToDevice, s);
        if (ue == hipSuccess) ue = hipStreamSynchronize(s);          // (the host vectors go out of scope with this call)
        if (ue != hipSuccess) { td_set_error("td_session_create: uploading the pocket groups failed: %s", hipGetErrorString(ue)); return fail2(TD_EHIP); }
    }
#undef TD_TRY
#undef TD_TRY_HIP
#define TD_TRY(expr) do { if ((rc = (expr)) != TD_OK) return fail2(rc); } while (0)
#define TD_TRY_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { td_set_error("%s failed: %s", #expr, hipGetErrorString(_e)); return fail2(TD_EHIP); } } while (0)
    // embeddings + packed order (ligand rows are placeholders until the first step), protein row list
    int32_t *prot_out = S->chunked ? S->plan.prot_node : S->prot_node;
    TD_TRY(td_launch_compose(m, d_protein_pos, d_protein_v, S->pptr, N_p, tmp_lpos, tmp_lv, S->lptr, N_l, B, h0_f, w.x4a,
                             w.node_ptr, w.gid, w.lig_node, prot_out, s));
    if (S->chunked) {
        S->prot_node = S->plan.prot_node;
        TD_TRY(plan_layout(S->plan, w.node_ptr, w.gid, s));
    }
    set_rows = share ? canon_rows : S->prot_node;
    set_count = share ? Nc : N_p;
    TD_TRY_HIP(hipMemcpyAsync(w.x4b, w.x4a, n * sizeof(float4), hipMemcpyDeviceToDevice, s));
    if (!S->caching) {
        *out = S;
        return TD_OK;
    }
    // ---- protein-only graph, its gate rows, layer-0 projections / queries, layer-0 x2h output
    GraphTab gt = session_tab(S);                  // the step's table (alpha is scratch here)
    GraphTab st = gt;                              // the static table
    st.nbr = snbr_f; st.ew = ews_f;
    TD_TRY_HIP(hipMemsetAsync(snbr_f, 0xff, nc * TD_K * 4, s));
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
        TD_TRY(td_launch_knn_static(w.x4a, w.node_ptr, w.gid, set_rows, set_count, max_graph_nodes, snbr_f, skeys_f, s, m->cfg.knn));
        TD_TRY(td_launch_gate(m->gate, w.x4a, snbr_f, set_count, set_rows, nullptr, ews_f, s));
    }
    const TdLayer &L0 = m->layers[0];
    TD_TRY(td_launch_node_proj(L0.nodeX2h, h0_f, N, nullptr, 0x1f, S->P0, S->q0, s));
    TD_TRY(key_pass(L0.hk, L0, w.x4a, st, st.ew, st.nbr, S->P0, S->q0, set_rows, nullptr, set_count, gt.alpha, s));
    TD_TRY_HIP(hipMemcpyAsync(h1_f, h0_f, n * TD_H * 4, hipMemcpyDeviceToDevice, s));
    TD_TRY(value_pass(L0.hv, L0, w.x4a, st, st.nbr, S->P0, set_rows, nullptr, set_count, h1_f, gt.alpha, nullptr, 0, s));
    if (S->use_fwd) {      // layer-1 x2h output of the protein-only graph (valid wherever the ligand is two hops away)
        const TdLayer &L1 = m->layers[1];
        TD_TRY(td_launch_node_proj(L1.nodeX2h, h1_f, N, nullptr, 0x1f, w.P, w.q, s));
        TD_TRY(key_pass(L1.hk, L1, w.x4a, st, st.ew, st.nbr, w.P, w.q, set_rows, nullptr, set_count, gt.alpha, s));
        TD_TRY_HIP(hipMemcpyAsync(h2_f, h1_f, n * TD_H * 4, hipMemcpyDeviceToDevice, s));
        TD_TRY(value_pass(L1.hv, L1, w.x4a, st, st.nbr, w.P, set_rows, nullptr, set_count, h2_f, gt.alpha, nullptr, 0, s));
    }
    if (share) {           // the canonical rows of the scratch tables -> the compact shared tables; the scratch block goes back in stream order
        TD_TRY(td_launch_compact_rows(skeys_f, canon_rows, Nc, TD_K * 8, S->skeys, s));
        TD_TRY(td_launch_compact_rows(snbr_f, canon_rows, Nc, TD_K * 4, S->snbr, s));
        TD_TRY(td_launch_compact_rows(ews_f, canon_rows, Nc, TD_K * 4, S->ews, s));
        TD_TRY(td_launch_compact_rows(h0_f, canon_rows, Nc, TD_H * 4, S->h0, s));
        TD_TRY(td_launch_compact_rows(h1_f, canon_rows, Nc, TD_H * 4, S->h1s, s));
        if (S->use_fwd) TD_TRY(td_launch_compact_rows(h2_f, canon_rows, Nc, TD_H * 4, S->h2s, s));
        free_async_or_sync(scratch, s);
        scratch = nullptr;
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
                                          w.lig_node, Nl, S->max_graph_nodes, S->cgraph, S->cbase)) != TD_OK) return rc;
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
    const FwdReach fwd{S->fwd_rows, S->fwd_rest, S->fwd_counts, S->h2s, S->cbase};
    if ((rc = run_backbone(m, w, gt, w.h, N, Nl, 0, &xf, s, false, true, S->hop_rows, S->hop_count, S->hop_levels,
                           S->use_fwd ? &fwd : nullptr)) != TD_OK) return rc;
    ProfScope ps(PC_HEAD, s);
    return td_launch_head(m->head, w.h, xf, w.lig_node, Nl, m->cfg.ligand_num_classes, d_pred_ligand_pos,
                          d_pred_ligand_v, d_final_ligand_h, s);
}

namespace tdapi {
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
}  // namespace tdapi

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
    // rows of the static tables and distinct protein blocks of the batch (per-pocket sharing; -1: tables indexed by node)
    if (n_counts > 3 + TD_HOP_LEVELS) host_counts[3 + TD_HOP_LEVELS] = S->cgraph ? (int32_t)S->static_rows : -1;
    if (n_counts > 4 + TD_HOP_LEVELS) host_counts[4 + TD_HOP_LEVELS] = S->cgraph ? (int32_t)S->pocket_groups : -1;
    TD_CHECK_HIP(hipStreamSynchronize(s));
    return TD_OK;
}

