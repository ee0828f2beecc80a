// C ABI of libtargetdiff_hip.so, part 3 of 5: workspace carving, the layer sequence of one denoiser evaluation (shared by the stateless
// entry points and the sampling session) and the chunked layout of general graphs (GraphPlan).
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

// ------------------------------------------------------------------------------------------ workspace
namespace tdapi {

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


// lig / Nl (x2h passes): the ligand rows of the batch, all of them among `rows`
int key_pass(const TdEdgeMlp &mlp, const TdLayer &L, const float4 *x4, const GraphTab &gt, const float *ew, const int32_t *nbr,
             const float *P, const float *q, const int32_t *rows, const int32_t *count_ptr, int64_t count, float *alpha, hipStream_t s,
             const int32_t *lig, int64_t Nl, bool h2x_stage) {
    return td_launch_edge_key16(mlp, L, x4, nbr, ew, P, q, rows, count_ptr, count, alpha, s, h2x_stage, gt.cptr, lig, Nl, gt.cpn_p);
}
// lig / Nl: the ligand rows of the batch (all of them are among `rows`: every row list of a step contains the ligand atoms)
int value_pass(const TdEdgeMlp &mlp, const TdLayer &L, const float4 *x4, const GraphTab &gt, const int32_t *nbr, const float *P,
               const int32_t *rows, const int32_t *count_ptr, int64_t count, float *h, const float *alpha, const int32_t *lig,
               int64_t Nl, hipStream_t s, float *out) {
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


int run_backbone_stages(const td_model *m, Workspace &w, const GraphTab &gt, float *h, int64_t N, int64_t Nl, int fix_x,
                        float4 **x_final, hipStream_t s, bool init_xn);

int run_backbone(const td_model *m, Workspace &w, const GraphTab &gt, float *h, int64_t N, int64_t Nl, int fix_x,
                 float4 **x_final, hipStream_t s, bool init_xn, bool layer0_x2h_done,
                 const int32_t *hop_rows, const int32_t *hop_count, int hop_levels, const FwdReach *fwd) {
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
            if (use_fwd && (rc = td_launch_restore_rows(fwd->rest, fwd->counts + 1, N, fwd->hs, h, s, w.gid, w.node_ptr, fwd->cbase)) != TD_OK) return rc;
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


// td_debug_fail_alloc: the n-th stream-ordered allocation from now fails (fault injection for the error paths)
std::atomic<int> g_fail_alloc{0};
hipError_t td_malloc_async(void **p, size_t bytes, hipStream_t s) {
    int n = g_fail_alloc.load(std::memory_order_relaxed);
    while (n > 0 && !g_fail_alloc.compare_exchange_weak(n, n - 1)) {}
    if (n == 1) { *p = nullptr; return hipErrorOutOfMemory; }
    return hipMallocAsync(p, bytes, s);
}
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
}  // namespace tdapi

extern "C" size_t td_workspace_bytes(const td_model *m, int64_t N, int64_t B, int64_t N_l) {
    (void)m;
    return carve(nullptr, N, B, N_l).bytes;
}
