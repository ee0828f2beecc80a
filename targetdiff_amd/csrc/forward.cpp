// C ABI of libtargetdiff_hip.so, part 4 of 5: the stateless entry points (graph construction, backbone, denoiser, posterior,
// likelihood terms, embeddings), the standalone EGNN refine net and the debug hooks.
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

namespace tdapi {
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
}  // namespace tdapi

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

namespace tdapi {
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
}  // namespace tdapi

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
