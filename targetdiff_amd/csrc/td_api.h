// Internal declarations shared by the five translation units of the C ABI (entry.cpp, pack.cpp, plan.cpp, forward.cpp, session.cpp).
// Not part of the interface: include/targetdiff_hip.h is.
#pragma once

#include <atomic>
#include <mutex>
#include <vector>

#include "td_internal.h"

void td_set_error(const char *fmt, ...);

namespace tdapi {

// ---- per-class kernel timers (entry.cpp): HIP events on the launch stream around the selected classes
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
extern Profiler g_prof;
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

// ---- the weight blob (pack.cpp)
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

inline size_t mlp_floats(int in, int hid, int out) { return (size_t)hid * in + 3 * (size_t)hid + (size_t)out * hid + out; }

// stages per layer (0 in the config = 1): m->layers has stage_rows(c) rows per reference layer
inline int num_x2h(const td_config &c) { return c.num_x2h > 0 ? c.num_x2h : 1; }
inline int num_h2x(const td_config &c) { return c.num_h2x > 0 ? c.num_h2x : 1; }
inline int stage_rows(const td_config &c) { return num_x2h(c) > num_h2x(c) ? num_x2h(c) : num_h2x(c); }

inline int kv_in(const td_config &c) { return 2 * c.hidden_dim + c.edge_feat_dim + 4 * c.num_r_gaussian; }

inline bool config_supported(const td_config &c) {
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
inline bool caching_graph(const td_config &c) { return c.cutoff_mode == TD_CUTOFF_KNN && c.knn <= TD_K; }
inline bool default_graph(const td_config &c) {
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

size_t pack_B128(Packer &pk, const float *W, int ld, int col0);
size_t pack_B128_split(Packer &pk, const float *W, int ld, int col0);
size_t pack_vec(Packer &pk, const float *v, size_t n, size_t padded = 0);
float gaussian_coeff(const float *off);
inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// ---- workspace, graph tables, the layer sequence (plan.cpp)
struct Workspace {
    float4 *x4a, *x4b;
    int32_t *gid, *nbr, *lig_node, *node_ptr;
    float *ew, *P, *q, *h, *alpha;
    float *Px, *qx;          // h2x-stage projections / queries (separate from P / q: both stages project in one launch)
    size_t bytes;
};

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

// Sampling session, layer 1: rows outside the ligand's one-hop forward reach keep the protein-only graph's layer-1
// output (`hs`), so the attention passes run on `rows` only and `rest` is restored from the cache afterwards.
struct FwdReach {
    const int32_t *rows, *rest, *counts;     // counts[0] = |rows|, counts[1] = |rest|
    const float *hs;
    const int32_t *cbase = nullptr;          // hs is a compact per-pocket table (session.cpp): row i of graph g = cbase[g] + (i - node_ptr[g])
};

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

Workspace carve(char *base, int64_t N, int64_t B, int64_t Nl);
inline GraphTab default_tab(Workspace &w) { return GraphTab{nullptr, w.nbr, w.ew, w.alpha, 1, 0, nullptr}; }
inline GraphTab plan_tab(const GraphPlan &p) { return GraphTab{p.cptr, p.cnbr, p.ew, p.alpha, p.cpn_p, p.NCl, nullptr}; }
inline int num_blocks(const td_config &c) { return c.num_blocks > 1 ? c.num_blocks : 1; }
// lig / Nl (x2h passes): the ligand rows of the batch, all of them among `rows`
int key_pass(const TdEdgeMlp &mlp, const TdLayer &L, const float4 *x4, const GraphTab &gt, const float *ew, const int32_t *nbr,
             const float *P, const float *q, const int32_t *rows, const int32_t *count_ptr, int64_t count, float *alpha, hipStream_t s,
             const int32_t *lig = nullptr, int64_t Nl = 0, bool h2x_stage = false);
int value_pass(const TdEdgeMlp &mlp, const TdLayer &L, const float4 *x4, const GraphTab &gt, const int32_t *nbr, const float *P,
               const int32_t *rows, const int32_t *count_ptr, int64_t count, float *h, const float *alpha, const int32_t *lig,
               int64_t Nl, hipStream_t s, float *out = nullptr);
int h2x_project(const TdLayer &L, Workspace &w, float *h, int64_t N, int64_t Nl, float *P, float *q,
                const int32_t *hop_rows, const int32_t *hop_count, hipStream_t s);
int h2x_attend(const td_model *m, const TdLayer &L, Workspace &w, const GraphTab &gt, int64_t Nl, float4 *xc, float4 *xn, float *P,
               float *q, hipStream_t s);
// L x (node_proj, x2h, node_proj, h2x) on a composed batch whose graph (gt.nbr) and edge gate (gt.ew) are in place.  h is
// updated in place; returns the buffer holding the final coordinates through *x_final.  init_xn: x4b is not yet a copy of x4a.
// hop_rows (optional, sampling session): the ligand atoms and their neighbours -- the only rows whose h2x-stage projections and
// last-layer features are ever read when just the ligand outputs are consumed.
int run_backbone(const td_model *m, Workspace &w, const GraphTab &gt, float *h, int64_t N, int64_t Nl, int fix_x,
                 float4 **x_final, hipStream_t s, bool init_xn, bool layer0_x2h_done = false,
                 const int32_t *hop_rows = nullptr, const int32_t *hop_count = nullptr, int hop_levels = 0,
                 const FwdReach *fwd = nullptr);
int build_default_graph(const td_model *m, Workspace &w, int64_t N, int max_graph_nodes, hipStream_t s);
// stream-ordered memory: td_debug_fail_alloc makes the n-th allocation from now fail (fault injection for the error paths)
extern std::atomic<int> g_fail_alloc;
hipError_t td_malloc_async(void **p, size_t bytes, hipStream_t s);
void free_async_or_sync(void *p, hipStream_t s);          // (session.cpp)
void plan_destroy(GraphPlan &p, hipStream_t s);
int plan_create(const td_config &c, const int32_t *host_pptr, const int32_t *host_lptr, int64_t B, hipStream_t s, GraphPlan *out);
int plan_layout(GraphPlan &p, const int32_t *node_ptr, const int32_t *gid, hipStream_t s);
int build_general_graph(const td_model *m, GraphPlan &p, Workspace &w, int64_t N, int64_t Nl, int max_graph_nodes, hipStream_t s);
int plan_from_mask(const td_config &c, const uint8_t *d_mask, const int32_t *d_node_ptr, int64_t N, int64_t B, hipStream_t s,
                   GraphPlan *out, int64_t *nl_out);
int fetch_ptrs(const int32_t *d_a, const int32_t *d_b, int64_t B, std::vector<int32_t> &a, std::vector<int32_t> &b, hipStream_t s);
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

}  // namespace tdapi
