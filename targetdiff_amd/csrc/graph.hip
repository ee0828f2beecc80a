// Graph bookkeeping, context composition and the batched k-nearest-neighbour search (gfx950).
//
// Replaces on the reference path:
//   * compose_context                      models/common.py:120-137
//   * atom embeddings + node indicator     models/molopt_score_model.py:333-338
//   * knn_graph(x, k=32, batch)            models/uni_transformer.py:280 (torch_cluster 1.6.0, external)
#include "td_device.h"
#include "td_internal.h"

// ------------------------------------------------------------------------------------------ graph_ptr
__global__ void graph_ptr_kernel(const int64_t *__restrict__ batch, int64_t N, int64_t B, int32_t *__restrict__ ptr) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (N == 0) {
        if (i <= B) ptr[i] = 0;
        return;
    }
    if (i >= N) return;
    int64_t b = batch[i];
    int64_t prev = (i == 0) ? -1 : batch[i - 1];
    for (int64_t g = prev + 1; g <= b && g <= B; ++g) ptr[g] = (int32_t)i;
    if (i == N - 1)
        for (int64_t g = b + 1; g <= B; ++g) ptr[g] = (int32_t)N;
}

int td_launch_graph_ptr(const int64_t *batch, int64_t N, int64_t B, int32_t *ptr, hipStream_t s) {
    int64_t n = N > B + 1 ? N : B + 1;
    graph_ptr_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(batch, N, B, ptr);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

__device__ __forceinline__ int td_find_graph(const int32_t *__restrict__ ptr, int B, int i) {
    int lo = 0, hi = B;   // invariant: ptr[lo] <= i < ptr[hi]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (ptr[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ void node_gid_kernel(const int32_t *__restrict__ ptr, int64_t N, int B, int32_t *__restrict__ gid) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) gid[i] = td_find_graph(ptr, B, (int)i);
}

int td_launch_node_gid(const int32_t *node_ptr, int64_t N, int64_t B, int32_t *gid, hipStream_t s) {
    if (N == 0) return TD_OK;
    node_gid_kernel<<<dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s>>>(node_ptr, N, (int)B, gid);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

// ------------------------------------------------------------------------------------------ x packing
// Internal coordinate layout: float4 (x, y, z, is_ligand) -- one 16-byte gather per neighbour.
__global__ void pack_x_kernel(const float *__restrict__ x3, const uint8_t *__restrict__ mask, int64_t N,
                              float4 *__restrict__ x4) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) x4[i] = make_float4(x3[3 * i], x3[3 * i + 1], x3[3 * i + 2], mask[i] ? 1.f : 0.f);
}
__global__ void unpack_x_kernel(const float4 *__restrict__ x4, int64_t N, float *__restrict__ x3) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) {
        float4 v = x4[i];
        x3[3 * i] = v.x; x3[3 * i + 1] = v.y; x3[3 * i + 2] = v.z;
    }
}
int td_launch_pack_x(const float *x3, const uint8_t *mask, int64_t N, float4 *x4, hipStream_t s) {
    if (N == 0) return TD_OK;
    pack_x_kernel<<<dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s>>>(x3, mask, N, x4);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}
int td_launch_unpack_x(const float4 *x4, int64_t N, float *x3, hipStream_t s) {
    if (N == 0) return TD_OK;
    unpack_x_kernel<<<dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s>>>(x4, N, x3);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

__global__ void ligand_list_kernel(const uint8_t *__restrict__ mask, int64_t N, int32_t *__restrict__ lig_node,
                                   int32_t *__restrict__ count) {
    // single-block ordered compaction (N <= a few million; once per refine call, not per layer)
    __shared__ int s_base;
    __shared__ int s_cnt[1024];
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int64_t start = 0; start < N; start += blockDim.x) {
        int64_t i = start + threadIdx.x;
        int f = (i < N && mask[i]) ? 1 : 0;
        s_cnt[threadIdx.x] = f;
        __syncthreads();
        for (int off = 1; off < (int)blockDim.x; off <<= 1) {       // inclusive scan
            int v = (threadIdx.x >= (unsigned)off) ? s_cnt[threadIdx.x - off] : 0;
            __syncthreads();
            s_cnt[threadIdx.x] += v;
            __syncthreads();
        }
        if (f) lig_node[s_base + s_cnt[threadIdx.x] - 1] = (int32_t)i;
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1) s_base += s_cnt[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0 && count) *count = s_base;
}
int td_launch_ligand_list(const uint8_t *mask, int64_t N, int32_t *lig_node, int32_t *count, hipStream_t s) {
    ligand_list_kernel<<<dim3(1), dim3(1024), 0, s>>>(mask, N, lig_node, count);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

// ------------------------------------------------------------------------------------------ kNN
// One wavefront per query node.  Lane l scores the candidates beg + l, beg + l + 64, ... of the query's
// graph; key = (fp32 bits of d2) << 32 | global index, so the unsigned 64-bit order is exactly the
// (d2, index) order of the project's kNN rule.  The 32 smallest keys are extracted by 32 rounds of
// {per-lane minimum, 6-step butterfly wave minimum, winner retires its candidate}; round r's key ends up
// in lane r, i.e. the row comes out sorted ascending.  Graphs larger than 64*CH nodes are scanned in
// passes; the running best-32 (one per lane 0..31) joins the next pass as an extra candidate.
constexpr unsigned long long TD_KEY_MAX = ~0ull;
constexpr int TD_COMPOSE_ATOMS = 16;

// STATIC = true builds the protein-only neighbour lists of a sampling session: ligand candidates (x4.w > 0.5) are
// skipped and the sorted keys are kept (skeys) so that later steps only have to merge the few ligand atoms in.
template <int CH, bool STATIC>
__device__ __forceinline__ void knn_body(unsigned bid, const float4 *__restrict__ x4, const int32_t *__restrict__ ptr,
                                         const int32_t *__restrict__ gid, const int32_t *__restrict__ rows,
                                         int64_t N, int k, int32_t *__restrict__ nbr,
                                         unsigned long long *__restrict__ skeys) {
    // k <= 32 neighbours per row in a 32-slot row (slots >= k: -1).  The k nearest are the first k of the 32 nearest, so
    // every fan-in up to 32 shares the 32-slot fast path; STATIC extracts all 32 keys (the session merges into them).
    const int rounds = STATIC ? TD_K : k;
    const int lane = threadIdx.x & 63;
    const int64_t qi = (int64_t)bid * 4 + (threadIdx.x >> 6);
    if (qi >= N) return;
    const int64_t i = rows ? (int64_t)rows[qi] : qi;
    const int g = gid[i];
    const int beg = ptr[g], end = ptr[g + 1];
    const float4 xi = x4[i];
    unsigned long long best = TD_KEY_MAX;    // lanes 0..31: current r-th smallest key
    for (int base = beg; base < end; base += 64 * CH) {
        unsigned long long key[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            int j = base + lane + 64 * u;
            key[u] = TD_KEY_MAX;
            if (j < end && j != (int)i) {
                float4 xj = x4[j];
                float d2 = td_dist2(xj.x - xi.x, xj.y - xi.y, xj.z - xi.z);
                if (!STATIC || xj.w <= 0.5f) key[u] = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)j;
            }
        }
        unsigned long long carry = best;     // previous passes' winners compete again
        unsigned long long out = TD_KEY_MAX;
        for (int r = 0; r < rounds; ++r) {
            unsigned long long lmin = carry;
#pragma unroll
            for (int u = 0; u < CH; ++u) lmin = key[u] < lmin ? key[u] : lmin;
            const unsigned long long wmin = td_wave_min_u64(lmin);
            if (wmin != TD_KEY_MAX) {        // keys are unique (they embed the index): exactly one owner
                if (carry == wmin) carry = TD_KEY_MAX;
#pragma unroll
                for (int u = 0; u < CH; ++u)
                    if (key[u] == wmin) key[u] = TD_KEY_MAX;
            }
            if (lane == r) out = wmin;
        }
        best = out;
    }
    if (lane < TD_K) {
        nbr[i * TD_K + lane] = (best == TD_KEY_MAX || lane >= k) ? -1 : (int32_t)(unsigned)(best & 0xffffffffull);
        if (STATIC) skeys[i * TD_K + lane] = best;
    }
}
template <int CH, bool STATIC>
__global__ __launch_bounds__(256) void knn_kernel(const float4 *__restrict__ x4, const int32_t *__restrict__ ptr,
                                                  const int32_t *__restrict__ gid, const int32_t *__restrict__ rows,
                                                  int64_t N, int k, int32_t *__restrict__ nbr,
                                                  unsigned long long *__restrict__ skeys) {
    knn_body<CH, STATIC>(blockIdx.x, x4, ptr, gid, rows, N, k, nbr, skeys);
}

template <bool STATIC>
static int launch_knn_t(const float4 *x4, const int32_t *node_ptr, const int32_t *gid, const int32_t *rows, int64_t N,
                        int max_graph_nodes, int k, int32_t *nbr, unsigned long long *skeys, hipStream_t s) {
    if (N == 0) return TD_OK;
    dim3 grid((unsigned)((N + 3) / 4)), block(256);
    if (max_graph_nodes > 0 && max_graph_nodes <= 256)
        knn_kernel<4, STATIC><<<grid, block, 0, s>>>(x4, node_ptr, gid, rows, N, k, nbr, skeys);
    else if (max_graph_nodes > 0 && max_graph_nodes <= 384)
        knn_kernel<6, STATIC><<<grid, block, 0, s>>>(x4, node_ptr, gid, rows, N, k, nbr, skeys);
    else if (max_graph_nodes <= 704)         // also the "unknown" (0) default
        knn_kernel<11, STATIC><<<grid, block, 0, s>>>(x4, node_ptr, gid, rows, N, k, nbr, skeys);
    else
        knn_kernel<17, STATIC><<<grid, block, 0, s>>>(x4, node_ptr, gid, rows, N, k, nbr, skeys);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

int td_launch_knn(const float4 *x4, const int32_t *node_ptr, const int32_t *gid, int64_t N, int max_graph_nodes,
                  int32_t *nbr, hipStream_t s, int k) {
    return launch_knn_t<false>(x4, node_ptr, gid, nullptr, N, max_graph_nodes, k, nbr, nullptr, s);
}

// kNN of the listed query rows only (ligand atoms of a session step): full search over the query's graph.
int td_launch_knn_rows(const float4 *x4, const int32_t *node_ptr, const int32_t *gid, const int32_t *rows, int64_t count,
                       int max_graph_nodes, int32_t *nbr, hipStream_t s, int k) {
    return launch_knn_t<false>(x4, node_ptr, gid, rows, count, max_graph_nodes, k, nbr, nullptr, s);
}

// Protein-only neighbour lists + their sorted keys for the listed (protein) rows.
int td_launch_knn_static(const float4 *x4, const int32_t *node_ptr, const int32_t *gid, const int32_t *rows, int64_t count,
                         int max_graph_nodes, int32_t *nbr, unsigned long long *skeys, hipStream_t s, int k) {
    return launch_knn_t<true>(x4, node_ptr, gid, rows, count, max_graph_nodes, k, nbr, skeys, s);
}

// ------------------------------------------------------------------------------------------ session step: kNN merge
// Protein atoms never move (models/uni_transformer.py:206), so a protein query's neighbours among protein atoms are
// step-invariant: per step only the <= ~90 ligand atoms of its graph are scored and merged into the static sorted
// list.  If no ligand atom beats the 32nd static neighbour the row is unchanged ("clean"): its gate row and its
// layer-0 output are step-invariant too and are copied from the session cache instead of being recomputed.
// One wave per protein row.  Ligand rows of a graph are contiguous: [node_ptr[g] + n_prot(g), node_ptr[g+1]).
// cgraph / cbase (a session that shares its static tables between the replicas of a pocket, session.cpp): graph g's protein block is a bitwise
// copy of graph cgraph[g]'s (the first such graph of the batch: cgraph[g] <= g), whose rows sit at cbase[g] .. in the COMPACT static tables
// skeys / snbr / ews / h0 / h1s; row i of graph g reads compact row cbase[g] + (i - ptr[g]) and shifts the node indices it finds there (they
// are the canonical graph's) by ptr[g] - ptr[cgraph[g]].  nullptr: the tables are indexed by node.
struct TdMergeArgs {
    const float4 *x4; const int32_t *ptr, *pptr, *gid, *prot_rows; int64_t Np;
    const unsigned long long *skeys; const int32_t *snbr; const float *h0, *h1s, *ews; int32_t *nbr; float *h, *ew;
    uint8_t *clean, *flags2; int k;
    const int32_t *cgraph, *cbase;
};
__device__ __forceinline__ void knn_merge_body(unsigned bid, const TdMergeArgs &ma) {
    const float4 *__restrict__ x4 = ma.x4;
    const int32_t *__restrict__ ptr = ma.ptr, *__restrict__ pptr = ma.pptr, *__restrict__ gid = ma.gid, *__restrict__ prot_rows = ma.prot_rows;
    const int64_t Np = ma.Np;
    const unsigned long long *__restrict__ skeys = ma.skeys;
    const int32_t *__restrict__ snbr = ma.snbr;
    const float *__restrict__ h0 = ma.h0, *__restrict__ h1s = ma.h1s, *__restrict__ ews = ma.ews;
    int32_t *__restrict__ nbr = ma.nbr;
    float *__restrict__ h = ma.h, *__restrict__ ew = ma.ew;
    uint8_t *__restrict__ clean = ma.clean, *__restrict__ flags2 = ma.flags2;
    const int k = ma.k;
    const int lane = threadIdx.x & 63;
    const int64_t qi = (int64_t)bid * 4 + (threadIdx.x >> 6);
    if (qi >= Np) return;
    const int64_t i = prot_rows[qi];
    const int g = gid[i];
    const int lbeg = ptr[g] + (pptr[g + 1] - pptr[g]), lend = ptr[g + 1];
    const float4 xi = x4[i];
    int64_t c = i;                 // row of the static tables
    int rebase = 0;                // ... and what turns the node indices stored there into this graph's
    if (ma.cgraph) {
        rebase = ptr[g] - ptr[ma.cgraph[g]];
        c = (int64_t)ma.cbase[g] + (i - ptr[g]);
    }
    unsigned long long ks = lane < TD_K ? skeys[c * TD_K + lane] : TD_KEY_MAX;
    if (ks != TD_KEY_MAX) ks += (unsigned long long)(unsigned)rebase;          // the node index is the key's low word (< 2^31: no carry)
    const unsigned long long thr = __shfl(ks, k - 1);             // k-th static neighbour (MAX if fewer exist)
    unsigned long long kl[2] = {TD_KEY_MAX, TD_KEY_MAX};
    bool closer = false;
    bool overflow = lend - lbeg > 128;                            // > 128 ligand atoms: handled by extra passes below
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int j = lbeg + lane + 64 * u;
        if (j < lend) {
            const float4 xj = x4[j];
            const float d2 = td_dist2(xj.x - xi.x, xj.y - xi.y, xj.z - xi.z);
            kl[u] = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)j;
            closer |= kl[u] < thr;
        }
    }
    bool is_clean = !overflow && __ballot(closer) == 0ull;
    if (!is_clean && !overflow) {
        // Merge by rank.  Only the ligand atoms below the k-th static key can enter the list (typically a handful): each is broadcast
        // once (wave-uniform key), every static key counts the candidates below it (its shift in the merged order), and the candidate's own
        // rank is the number of static keys + candidates below it -- three ballots.  Keys are unique (the node index sits in the low
        // word), so the ranks are a permutation and the list comes out exactly as the k rounds of minimum extraction below produce it,
        // at ~ 15 instructions per candidate instead of ~ 40 per round.
        const bool c0 = kl[0] < thr, c1 = kl[1] < thr;
        const unsigned long long m0 = __ballot(c0), m1 = __ballot(c1);
        const int n_static = __popcll(__ballot(lane < TD_K && ks != TD_KEY_MAX));
        int shift = 0;
        bool lig_in = false;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            unsigned long long mm = u ? m1 : m0;
            while (mm) {
                const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)mm) - 1);
                mm &= mm - 1;
                const unsigned clo = __builtin_amdgcn_readlane((int)(unsigned)kl[u], src);
                const unsigned chi = __builtin_amdgcn_readlane((int)(unsigned)(kl[u] >> 32), src);
                const unsigned long long ck = ((unsigned long long)chi << 32) | clo;
                if (lane < TD_K && ck < ks) ++shift;
                const int rank = __popcll(__ballot(lane < TD_K && ks < ck)) + __popcll(__ballot(c0 && kl[0] < ck)) +
                                 __popcll(__ballot(c1 && kl[1] < ck));
                if (rank < k) {
                    lig_in = true;
                    if (lane == 0) nbr[i * TD_K + rank] = (int32_t)clo;
                }
            }
        }
        const int total = n_static + __popcll(m0) + __popcll(m1);
        if (lane < TD_K) {
            const int pos = lane + shift;
            if (ks != TD_KEY_MAX && pos < k) nbr[i * TD_K + pos] = (int32_t)(unsigned)(ks & 0xffffffffull);
            if (lane >= (total < k ? total : k)) nbr[i * TD_K + lane] = -1;
        }
        is_clean = !lig_in;
    } else if (!is_clean) {
        // more than 128 ligand atoms in the graph: k rounds of wave-minimum extraction over {static key, 2 ligand keys} per pass of 128
        unsigned long long best = ks;
        for (int base = lbeg; base < lend; base += 128) {
            if (base != lbeg) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int j = base + lane + 64 * u;
                    kl[u] = TD_KEY_MAX;
                    if (j < lend) {
                        const float4 xj = x4[j];
                        const float d2 = td_dist2(xj.x - xi.x, xj.y - xi.y, xj.z - xi.z);
                        kl[u] = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)j;
                    }
                }
            }
            unsigned long long carry = best, out = TD_KEY_MAX;
            for (int r = 0; r < k; ++r) {
                unsigned long long lmin = carry;
                lmin = kl[0] < lmin ? kl[0] : lmin;
                lmin = kl[1] < lmin ? kl[1] : lmin;
                const unsigned long long wmin = td_wave_min_u64(lmin);
                if (wmin != TD_KEY_MAX) {
                    if (carry == wmin) carry = TD_KEY_MAX;
                    if (kl[0] == wmin) kl[0] = TD_KEY_MAX;
                    if (kl[1] == wmin) kl[1] = TD_KEY_MAX;
                }
                if (lane == r) out = wmin;
            }
            best = out;
        }
        if (lane < TD_K) nbr[i * TD_K + lane] = (best == TD_KEY_MAX || lane >= k) ? -1 : (int32_t)(unsigned)(best & 0xffffffffull);
        // the winners are the k smallest of static U ligand; if none of them is a ligand atom the row is still clean
        const bool lig_in = lane < k && best != TD_KEY_MAX && (int)(unsigned)(best & 0xffffffffull) >= lbeg;
        is_clean = __ballot(lig_in) == 0ull;
    } else if (lane < TD_K) {
        const int32_t sj = snbr[c * TD_K + lane];
        nbr[i * TD_K + lane] = sj >= 0 ? sj + rebase : -1;
    }
    // prepare the step's node state for this row: h = cached layer-0 output (clean) or the embedding h0 (dirty)
    const float2 hv = *reinterpret_cast<const float2 *>((is_clean ? h1s : h0) + c * TD_H + 2 * lane);
    *reinterpret_cast<float2 *>(h + i * TD_H + 2 * lane) = hv;
    if (is_clean && lane < TD_K) ew[i * TD_K + lane] = ews[c * TD_K + lane];
    if (lane == 0) {
        clean[i] = is_clean ? 1 : 0;
        if (flags2) flags2[i] = 0;             // forward-reach flag of this row, set later in the step
    }
}

__global__ __launch_bounds__(256) void knn_merge_kernel(TdMergeArgs ma) { knn_merge_body(blockIdx.x, ma); }

// The two independent halves of a session step's neighbour search in ONE launch: the full k-NN search of the ligand rows (blocks [0, GB):
// a few thousand long, latency-bound waves -- first, so that they run from the start; at the end of the grid they were its tail) and the
// protein rows' merge (blocks [GB, GB + GA)), which fills in beside them.
template <int CH>
__global__ __launch_bounds__(256) void knn_step_kernel(TdMergeArgs ma, unsigned GB, const int32_t *__restrict__ lig_rows, int64_t Nl,
                                                       int32_t *__restrict__ nbr) {
    if (blockIdx.x < GB) knn_body<CH, false>(blockIdx.x, ma.x4, ma.ptr, ma.gid, lig_rows, Nl, ma.k, nbr, nullptr);
    else knn_merge_body(blockIdx.x - GB, ma);
}

int td_launch_knn_merge(const float4 *x4, const int32_t *node_ptr, const int32_t *pptr, const int32_t *gid,
                        const int32_t *prot_rows, int64_t Np, const unsigned long long *skeys, const int32_t *snbr,
                        const float *h0, const float *h1s, const float *ews, int32_t *nbr, float *h, float *ew,
                        uint8_t *clean, uint8_t *flags2, hipStream_t s, int k, const int32_t *lig_rows, int64_t Nl,
                        int max_graph_nodes, const int32_t *cgraph, const int32_t *cbase) {
    const TdMergeArgs ma{x4, node_ptr, pptr, gid, prot_rows, Np, skeys, snbr, h0, h1s, ews, nbr, h, ew, clean, flags2, k, cgraph, cbase};
    const unsigned GA = (unsigned)((Np + 3) / 4), GB = lig_rows ? (unsigned)((Nl + 3) / 4) : 0u;
    if (GA + GB == 0) return TD_OK;
    if (GB == 0) {
        knn_merge_kernel<<<dim3(GA), dim3(256), 0, s>>>(ma);
    } else {           // (the same candidates-per-lane variants as launch_knn_t)
        const dim3 grid(GA + GB), block(256);
        if (max_graph_nodes > 0 && max_graph_nodes <= 256) knn_step_kernel<4><<<grid, block, 0, s>>>(ma, GB, lig_rows, Nl, nbr);
        else if (max_graph_nodes > 0 && max_graph_nodes <= 384) knn_step_kernel<6><<<grid, block, 0, s>>>(ma, GB, lig_rows, Nl, nbr);
        else if (max_graph_nodes <= 704) knn_step_kernel<11><<<grid, block, 0, s>>>(ma, GB, lig_rows, Nl, nbr);       // also the "unknown" (0) default
        else knn_step_kernel<17><<<grid, block, 0, s>>>(ma, GB, lig_rows, Nl, nbr);
    }
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

// The same merge on general graphs (k-NN with 32 < k <= 64, and the protein rows of `hybrid`): 64 sorted static keys per
// protein row (one per lane), neighbour table / gate rows indexed by chunk (cpn chunks of 32 slots per protein row, contiguous
// from chunk cptr[i]).  Clean rows copy their cached chunks; the others are merged with the graph's ligand atoms, k rounds.
__global__ __launch_bounds__(256) void knn_merge_general_kernel(
    const float4 *__restrict__ x4, const int32_t *__restrict__ ptr, const int32_t *__restrict__ pptr,
    const int32_t *__restrict__ gid, const int32_t *__restrict__ prot_rows, int64_t Np,
    const unsigned long long *__restrict__ skeys, const int32_t *__restrict__ snbr, const float *__restrict__ h0,
    const float *__restrict__ h1s, const float *__restrict__ ews, const int32_t *__restrict__ cptr, int cpn,
    int32_t *__restrict__ nbr, float *__restrict__ h, float *__restrict__ ew, uint8_t *__restrict__ clean,
    uint8_t *__restrict__ flags2, int k) {
    const int lane = threadIdx.x & 63;
    const int64_t qi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= Np) return;
    const int64_t i = prot_rows[qi];
    const int g = gid[i];
    const int lbeg = ptr[g] + (pptr[g + 1] - pptr[g]), lend = ptr[g + 1];
    const float4 xi = x4[i];
    const int64_t c0 = cptr[i];
    const unsigned long long ks = skeys[i * 64 + lane];
    const unsigned long long thr = __shfl(ks, k - 1);             // k-th static neighbour (MAX if fewer exist)
    bool closer = false;
    for (int base = lbeg; base < lend; base += 64) {
        const int j = base + lane;
        if (j < lend) {
            const float4 xj = x4[j];
            const float d2 = td_dist2(xj.x - xi.x, xj.y - xi.y, xj.z - xi.z);
            closer |= (((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)j) < thr;
        }
    }
    bool is_clean = __ballot(closer) == 0ull;
    if (!is_clean) {
        unsigned long long best = ks;
        for (int base = lbeg; base < lend; base += 128) {
            unsigned long long kl[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = base + lane + 64 * u;
                kl[u] = TD_KEY_MAX;
                if (j < lend) {
                    const float4 xj = x4[j];
                    const float d2 = td_dist2(xj.x - xi.x, xj.y - xi.y, xj.z - xi.z);
                    kl[u] = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)j;
                }
            }
            unsigned long long carry = best, out = TD_KEY_MAX;
            for (int r = 0; r < k; ++r) {
                unsigned long long lmin = carry;
                lmin = kl[0] < lmin ? kl[0] : lmin;
                lmin = kl[1] < lmin ? kl[1] : lmin;
                const unsigned long long wmin = td_wave_min_u64(lmin);
                if (wmin != TD_KEY_MAX) {
                    if (carry == wmin) carry = TD_KEY_MAX;
                    if (kl[0] == wmin) kl[0] = TD_KEY_MAX;
                    if (kl[1] == wmin) kl[1] = TD_KEY_MAX;
                }
                if (lane == r) out = wmin;
            }
            best = out;
        }
        if (lane < cpn * TD_K) nbr[c0 * TD_K + lane] = (best == TD_KEY_MAX || lane >= k) ? -1 : (int32_t)(unsigned)(best & 0xffffffffull);
        const bool lig_in = lane < k && best != TD_KEY_MAX && (int)(unsigned)(best & 0xffffffffull) >= lbeg;
        is_clean = __ballot(lig_in) == 0ull;
    } else if (lane < cpn * TD_K) {
        nbr[c0 * TD_K + lane] = snbr[c0 * TD_K + lane];
    }
    const float2 hv = *reinterpret_cast<const float2 *>((is_clean ? h1s : h0) + i * TD_H + 2 * lane);
    *reinterpret_cast<float2 *>(h + i * TD_H + 2 * lane) = hv;
    if (is_clean && lane < cpn * TD_K) ew[c0 * TD_K + lane] = ews[c0 * TD_K + lane];
    if (lane == 0) {
        clean[i] = is_clean ? 1 : 0;
        if (flags2) flags2[i] = 0;
    }
}

int td_launch_knn_merge_general(const float4 *x4, const int32_t *node_ptr, const int32_t *pptr, const int32_t *gid,
                                const int32_t *prot_rows, int64_t Np, const unsigned long long *skeys, const int32_t *snbr,
                                const float *h0, const float *h1s, const float *ews, const int32_t *cptr, int cpn, int32_t *nbr,
                                float *h, float *ew, uint8_t *clean, uint8_t *flags2, hipStream_t s, int k) {
    if (Np == 0) return TD_OK;
    knn_merge_general_kernel<<<dim3((unsigned)((Np + 3) / 4)), dim3(256), 0, s>>>(x4, node_ptr, pptr, gid, prot_rows, Np, skeys, snbr,
                                                                                h0, h1s, ews, cptr, cpn, nbr, h, ew, clean, flags2, k);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

// Slots of a block-wide compaction into up to L lists at once: one atomicAdd per block and list (a wave-level ballot per
// list, wave totals through LDS).  flag[l] -> slot[l] = index in list l (valid where flag[l]).  blockDim.x = 256 (4 waves).
// Order inside a list is irrelevant to the arithmetic (rows are independent); the atomics only hand out slots.
template <int L>
__device__ __forceinline__ void td_block_slots(const bool (&flag)[L], int32_t *__restrict__ counters, int (&slot)[L]) {
    __shared__ int s_cnt[L][4];
    __shared__ int s_base[L];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    unsigned long long m[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        m[l] = __ballot(flag[l]);
        if (lane == 0) s_cnt[l][wave] = __popcll(m[l]);
    }
    __syncthreads();
    if (threadIdx.x < L) {
        const int l = threadIdx.x;
        const int total = s_cnt[l][0] + s_cnt[l][1] + s_cnt[l][2] + s_cnt[l][3];
        s_base[l] = total ? atomicAdd(counters + l, total) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int l = 0; l < L; ++l) {
        int before = 0;
        for (int w = 0; w < wave; ++w) before += s_cnt[l][w];
        slot[l] = s_base[l] + before + __popcll(m[l] & below);
    }
}

// dirty rows = ligand rows + protein rows whose neighbour row changed; order is irrelevant (rows are independent).
// `count` was zeroed by the step's first kernel (ligand_update_kernel).
__global__ __launch_bounds__(256) void compact_dirty_kernel(const uint8_t *__restrict__ clean, const float4 *__restrict__ x4,
                                                            int64_t N, int32_t *__restrict__ rows, int32_t *__restrict__ count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool flag[1] = {i < N && (x4[i].w > 0.5f || !clean[i])};
    int slot[1];
    td_block_slots<1>(flag, count, slot);
    if (flag[0]) rows[slot[0]] = (int32_t)i;
}

int td_launch_compact_dirty(const uint8_t *clean, const float4 *x4, int64_t N, int32_t *rows, int32_t *count,
                            hipStream_t s) {
    if (N == 0) return TD_OK;
    compact_dirty_kernel<<<dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s>>>(clean, x4, N, rows, count);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

// Forward reach of the ligand, one hop: flags2[i] = 1 when row i is dirty (a ligand atom, or a protein row with a ligand
// neighbour) or has a dirty neighbour -- the only rows whose layer-1 x2h output differs from the protein-only graph's.
// Rows outside keep the cached static feature row (copied by the same kernel: one wave per row would be wasteful, so the
// copy is a second kernel over the complement list).
__global__ void forward_reach_kernel(const uint8_t *__restrict__ clean, const float4 *__restrict__ x4,
                                     const int32_t *__restrict__ nbr, int64_t N, uint8_t *__restrict__ flags2) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = t >> 5;
    if (i >= N) return;
    const int e = (int)(t & 31);
    const int j = nbr[i * TD_K + e];
    bool d = e == 0 && (x4[i].w > 0.5f || !clean[i]);
    if (j >= 0) d = d || x4[j].w > 0.5f || !clean[j];
    if (d) flags2[i] = 1;                                  // racing writers store the same value
}

// rows with flags != 0 -> rows_on / count[0]; the others -> rows_off / count[1].  Last reader of the step's `clean` flags:
// clears them (`zero_after`) for their second life as receptive-field level flags (td_launch_hop_levels).
__global__ __launch_bounds__(256) void compact_split_kernel(const uint8_t *__restrict__ flags, int64_t N,
                                                            int32_t *__restrict__ rows_on, int32_t *__restrict__ rows_off,
                                                            int32_t *__restrict__ count, uint8_t *__restrict__ zero_after) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < N, on = in && flags[i];
    const bool flag[2] = {on, in && !on};
    int slot[2];
    td_block_slots<2>(flag, count, slot);
    if (on) rows_on[slot[0]] = (int32_t)i;
    else if (in) rows_off[slot[1]] = (int32_t)i;
    if (in && zero_after) zero_after[i] = 0;
}

// flags2 and counts2 were zeroed earlier in the step (knn_merge_kernel / ligand_update_kernel); `clean` is cleared here.
int td_launch_forward_reach(const uint8_t *clean, const float4 *x4, const int32_t *nbr, int64_t N, uint8_t *flags2,
                            int32_t *rows_on, int32_t *rows_off, int32_t *counts2, uint8_t *clean_to_zero, hipStream_t s) {
    if (N == 0) return TD_OK;
    forward_reach_kernel<<<dim3((unsigned)((N * 32 + 255) / 256)), dim3(256), 0, s>>>(clean, x4, nbr, N, flags2);
    TD_CHECK_HIP(hipGetLastError());
    compact_split_kernel<<<dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s>>>(flags2, N, rows_on, rows_off, counts2, clean_to_zero);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

// h[i] = hs[i] for the listed rows (device-side count).  cgraph / cbase (see TdMergeArgs): hs is a compact table shared by the replicas of
// a pocket -- the listed rows are protein rows (a ligand row is always inside the forward reach), row i of graph g reads cbase[g] + (i - ptr[g])
__global__ void restore_rows_kernel(const int32_t *__restrict__ rows, const int32_t *__restrict__ count_ptr,
                                    const float4 *__restrict__ hs, float4 *__restrict__ h, const int32_t *__restrict__ gid,
                                    const int32_t *__restrict__ ptr, const int32_t *__restrict__ cbase) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t a = t >> 5;                 // 32 float4 per row
    if (a >= *count_ptr) return;
    const int64_t i = rows[a];
    int64_t c = i;
    if (cbase) {
        const int g = gid[i];
        c = (int64_t)cbase[g] + (i - ptr[g]);
    }
    h[i * 32 + (t & 31)] = hs[c * 32 + (t & 31)];
}

int td_launch_restore_rows(const int32_t *rows, const int32_t *count_ptr, int64_t max_rows, const float *hs, float *h,
                           hipStream_t s, const int32_t *gid, const int32_t *node_ptr, const int32_t *cbase) {
    if (max_rows == 0) return TD_OK;
    restore_rows_kernel<<<dim3((unsigned)((max_rows * 32 + 255) / 256)), dim3(256), 0, s>>>(
        rows, count_ptr, reinterpret_cast<const float4 *>(hs), reinterpret_cast<float4 *>(h), gid, node_ptr, cbase);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

// ---- per-pocket sharing of a session's static tables: which graphs of the batch carry the same protein block --------------------------------
// One workgroup per graph: a 64-bit hash (FNV-1a over the words, folded per lane and combined in lane order) of its protein positions and
// feature rows.  The host groups graphs by (atom count, hash) and has pocket_verify_kernel compare every candidate with its group's first
// graph word by word, so a hash collision costs a graph its sharing, never its results.
__global__ __launch_bounds__(256) void pocket_hash_kernel(const float *__restrict__ ppos, const float *__restrict__ pv, const int32_t *__restrict__ pptr,
                                                          int F, unsigned long long *__restrict__ out) {
    __shared__ unsigned long long part[256];
    const int g = blockIdx.x, p0 = pptr[g], np = pptr[g + 1] - p0;
    const unsigned *a = reinterpret_cast<const unsigned *>(ppos) + (size_t)p0 * 3, *b = reinterpret_cast<const unsigned *>(pv) + (size_t)p0 * F;
    unsigned long long hsh = 1469598103934665603ull;
    for (int64_t t = threadIdx.x; t < (int64_t)np * 3; t += 256) hsh = (hsh ^ a[t]) * 1099511628211ull;
    for (int64_t t = threadIdx.x; t < (int64_t)np * F; t += 256) hsh = (hsh ^ b[t]) * 1099511628211ull;
    part[threadIdx.x] = hsh;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long acc = 1469598103934665603ull;
        for (int t = 0; t < 256; ++t) acc = (acc ^ part[t]) * 1099511628211ull;
        out[g] = acc;
    }
}
// flags[g] = 1 unless graph g's protein block equals graph cand[g]'s bit for bit (cand[g] == g: trivially equal)
__global__ __launch_bounds__(256) void pocket_verify_kernel(const float *__restrict__ ppos, const float *__restrict__ pv, const int32_t *__restrict__ pptr,
                                                            int F, const int32_t *__restrict__ cand, int32_t *__restrict__ flags) {
    const int g = blockIdx.x, c = cand[g];
    if (c == g) return;
    const int p0 = pptr[g], q0 = pptr[c], np = pptr[g + 1] - p0;
    bool diff = (pptr[c + 1] - q0) != np;
    if (!diff) {
        const unsigned *a = reinterpret_cast<const unsigned *>(ppos), *b = reinterpret_cast<const unsigned *>(pv);
        for (int64_t t = threadIdx.x; t < (int64_t)np * 3 && !diff; t += 256) diff = a[(size_t)p0 * 3 + t] != a[(size_t)q0 * 3 + t];
        for (int64_t t = threadIdx.x; t < (int64_t)np * F && !diff; t += 256) diff = b[(size_t)p0 * F + t] != b[(size_t)q0 * F + t];
    }
    if (diff) flags[g] = 1;
}
int td_launch_pocket_hash(const float *ppos, const float *pv, const int32_t *pptr, int64_t B, int F, unsigned long long *out, hipStream_t s) {
    pocket_hash_kernel<<<dim3((unsigned)B), dim3(256), 0, s>>>(ppos, pv, pptr, F, out);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}
int td_launch_pocket_verify(const float *ppos, const float *pv, const int32_t *pptr, int64_t B, int F, const int32_t *cand, int32_t *flags,
                            hipStream_t s) {
    pocket_verify_kernel<<<dim3((unsigned)B), dim3(256), 0, s>>>(ppos, pv, pptr, F, cand, flags);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}
// dst[c] = src[rows[c]] for rows of `row16` 16-byte words (the canonical rows of full-size set-up tables into the compact shared ones)
__global__ void compact_rows_kernel(const uint4 *__restrict__ src, const int32_t *__restrict__ rows, int64_t n_rows, int row16,
                                    uint4 *__restrict__ dst) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t c = t / row16;
    if (c >= n_rows) return;
    const int w = (int)(t - c * row16);
    dst[c * row16 + w] = src[(int64_t)rows[c] * row16 + w];
}
int td_launch_compact_rows(const void *src, const int32_t *rows, int64_t n_rows, int row_bytes, void *dst, hipStream_t s) {
    if (n_rows == 0) return TD_OK;
    const int row16 = row_bytes / 16;
    const int64_t total = n_rows * row16;
    compact_rows_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(reinterpret_cast<const uint4 *>(src), rows, n_rows, row16,
                                                                                   reinterpret_cast<uint4 *>(dst));
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

// Receptive field of the ligand outputs, one level per layer counted from the last: level 1 = the ligand atoms and
// their neighbours (the last h2x reads the neighbours' projections), level k + 1 = level k plus its neighbours.  When
// only ligand outputs are consumed, the layer e from the end needs new features for level e + 1 only.
// flags[i] = first level that reaches node i (0 = not reached yet).
__global__ void expand_hop_kernel(const int32_t *__restrict__ rows, int64_t count, const int32_t *__restrict__ count_ptr,
                                  const int32_t *__restrict__ nbr, int level, uint8_t *__restrict__ flags) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t a = t >> 5;
    if (count_ptr) count = *count_ptr;
    if (a >= count) return;
    const int i = rows[a];
    const int e = (int)(t & 31);
    if (e == 0 && flags[i] == 0) flags[i] = (uint8_t)level;
    const int j = nbr[(int64_t)i * TD_K + e];
    if (j >= 0 && flags[j] == 0) flags[j] = (uint8_t)level;     // racing writers store the same value
}

// level k (k >= 2): every row reached at a level < k marks its still unreached neighbours with k.  Runs over all rows, so
// the levels need no compacted row lists in between.
__global__ void expand_level_kernel(const int32_t *__restrict__ nbr, int64_t N, int level, uint8_t *__restrict__ flags) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = t >> 5;
    if (i >= N) return;
    const int f = flags[i];
    if (f == 0 || f >= level) return;                      // rows marked in this very pass (f == level) do not propagate yet
    const int j = nbr[i * TD_K + (t & 31)];
    if (j >= 0 && flags[j] == 0) flags[j] = (uint8_t)level;
}

// all level lists in one pass: level k = rows with 0 < flags <= k
template <int LEVELS>
__global__ __launch_bounds__(256) void compact_levels_kernel(const uint8_t *__restrict__ flags, int64_t N, int32_t *__restrict__ rows,
                                                             int32_t *__restrict__ counts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int f = i < N ? (int)flags[i] : 0;
    bool flag[LEVELS];
#pragma unroll
    for (int k = 0; k < LEVELS; ++k) flag[k] = f != 0 && f <= k + 1;
    int slot[LEVELS];
    td_block_slots<LEVELS>(flag, counts, slot);
#pragma unroll
    for (int k = 0; k < LEVELS; ++k)
        if (flag[k]) rows[(size_t)k * N + slot[k]] = (int32_t)i;
}

// rows: [levels][N] row lists, counts: [levels] device-side lengths
// `zeroed`: flags and counts were already cleared earlier in the step (compact_split_kernel / ligand_update_kernel)
int td_launch_hop_levels(const int32_t *lig_node, int64_t Nl, const int32_t *nbr, int64_t N, uint8_t *flags,
                         int32_t *rows, int32_t *counts, int levels, hipStream_t s, bool zeroed) {
    if (N == 0 || levels <= 0) return TD_OK;
    if (levels > TD_HOP_LEVELS) levels = TD_HOP_LEVELS;
    if (!zeroed) {
        TD_CHECK_HIP(hipMemsetAsync(flags, 0, (size_t)N, s));
        TD_CHECK_HIP(hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)TD_HOP_LEVELS, s));
    }
    if (Nl > 0) expand_hop_kernel<<<dim3((unsigned)((Nl * 32 + 255) / 256)), dim3(256), 0, s>>>(lig_node, Nl, nullptr, nbr, 1, flags);
    TD_CHECK_HIP(hipGetLastError());
    for (int k = 2; k <= levels; ++k) {
        expand_level_kernel<<<dim3((unsigned)((N * 32 + 255) / 256)), dim3(256), 0, s>>>(nbr, N, k, flags);
        TD_CHECK_HIP(hipGetLastError());
    }
    const dim3 grid((unsigned)((N + 255) / 256));
    switch (levels) {
        case 1: compact_levels_kernel<1><<<grid, dim3(256), 0, s>>>(flags, N, rows, counts); break;
        case 2: compact_levels_kernel<2><<<grid, dim3(256), 0, s>>>(flags, N, rows, counts); break;
        case 3: compact_levels_kernel<3><<<grid, dim3(256), 0, s>>>(flags, N, rows, counts); break;
        default: compact_levels_kernel<4><<<grid, dim3(256), 0, s>>>(flags, N, rows, counts); break;
    }
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

// ---- every row list of a sampling step in one launch ---------------------------------------------------------------------
// dirty rows (compact_dirty_kernel), the ligand's one-hop forward reach and its complement (forward_reach_kernel +
// compact_split_kernel), the receptive-field levels (expand_hop / expand_level / compact_levels): all of them follow from the
// step's neighbour table and the clean flags, and edges never leave a graph -- so one workgroup per graph keeps the three
// per-node flag bytes in LDS, walks its rows level by level with __syncthreads() in between (16 waves, four neighbour loads in
// flight per thread), and hands out list slots with one atomicAdd per 1024-node tile and list.  Same sets as the separate kernels (the order inside a list is irrelevant: rows
// are independent); 8 launches of a step become one, which is what small batches feel.
constexpr int TD_STEP_LISTS = 3 + TD_HOP_LEVELS;        // dirty, reach, rest, levels 1 .. 4

template <int L, int WAVES>
__device__ __forceinline__ void td_block_slots_each(const bool (&flag)[L], int32_t *const (&counter)[L], int (&slot)[L]) {
    __shared__ int s_cnt[L][WAVES];
    __shared__ int s_base[L];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    unsigned long long m[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        m[l] = __ballot(flag[l]);
        if (lane == 0) s_cnt[l][wave] = __popcll(m[l]);
    }
    __syncthreads();
    if (threadIdx.x < L) {
        const int l = threadIdx.x;
        int total = 0;
        for (int w = 0; w < WAVES; ++w) total += s_cnt[l][w];
        int32_t *c = counter[0];
#pragma unroll
        for (int q = 1; q < L; ++q) c = l == q ? counter[q] : c;
        s_base[l] = (total && c) ? atomicAdd(c, total) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int l = 0; l < L; ++l) {
        int before = 0;
        for (int w = 0; w < wave; ++w) before += s_cnt[l][w];
        slot[l] = s_base[l] + before + __popcll(m[l] & below);
    }
    __syncthreads();          // s_cnt / s_base are reused by the next tile
}

constexpr int TD_LISTS_THREADS = 1024;

// CHUNKED (general graphs): the neighbour table is chunk-indexed -- the graph's chunks are cptr[n0] .. cptr[n0 + n) - 1, chunk c
// belongs to dst node chunk_node[c] -- and one more list comes out: the chunks of the dirty rows (what the edge gate recomputes).
template <bool CHUNKED>
__global__ __launch_bounds__(TD_LISTS_THREADS) void step_lists_kernel(const uint8_t *__restrict__ clean, const float4 *__restrict__ x4,
                                                                      const int32_t *__restrict__ nbr, const int32_t *__restrict__ node_ptr,
                                                                      const int32_t *__restrict__ cptr, const int32_t *__restrict__ chunk_node,
                                                                      int64_t N, int cap, TdStepLists out) {
    constexpr int T = TD_LISTS_THREADS, RPT = T / TD_K;      // rows per trip: thread = (row a0 + tid / 32, slot tid % 32)
    extern __shared__ uint8_t s_flags[];
    uint8_t *dirty = s_flags, *reach = s_flags + cap, *level = s_flags + 2 * cap, *lig = s_flags + 3 * cap;
    const int n0 = node_ptr[blockIdx.x], n = node_ptr[blockIdx.x + 1] - n0;
    const int tid = threadIdx.x, e = tid & 31, r0 = tid >> 5;
    // "rows" of the table: the graph's nodes, or (CHUNKED) its chunks
    const int t0 = CHUNKED ? cptr[n0] : n0, nt = CHUNKED ? cptr[n0 + n] - t0 : n;
    for (int a = tid; a < n; a += T) {
        const bool l = x4[n0 + a].w > 0.5f;
        const bool d = l || !clean[n0 + a];
        dirty[a] = d;
        reach[a] = d;
        lig[a] = l;
        level[a] = l ? 1 : 0;                           // level 1 starts from the ligand atoms
    }
    __syncthreads();
    const int32_t *row0 = nbr + (int64_t)t0 * TD_K + e;
    auto dst_of = [&](int t) -> int { return CHUNKED ? chunk_node[t0 + t] - n0 : t; };
    // forward reach = rows with a dirty in-neighbour; level 1 = the ligand rows' in-neighbours
    for (int a0 = r0; a0 < nt; a0 += 4 * RPT) {
        int j[4], dn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = a0 + u * RPT;
            j[u] = t < nt ? row0[(int64_t)t * TD_K] : -1;
            dn[u] = t < nt ? dst_of(t) : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (j[u] < 0) continue;
            if (dirty[j[u] - n0]) reach[dn[u]] = 1;                           // racing writers store the same value
            if (lig[dn[u]] && level[j[u] - n0] == 0) level[j[u] - n0] = 1;
        }
    }
    __syncthreads();
    for (int k = 2; k <= out.levels; ++k) {
        // rows reached at level k - 1 mark their still unreached in-neighbours with k (earlier levels did so in earlier passes)
        for (int a0 = r0; a0 < nt; a0 += 4 * RPT) {
            int j[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = a0 + u * RPT;
                j[u] = (t < nt && level[dst_of(t)] == k - 1) ? row0[(int64_t)t * TD_K] : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (j[u] >= 0 && level[j[u] - n0] == 0) level[j[u] - n0] = (uint8_t)k;
        }
        __syncthreads();
    }
    int32_t *const counters[TD_STEP_LISTS] = {out.dirty_count,
                                              out.reach_rows ? out.reach_counts : nullptr, out.reach_rows ? out.reach_counts + 1 : nullptr,
                                              out.level_counts, out.level_counts + 1, out.level_counts + 2, out.level_counts + 3};
    for (int base = 0; base < n; base += T) {         // uniform trip count: every thread reaches the barriers inside
        const int a = base + tid;
        const bool in = a < n;
        const int lv = in ? level[a] : 0;
        bool flag[TD_STEP_LISTS];
        flag[0] = in && dirty[a];
        flag[1] = in && out.reach_rows && reach[a];
        flag[2] = in && out.reach_rows && !reach[a];
#pragma unroll
        for (int k = 0; k < TD_HOP_LEVELS; ++k) flag[3 + k] = k < out.levels && lv != 0 && lv <= k + 1;
        int slot[TD_STEP_LISTS];
        td_block_slots_each<TD_STEP_LISTS, T / 64>(flag, counters, slot);
        const int32_t i = n0 + a;
        if (flag[0]) out.dirty_rows[slot[0]] = i;
        if (flag[1]) out.reach_rows[slot[1]] = i;
        if (flag[2]) out.rest_rows[slot[2]] = i;
#pragma unroll
        for (int k = 0; k < TD_HOP_LEVELS; ++k)
            if (flag[3 + k]) out.level_rows[(size_t)k * N + slot[3 + k]] = i;
    }
    if (CHUNKED && out.dirty_chunks) {
        int32_t *const cc[1] = {out.dirty_chunk_count};
        for (int base = 0; base < nt; base += T) {
            const int t = base + tid;
            const bool flag[1] = {t < nt && dirty[dst_of(t < nt ? t : 0)]};
            int slot[1];
            td_block_slots_each<1, T / 64>(flag, cc, slot);
            if (flag[0]) out.dirty_chunks[slot[0]] = t0 + t;
        }
    }
}

// `max_nodes`: an upper bound on the nodes of one graph (exact, not a hint: it sizes the LDS flag arrays).  Counters are zeroed
// by the step's first kernel.  Returns TD_EINVAL when a graph is too large for LDS (the caller then uses the separate kernels).
int td_launch_step_lists(const uint8_t *clean, const float4 *x4, const int32_t *nbr, const int32_t *node_ptr, int64_t N,
                         int64_t B, int max_nodes, const TdStepLists &out, hipStream_t s, const int32_t *cptr,
                         const int32_t *chunk_node) {
    if (N == 0 || B == 0) return TD_OK;
    const int cap = (max_nodes + 15) & ~15;
    const size_t bytes = (size_t)4 * cap;
    if (bytes > 48 * 1024 || out.levels > TD_HOP_LEVELS) return TD_EINVAL;
    if (cptr)
        step_lists_kernel<true><<<dim3((unsigned)B), dim3(TD_LISTS_THREADS), bytes, s>>>(clean, x4, nbr, node_ptr, cptr, chunk_node, N, cap, out);
    else
        step_lists_kernel<false><<<dim3((unsigned)B), dim3(TD_LISTS_THREADS), bytes, s>>>(clean, x4, nbr, node_ptr, nullptr, nullptr, N, cap, out);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

// per-step refresh of the ligand rows: x4 = (pos, 1), h = Linear(one_hot(v)) ; 1
// First kernel of a session step, so it also resets the step's bookkeeping (instead of a handful of memset launches): the
// device-side counters of the row lists and the forward-reach flags of the ligand rows (knn_merge_kernel clears the
// protein rows').
__global__ __launch_bounds__(128) void ligand_update_kernel(const float *__restrict__ lpos, const int64_t *__restrict__ lv,
                                                            const int32_t *__restrict__ lig_node, int64_t Nl, int C,
                                                            const float *__restrict__ WlT, const float *__restrict__ bl,
                                                            float *__restrict__ h, float4 *__restrict__ x4, TdStepReset rs,
                                                            const float *__restrict__ gbias, const int32_t *__restrict__ gid) {
    const int n = threadIdx.x;
    if (blockIdx.x == 0) {
        if (rs.c0 && n < rs.n0) rs.c0[n] = 0;
        if (rs.c1 && n < rs.n1) rs.c1[n] = 0;
        if (rs.c2 && n < rs.n2) rs.c2[n] = 0;
    }
    const int64_t a0 = (int64_t)blockIdx.x * TD_COMPOSE_ATOMS;
    const float bias = bl[n];
    for (int a = 0; a < TD_COMPOSE_ATOMS; ++a) {
        const int64_t at = a0 + a;
        if (at >= Nl) break;
        const int64_t p = lig_node[at];
        int v = (int)lv[at];
        v = v < 0 ? 0 : (v >= C ? C - 1 : v);
        // gbias (optional): the time-embedding columns of ligand_atom_emb applied to the graph's time feature
        // (models/molopt_score_model.py:319-329), one row per graph
        h[p * TD_H + n] = (gbias ? WlT[v * TD_H + n] + gbias[(size_t)gid[p] * TD_H + n] : WlT[v * TD_H + n]) + bias;
        if (n == 0) {
            x4[p] = make_float4(lpos[3 * at], lpos[3 * at + 1], lpos[3 * at + 2], 1.f);
            if (rs.flags2) rs.flags2[p] = 0;
        }
    }
}

// ------------------------------------------------------------------------------------------ compose
// h0 = [Linear(protein_v) ; 0] / [Linear(one_hot(ligand_v)) ; 1], written straight into the packed
// (graph-major, protein-then-ligand) node order that compose_context's stable sort produces.

__global__ __launch_bounds__(128) void compose_protein_kernel(
    const float *__restrict__ ppos, const float *__restrict__ pv, const int32_t *__restrict__ pptr,
    const int32_t *__restrict__ lptr, int64_t Np, int B, int F, const float *__restrict__ WpT,
    const float *__restrict__ bp, float *__restrict__ h, float4 *__restrict__ x4, int32_t *__restrict__ gid,
    int32_t *__restrict__ prot_node) {
    __shared__ float s_v[TD_COMPOSE_ATOMS][32];
    const int n = threadIdx.x;
    const int64_t a0 = (int64_t)blockIdx.x * TD_COMPOSE_ATOMS;
    for (int idx = n; idx < TD_COMPOSE_ATOMS * 32; idx += 128) {
        int a = idx >> 5, c = idx & 31;
        s_v[a][c] = (a0 + a < Np && c < F) ? pv[(a0 + a) * F + c] : 0.f;
    }
    __syncthreads();
    float w[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) w[c] = (c < F) ? WpT[c * TD_H + n] : 0.f;
    const float bias = bp[n];
    for (int a = 0; a < TD_COMPOSE_ATOMS; ++a) {
        int64_t at = a0 + a;
        if (at >= Np) break;
        int g = td_find_graph(pptr, B, (int)at);
        int64_t p = at + lptr[g];
        float acc = bias;
#pragma unroll
        for (int c = 0; c < 32; ++c) acc = fmaf(w[c], s_v[a][c], acc);
        h[p * TD_H + n] = acc;
        if (n == 0) {
            x4[p] = make_float4(ppos[3 * at], ppos[3 * at + 1], ppos[3 * at + 2], 0.f);
            gid[p] = g;
            if (prot_node) prot_node[at] = (int32_t)p;
        }
    }
}

__global__ __launch_bounds__(128) void compose_ligand_kernel(
    const float *__restrict__ lpos, const int64_t *__restrict__ lv, const int32_t *__restrict__ pptr,
    const int32_t *__restrict__ lptr, int64_t Nl, int B, int C, const float *__restrict__ WlT,
    const float *__restrict__ bl, float *__restrict__ h, float4 *__restrict__ x4, int32_t *__restrict__ gid,
    int32_t *__restrict__ lig_node, const float *__restrict__ gbias) {
    const int n = threadIdx.x;
    const int64_t a0 = (int64_t)blockIdx.x * TD_COMPOSE_ATOMS;
    const float bias = bl[n];
    for (int a = 0; a < TD_COMPOSE_ATOMS; ++a) {
        int64_t at = a0 + a;
        if (at >= Nl) break;
        int g = td_find_graph(lptr, B, (int)at);
        int64_t p = (int64_t)pptr[g + 1] + at;
        int v = (int)lv[at];
        v = v < 0 ? 0 : (v >= C ? C - 1 : v);
        h[p * TD_H + n] = (gbias ? WlT[v * TD_H + n] + gbias[(size_t)g * TD_H + n] : WlT[v * TD_H + n]) + bias;
        if (n == 0) {
            x4[p] = make_float4(lpos[3 * at], lpos[3 * at + 1], lpos[3 * at + 2], 1.f);
            gid[p] = g;
            lig_node[at] = (int32_t)p;
        }
    }
}

__global__ void node_ptr_kernel(const int32_t *__restrict__ pptr, const int32_t *__restrict__ lptr, int B,
                                int32_t *__restrict__ node_ptr) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g <= B) node_ptr[g] = pptr[g] + lptr[g];
}

int td_launch_compose(const td_model *m, const float *ppos, const float *pv, const int32_t *pptr, int64_t Np,
                      const float *lpos, const int64_t *lv, const int32_t *lptr, int64_t Nl, int64_t B,
                      float *h, float4 *x4, int32_t *node_ptr, int32_t *gid, int32_t *lig_node, int32_t *prot_node,
                      hipStream_t s, const float *gbias) {
    node_ptr_kernel<<<dim3((unsigned)((B + 1 + 255) / 256)), dim3(256), 0, s>>>(pptr, lptr, (int)B, node_ptr);
    TD_CHECK_HIP(hipGetLastError());
    if (Np > 0) {
        unsigned nb = (unsigned)((Np + TD_COMPOSE_ATOMS - 1) / TD_COMPOSE_ATOMS);
        compose_protein_kernel<<<dim3(nb), dim3(128), 0, s>>>(ppos, pv, pptr, lptr, Np, (int)B,
                                                             m->cfg.protein_feat_dim, m->emb.WpT, m->emb.bp, h, x4,
                                                             gid, prot_node);
        TD_CHECK_HIP(hipGetLastError());
    }
    if (Nl > 0) {
        unsigned nb = (unsigned)((Nl + TD_COMPOSE_ATOMS - 1) / TD_COMPOSE_ATOMS);
        compose_ligand_kernel<<<dim3(nb), dim3(128), 0, s>>>(lpos, lv, pptr, lptr, Nl, (int)B,
                                                            m->cfg.ligand_num_classes, m->emb.WlT, m->emb.bl, h, x4,
                                                            gid, lig_node, gbias);
        TD_CHECK_HIP(hipGetLastError());
    }
    return TD_OK;
}

int td_launch_ligand_update(const td_model *m, const float *lpos, const int64_t *lv, const int32_t *lig_node, int64_t Nl,
                            float *h, float4 *x4, hipStream_t s, const TdStepReset *reset, const float *gbias,
                            const int32_t *gid) {
    if (Nl == 0) return TD_OK;
    unsigned nb = (unsigned)((Nl + TD_COMPOSE_ATOMS - 1) / TD_COMPOSE_ATOMS);
    const TdStepReset rs = reset ? *reset : TdStepReset{};
    ligand_update_kernel<<<dim3(nb), dim3(128), 0, s>>>(lpos, lv, lig_node, Nl, m->cfg.ligand_num_classes, m->emb.WlT,
                                                      m->emb.bl, h, x4, rs, gbias, gid);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

// ================================================================================================ general graphs
// Chunked neighbour table for graphs whose rows are not exactly 32 wide (k-NN with k != 32, `hybrid`, radius with a
// fan-out cap; models/uni_transformer.py:276-286): the in-edges of node i occupy the chunks cptr[i] .. cptr[i+1]-1 of 32
// slots each (-1 padded), i.e. a flat slot array of length 32 * chunks starting at slot 32 * cptr[i].  A chunk is what one
// wave of the edge kernels processes (one 32-row MFMA tile); chunk_node[c] is its dst node.  The default graph (k = 32)
// is the special case chunk == node.
__global__ void layout_kernel(const int32_t *__restrict__ node_ptr, const int32_t *__restrict__ pptr,
                              const int32_t *__restrict__ gid, const int32_t *__restrict__ g_cbase,
                              const int32_t *__restrict__ g_cl, const int32_t *__restrict__ g_lbase, int cpn_p, int64_t N,
                              int32_t *__restrict__ cptr, int32_t *__restrict__ chunk_node, int32_t *__restrict__ lig_chunks,
                              int32_t total_chunks) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > N) return;
    if (i == N) { cptr[N] = total_chunks; return; }
    const int g = gid[i];
    const int li = (int)i - node_ptr[g], np = pptr[g + 1] - pptr[g];
    int c0, cnt;
    if (li < np) { c0 = g_cbase[g] + li * cpn_p; cnt = cpn_p; }
    else { c0 = g_cbase[g] + np * cpn_p + (li - np) * g_cl[g]; cnt = g_cl[g]; }
    cptr[i] = c0;
    for (int t = 0; t < cnt; ++t) chunk_node[c0 + t] = (int32_t)i;
    if (li >= np)
        for (int t = 0; t < cnt; ++t) lig_chunks[g_lbase[g] + (li - np) * cnt + t] = c0 + t;
}

int td_launch_layout(const int32_t *node_ptr, const int32_t *pptr, const int32_t *gid, const int32_t *g_cbase,
                     const int32_t *g_cl, const int32_t *g_lbase, int cpn_p, int64_t N, int32_t *cptr, int32_t *chunk_node,
                     int32_t *lig_chunks, int32_t total_chunks, hipStream_t s) {
    layout_kernel<<<dim3((unsigned)((N + 1 + 255) / 256)), dim3(256), 0, s>>>(node_ptr, pptr, gid, g_cbase, g_cl, g_lbase, cpn_p, N,
                                                                             cptr, chunk_node, lig_chunks, total_chunks);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

// k nearest same-graph nodes for any k <= 64, written into the node's slot array (ascending (d2, index), like knn_kernel).
// MODE 1 (PROT_ONLY): candidates are the protein atoms of the graph only and the winners go to slots slot0 .. slot0 + k - 1
//         where slot0 = (number of ligand atoms of the graph) - 1: the protein half of a hybrid ligand row.
// MODE 2 (STATIC): the protein-only lists of a sampling session: ligand candidates are skipped, all 64 rounds run and the
//         sorted keys are kept (skeys[i][64]) so that later steps only merge the graph's ligand atoms in.
template <int CH, int MODE>
__global__ __launch_bounds__(256) void knn_general_kernel(const float4 *__restrict__ x4, const int32_t *__restrict__ ptr,
                                                          const int32_t *__restrict__ pptr, const int32_t *__restrict__ gid,
                                                          const int32_t *__restrict__ rows, int64_t count, int k,
                                                          const int32_t *__restrict__ cptr, int32_t *__restrict__ cnbr,
                                                          unsigned long long *__restrict__ skeys) {
    constexpr bool PROT_ONLY = MODE == 1, STATIC = MODE == 2;
    const int lane = threadIdx.x & 63;
    const int64_t qi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= count) return;
    const int64_t i = rows ? (int64_t)rows[qi] : qi;
    const int g = gid[i];
    const int beg = ptr[g];
    const int np = pptr[g + 1] - pptr[g];
    const int end = (PROT_ONLY || STATIC) ? beg + np : ptr[g + 1];
    const int slot0 = PROT_ONLY ? (ptr[g + 1] - beg - np) - 1 : 0;
    const int rounds = STATIC ? 64 : k;
    const float4 xi = x4[i];
    unsigned long long best = TD_KEY_MAX;    // lanes 0..rounds-1: current r-th smallest key
    for (int base = beg; base < end; base += 64 * CH) {
        unsigned long long key[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int j = base + lane + 64 * u;
            key[u] = TD_KEY_MAX;
            if (j < end && j != (int)i) {
                const float4 xj = x4[j];
                const float d2 = td_dist2(xj.x - xi.x, xj.y - xi.y, xj.z - xi.z);
                key[u] = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)j;
            }
        }
        unsigned long long carry = best, out = TD_KEY_MAX;
        for (int r = 0; r < rounds; ++r) {
            unsigned long long lmin = carry;
#pragma unroll
            for (int u = 0; u < CH; ++u) lmin = key[u] < lmin ? key[u] : lmin;
            const unsigned long long wmin = td_wave_min_u64(lmin);
            if (wmin != TD_KEY_MAX) {
                if (carry == wmin) carry = TD_KEY_MAX;
#pragma unroll
                for (int u = 0; u < CH; ++u)
                    if (key[u] == wmin) key[u] = TD_KEY_MAX;
            }
            if (lane == r) out = wmin;
        }
        best = out;
    }
    if (lane < k && best != TD_KEY_MAX) cnbr[(int64_t)cptr[i] * TD_K + slot0 + lane] = (int32_t)(unsigned)(best & 0xffffffffull);
    if (STATIC) skeys[i * 64 + lane] = best;
}

// ligand half of a hybrid ligand row (models/common.py:166-171): every other ligand atom of the graph, ascending index
__global__ void hybrid_ligand_kernel(const int32_t *__restrict__ ptr, const int32_t *__restrict__ pptr,
                                     const int32_t *__restrict__ gid, const int32_t *__restrict__ lig_node, int64_t Nl,
                                     const int32_t *__restrict__ cptr, int32_t *__restrict__ cnbr) {
    const int lane = threadIdx.x & 63;
    const int64_t qi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= Nl) return;
    const int i = lig_node[qi];
    const int g = gid[i];
    const int lbeg = ptr[g] + (pptr[g + 1] - pptr[g]), lend = ptr[g + 1];
    int32_t *row = cnbr + (int64_t)cptr[i] * TD_K;
    for (int s = lane; s < lend - lbeg - 1; s += 64) {
        const int j = lbeg + s;
        row[s] = j >= i ? j + 1 : j;
    }
}

int td_launch_hybrid_ligand_half(const int32_t *node_ptr, const int32_t *pptr, const int32_t *gid, const int32_t *lig_node,
                                 int64_t Nl, const int32_t *cptr, int32_t *cnbr, hipStream_t s) {
    if (Nl == 0) return TD_OK;
    hybrid_ligand_kernel<<<dim3((unsigned)((Nl + 3) / 4)), dim3(256), 0, s>>>(node_ptr, pptr, gid, lig_node, Nl, cptr, cnbr);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

// radius graph with a fan-out cap (rule: oracle/shims.py radius_neighbours): the first `cap` nodes j != i of the same
// graph, in index order, with d2 = (dx*dx + dy*dy) + dz*dz < r*r (fp32, strict)
__global__ __launch_bounds__(256) void radius_kernel(const float4 *__restrict__ x4, const int32_t *__restrict__ ptr,
                                                     const int32_t *__restrict__ gid, int64_t N, float r2, int cap,
                                                     const int32_t *__restrict__ cptr, int32_t *__restrict__ cnbr) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    const int g = gid[i];
    const int beg = ptr[g], end = ptr[g + 1];
    const float4 xi = x4[i];
    int32_t *row = cnbr + (cptr ? (int64_t)cptr[i] : i) * TD_K;       // cptr == nullptr: one 32-slot row per node (cap <= 32)
    int cnt = 0;
    for (int base = beg; base < end && cnt < cap; base += 64) {
        const int j = base + lane;
        bool hit = false;
        if (j < end && j != (int)i) {
            const float4 xj = x4[j];
            hit = td_dist2(xj.x - xi.x, xj.y - xi.y, xj.z - xi.z) < r2;
        }
        const unsigned long long m = __ballot(hit);
        const int rank = cnt + __popcll(m & ((1ull << lane) - 1ull));
        if (hit && rank < cap) row[rank] = j;
        cnt += __popcll(m);
    }
}

template <int MODE>
static int launch_knn_general(const float4 *x4, const int32_t *node_ptr, const int32_t *pptr, const int32_t *gid,
                              const int32_t *rows, int64_t count, int k, int max_graph_nodes, const int32_t *cptr,
                              int32_t *cnbr, hipStream_t s, unsigned long long *skeys = nullptr) {
    if (count == 0) return TD_OK;
    dim3 grid((unsigned)((count + 3) / 4)), block(256);
    if (max_graph_nodes > 0 && max_graph_nodes <= 384)
        knn_general_kernel<6, MODE><<<grid, block, 0, s>>>(x4, node_ptr, pptr, gid, rows, count, k, cptr, cnbr, skeys);
    else if (max_graph_nodes <= 704)
        knn_general_kernel<11, MODE><<<grid, block, 0, s>>>(x4, node_ptr, pptr, gid, rows, count, k, cptr, cnbr, skeys);
    else
        knn_general_kernel<17, MODE><<<grid, block, 0, s>>>(x4, node_ptr, pptr, gid, rows, count, k, cptr, cnbr, skeys);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

// Session of a general k-NN / hybrid graph: the protein-only lists (sorted keys kept) of the listed protein rows
int td_launch_knn_general_static(const float4 *x4, const int32_t *node_ptr, const int32_t *pptr, const int32_t *gid,
                                 const int32_t *prot_rows, int64_t Np, int k, int max_graph_nodes, const int32_t *cptr,
                                 int32_t *snbr, unsigned long long *skeys, hipStream_t s) {
    return launch_knn_general<2>(x4, node_ptr, pptr, gid, prot_rows, Np, k, max_graph_nodes, cptr, snbr, s, skeys);
}

// Per-step neighbour rows of the ligand atoms on a general graph: k-NN rows over the whole graph, or (hybrid) the k nearest
// protein atoms behind the other ligand atoms of the graph (that half is step-invariant and written once by
// td_launch_hybrid_ligand_half).  The rows' slots must hold -1 where nothing is written (k-NN: fewer than k candidates).
int td_launch_ligand_rows_general(int mode, const float4 *x4, const int32_t *node_ptr, const int32_t *pptr, const int32_t *gid,
                                  const int32_t *lig_node, int64_t Nl, int k, int max_graph_nodes, const int32_t *cptr,
                                  int32_t *cnbr, hipStream_t s) {
    if (mode == 1) return launch_knn_general<1>(x4, node_ptr, pptr, gid, lig_node, Nl, k, max_graph_nodes, cptr, cnbr, s);
    return launch_knn_general<0>(x4, node_ptr, pptr, gid, lig_node, Nl, k, max_graph_nodes, cptr, cnbr, s);
}

// Fill the chunk table of a general graph.  mode 0: k-NN (any k <= 64) on every row; 1: hybrid (protein rows: k-NN over
// the whole graph; ligand rows: the other ligand atoms + the k nearest protein atoms); 2: radius r with fan-out cap.
int td_launch_graph_general(int mode, const float4 *x4, const int32_t *node_ptr, const int32_t *pptr, const int32_t *gid,
                            const int32_t *prot_node, int64_t Np, const int32_t *lig_node, int64_t Nl, int64_t N, int k,
                            float radius, int max_graph_nodes, const int32_t *cptr, int32_t *cnbr, int64_t NC, hipStream_t s) {
    TD_CHECK_HIP(hipMemsetAsync(cnbr, 0xff, (size_t)NC * TD_K * sizeof(int32_t), s));
    int rc = TD_OK;
    if (mode == 0) {
        rc = launch_knn_general<0>(x4, node_ptr, pptr, gid, nullptr, N, k, max_graph_nodes, cptr, cnbr, s);
    } else if (mode == 1) {
        rc = launch_knn_general<0>(x4, node_ptr, pptr, gid, prot_node, Np, k, max_graph_nodes, cptr, cnbr, s);
        if (rc == TD_OK && Nl > 0) {
            rc = td_launch_hybrid_ligand_half(node_ptr, pptr, gid, lig_node, Nl, cptr, cnbr, s);
            if (rc == TD_OK) rc = launch_knn_general<1>(x4, node_ptr, pptr, gid, lig_node, Nl, k, max_graph_nodes, cptr, cnbr, s);
        }
    } else {
        if (N > 0) {
            radius_kernel<<<dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s>>>(x4, node_ptr, gid, N, radius * radius, k, cptr, cnbr);
            TD_CHECK_HIP(hipGetLastError());
        }
    }
    return rc;
}

// dense [N][width] view (-1 padded) of the chunked table: what td_knn (k != 32) and td_graph_build hand out
__global__ void slots_to_dense_kernel(const int32_t *__restrict__ cptr, const int32_t *__restrict__ cnbr, int64_t N, int width,
                                      int32_t *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * width) return;
    const int64_t i = t / width;
    const int c = (int)(t - i * width);
    const int slots = (cptr[i + 1] - cptr[i]) * TD_K;
    out[t] = c < slots ? cnbr[(int64_t)cptr[i] * TD_K + c] : -1;
}

int td_launch_slots_to_dense(const int32_t *cptr, const int32_t *cnbr, int64_t N, int width, int32_t *out, hipStream_t s) {
    if (N == 0) return TD_OK;
    slots_to_dense_kernel<<<dim3((unsigned)((N * width + 255) / 256)), dim3(256), 0, s>>>(cptr, cnbr, N, width, out);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}

// radius graph with a fan-out cap <= 32 straight into the 32-slot neighbour table of the fast path
int td_launch_radius32(const float4 *x4, const int32_t *node_ptr, const int32_t *gid, int64_t N, float radius, int cap,
                       int32_t *nbr, hipStream_t s) {
    if (N == 0) return TD_OK;
    TD_CHECK_HIP(hipMemsetAsync(nbr, 0xff, (size_t)N * TD_K * sizeof(int32_t), s));
    radius_kernel<<<dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s>>>(x4, node_ptr, gid, N, radius * radius, cap, nullptr, nbr);
    TD_CHECK_HIP(hipGetLastError());
    return TD_OK;
}
