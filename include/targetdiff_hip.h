/*
 * targetdiff_hip.h -- C ABI of libtargetdiff_hip.so: the MI355X-native (gfx950) implementation of the
 * TargetDiff denoising hot path.
 *
 * The reference (guanjq/targetdiff) is pure Python: it has no FFI / plugin interface, its seams are
 * Python call signatures.  Each entry point below replaces one of those seams and
 * cites it.  All pointers named d_* are DEVICE pointers (HIP), caller-owned; outputs are
 * caller-allocated; `stream` is a hipStream_t passed as void*.  Every function returns TD_OK (0) or a
 * negative TD_E* code; td_last_error() gives the message for the calling thread.  No exceptions cross
 * the ABI.  Process-wide state: the per-thread error string, the td_profile_* timers (measurement only) and a per-kernel,
 * per-device "dynamic LDS size configured" bit; every switch lives in the td_model handle (td_model_set_option), nothing is
 * read from the environment.  A td_model may be shared by streams/threads of its device; a workspace or a session must
 * not be shared by concurrent calls.
 *
 * Packed-graph convention (what compose_context produces, models/common.py:120-137): the batch holds B
 * graphs stored contiguously; inside a graph the protein atoms come first, then the ligand atoms.
 * node_ptr[b]..node_ptr[b+1] is graph b's node range.
 */
#ifndef TARGETDIFF_HIP_H
#define TARGETDIFF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: the entry points declared below are its whole dynamic symbol table. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define TD_OK 0
#define TD_EINVAL (-1)      /* bad argument / unsupported configuration */
#define TD_ENOMEM (-2)      /* workspace too small or allocation failure */
#define TD_EHIP (-3)        /* HIP runtime error (message in td_last_error) */

#define TD_ABI_VERSION 5

typedef struct td_model td_model;

/* Model hyper-parameters (configs/training.yml:9-42, read from the checkpoint's config at
 * scripts/sample_diffusion.py:158-162).  The HIP kernels are specialised for the live architecture (hidden 128, 16 heads,
 * 20 Gaussians, 4 edge types, uni_o2, global edge gate); td_model_create returns TD_EINVAL for anything else.  The graph
 * construction of models/uni_transformer.py:276-286 is a run-time choice:
 *   TD_CUTOFF_KNN     knn_graph(x, k = knn), any 1 <= knn <= 64; knn <= 32 (32 = the live configuration) takes the fast path in
 *                     which one dst row is one 32-row MFMA tile (slots >= knn masked: the knn nearest are the first knn of
 *                     the 32 nearest) and the sampling session caches the static protein;
 *   TD_CUTOFF_HYBRID  batch_hybrid_edge_connection(add_p_index=True) (models/common.py:165-212): a ligand atom sees every
 *                     other ligand atom of its graph and its knn nearest protein atoms, a protein atom its knn nearest nodes;
 *   TD_CUTOFF_RADIUS  radius graph with a fan-out cap: the first max_num_neighbors nodes j != i of the graph (index order)
 *                     with |x_i - x_j|^2 < radius^2.  The reference's own radius mode is dead code (:278 reads an attribute
 *                     that is never assigned); the rule is this project's (oracle/shims.py).
 * Rows wider or narrower than 32 run as chunks of 32 slots through ragged (CSR-segment) variants of the edge kernels. */
#define TD_CUTOFF_KNN 0
#define TD_CUTOFF_HYBRID 1
#define TD_CUTOFF_RADIUS 2
#define TD_MAX_FANIN 64         /* largest knn / max_num_neighbors */

typedef struct td_config {
    int32_t hidden_dim;          /* 128 */
    int32_t n_heads;             /* 16 */
    int32_t knn;                 /* 32 (1 .. 64) */
    int32_t num_layers;          /* 9 (any >= 1) */
    int32_t num_r_gaussian;      /* 20 */
    int32_t edge_feat_dim;       /* 4 */
    int32_t protein_feat_dim;    /* 27 (<= 32) */
    int32_t ligand_num_classes;  /* 13 (<= 16) */
    int32_t num_timesteps;       /* 1000 */
    int32_t cutoff_mode;         /* TD_CUTOFF_* (0 = knn) */
    float radius;                /* TD_CUTOFF_RADIUS: cut-off in Angstrom */
    int32_t max_num_neighbors;   /* TD_CUTOFF_RADIUS: fan-out cap (1 .. 64) */
    int32_t model_mean_type;     /* 0 = 'C0' (the network predicts x0; configs/training.yml), 1 = 'noise' (it predicts x_t + eps:
                                    x0 = sqrt_recip_alphas_cumprod[t] x_t - sqrt_recipm1_alphas_cumprod[t] (pred - x_t),
                                    models/molopt_score_model.py:412-416, 663-666; sampling only, as in the reference) */
    int32_t num_blocks;          /* 0 or 1 (configs/training.yml) .. 8: the layer stack is applied num_blocks times, the graph and the
                                    edge gate rebuilt from the current coordinates before every pass, same weights
                                    (models/uni_transformer.py:306-323).  > 1: sampling sessions do not cache (their static-protein
                                    tables describe the first pass only) */
    int32_t ew_net_type;         /* 0 = 'global' (configs/training.yml: one gate MLP on the step-start distances), 1 = 'r' (every x2h / h2x
                                    stage has its own Linear(4 x 20 -> 1) + sigmoid on the layer's radial features,
                                    models/uni_transformer.py:34-35, 60-61, 102-103, 124-125), 2 = none (any other value of the
                                    reference's option: e_w = 1), 3 = 'm' (x2h: sigmoid(Linear(128 -> 1)) of the edge's value vector,
                                    :36-37, 62-63; h2x: e_w = 1, :126-127).  Other than 'global': default graph only, sessions do not cache */
    int32_t x2h_out_fc;          /* 1: h += node_output([attention output | h]) after every x2h stage (models/uni_transformer.py:39-40,
                                    81-84; the reference class's default, False in configs/training.yml); sessions do not cache */
    int32_t sync_twoup;          /* 1: the h2x stage of a layer reads the layer's INPUT features instead of its x2h output
                                    (models/uni_transformer.py:198); 0 in configs/training.yml and in the class */
    int32_t num_x2h, num_h2x;    /* stages per layer, 1 .. 4 (0 = 1, as in configs/training.yml): a layer runs num_x2h x2h stages on its
                                    start coordinates, then num_h2x h2x stages (models/uni_transformer.py:190-206), each stage with its
                                    own weights.  != 1: sessions do not cache, nothing is fused across stages */
    int32_t reserved[1];         /* zero */
} td_config;                     /* (ABI 5: the struct grew by 16 bytes; ew_net_type / x2h_out_fc took the place of ABI 4's reserved[2]) */

/* ---- library ------------------------------------------------------------------------------------ */
int td_abi_version(void);
const char *td_last_error(void);
/* 12 hex digits naming the SOURCES this binary was built from (SHA-256 over csrc/, this header and the compiler flags, set by
 * targetdiff_amd/build.py): bench lines and committed PMC profiles carry it, so a profile can be matched to the build it describes. */
const char *td_build_tag(void);

/* ---- model (replaces: ScorePosNet3D.__init__ + load_state_dict, models/molopt_score_model.py:194-311,
 *      scripts/sample_diffusion.py:158-163).  `host_weights` is the flat fp32 blob of the reference
 *      state_dict tensors in the order documented in targetdiff_amd/capi.py::flatten_state_dict
 *      (row-major [out, in] exactly as PyTorch stores them); the library re-packs them for its kernels
 *      (node-side projection split of the 340-wide first Linear, MFMA fragment order) and uploads them.
 *      `host_schedules` = 7 arrays of num_timesteps fp32 each, in this order: posterior_mean_c0_coef,
 *      posterior_mean_ct_coef, posterior_logvar, log_alphas_v, log_one_minus_alphas_v,
 *      log_alphas_cumprod_v, log_one_minus_alphas_cumprod_v (models/molopt_score_model.py:248-267); optionally an 8th,
 *      alphas_cumprod (td_perturb / td_likelihood_prior), and a 9th and 10th, sqrt_recip_alphas_cumprod and
 *      sqrt_recipm1_alphas_cumprod (:228-229; required by model_mean_type = 1). */
int td_model_create(const td_config *cfg, const float *host_weights, size_t num_weights,
                    const float *host_schedules, size_t num_schedule_floats, td_model **out);
void td_model_destroy(td_model *m);
size_t td_model_num_weights(const td_config *cfg);      /* expected length of the flat blob */
/* Per-model switches (stored in the handle: per device, nothing is read from the environment).  Set them before the model
 * is used; not to be changed while a call is in flight.
 *   "node_proj_split"        1 (default): the node-side 128 x 128 GEMMs run on v_mfma_f32_32x32x16_bf16 with both operands
 *                            split exactly into three bf16 pieces (an fp32 significand = 3 x 8 bits; 6 of the 9 piece products,
 *                            fp32 accumulation): fp32-equivalent results, 0 = plain fp32 MFMA
 *   "edge_key_split"         1 (default): the 21-wide radial / edge-type first layer of the attention passes on 32-slot rows
 *                            (x2h key and value passes, the h2x stage), of the chunked key pass and of the edge gate on v_mfma_f32_16x16x32_bf16
 *                            with the same exact three-piece split of both operands; 0 = fp32 MFMA (v_mfma_f32_16x16x4_f32)
 *   "edge_first_layer_f16"   1 (default): the 21-wide radial / type first layer of the attention kernels (x2h key / value passes, h2x stage) on v_mfma_f32_16x16x32_f16, weights
 *                            and inputs as pairs of f16 pieces (22 significant bits each; the MLP's first layer -- node projections included --
 *                            is packed in units of a power of two so that no piece meets the f16 subnormal floor): two products and two
 *                            16-byte table reads per tile; 0 = the exact bf16 x 3 piece triples (four products, three reads).  Needs
 *                            "edge_key_split"; the edge gate uses the bf16 form under either setting
 *   "edge_second_layer_f16"  1 (default): the per-edge 128-deep products of the x2h key / value passes (logits = z . U_i, alpha^T z) on
 *                            v_mfma_f32_16x16x32_f16, both operands as pairs of f16 pieces (22 significant bits each, scaled by exact
 *                            powers of two; three piece products, fp32 accumulation): within one to two fp32 roundings of the fp32
 *                            products; 0 = v_mfma_f32_16x16x4_f32 on the fp32 values.  Applies to the value pass
 *                            on every graph and to the key pass on rows of one 32-slot chunk (the default k = 32 graph; the protein rows
 *                            of `hybrid`, k < 32 and capped-radius graphs); the key pass of rows that span several chunks (k > 32, the
 *                            ligand rows of `hybrid`) follows it only while "edge_first_layer_f16" is on (fp32 logits otherwise)
 *   "h2x_fused"              1 (default): key + value halves of the h2x stage in one launch; 0 = two launches
 *   "session_hop_levels"     1 .. 4 (default 4): receptive-field levels a sampling session prunes the last layers with
 *   "session_forward_reach"  1 (default): layer 1 of a session runs on the ligand's one-hop forward reach only
 *   "session_share_pockets"  1 (default): a sampling session on the default graph keeps its static tables (protein-only sorted k-NN keys and lists,
 *                            cached gate rows, embeddings, layer-0 / layer-1 outputs of the protein-only graph) once per DISTINCT protein block
 *                            of the batch -- all samples of a pocket carry the same block (scripts/sample_diffusion.py:42) -- found by a hash +
 *                            bitwise comparison at td_session_create; same results bit for bit; 0 = once per graph
 *   "session_step_lists"     1 (default): the row lists of a session step come from one launch (a workgroup per graph);
 *                            0 = the separate list kernels (also used when a graph exceeds 12288 nodes) */
int td_model_set_option(td_model *m, const char *name, int32_t value);
int td_model_get_option(const td_model *m, const char *name, int32_t *value);

/* ---- workspace ---------------------------------------------------------------------------------- */
/* Bytes of scratch td_refine_forward / td_model_forward need for a batch of N nodes (N_l of them ligand)
 * in B graphs.  Graph modes other than the 32-slot default (k > 32, hybrid, radius cap > 32), td_knn and td_graph_build
 * additionally take ONE stream-ordered block (hipMallocAsync on `stream`, freed in stream order before the call returns) for
 * the chunked neighbour table, whose size depends on the per-graph atom counts and is not known from (N, B, N_l) alone. */
size_t td_workspace_bytes(const td_model *m, int64_t N, int64_t B, int64_t N_l);

/* ---- graph bookkeeping ---------------------------------------------------------------------------
 * batch [N] int64, sorted ascending (torch_geometric `batch` vector) -> ptr [B+1] int32. */
int td_graph_ptr(const int64_t *d_batch, int64_t N, int64_t B, int32_t *d_ptr, void *stream);

/* ---- kNN graph (replaces: torch_geometric.nn.knn_graph(x, k, batch, flow='source_to_target') at
 *      models/uni_transformer.py:280).  out_nbr [N, k] int32: row i = the k nearest same-graph nodes of
 *      node i, ascending by (d2, index), d2 = (dx*dx + dy*dy) + dz*dz in fp32 without FMA contraction;
 *      -1 padded when the graph has fewer than k+1 nodes.  1 <= k <= 64.  `max_graph_nodes` is a performance hint
 *      (0 = unknown). */
int td_knn(const float *d_x /*[N,3]*/, const int32_t *d_node_ptr /*[B+1]*/, int64_t N, int64_t B, int32_t k,
           int32_t max_graph_nodes, int32_t *d_out_nbr, void *stream);

/* ---- the model's graph on a composed batch (replaces: UniTransformerO2TwoUpdateGeneral._connect_edge,
 *      models/uni_transformer.py:276-286, for the model's cutoff_mode).  d_out_nbr [N, width] int32, -1 padded: the
 *      in-neighbours (edge sources) of node i.  kNN: ascending (d2, index).  hybrid ligand rows: the other ligand atoms
 *      (ascending index), then the knn nearest protein atoms (ascending (d2, index)).  radius: ascending index.  Entries
 *      beyond `width` are dropped.  Needs compose_context order (protein rows first in every graph). */
int td_graph_build(const td_model *m, const float *d_x, const uint8_t *d_mask_ligand, const int32_t *d_node_ptr, int64_t N,
                   int64_t B, int32_t max_graph_nodes, int32_t *d_out_nbr, int32_t width, void *stream);

/* ---- backbone (replaces: refine_net(h, x, mask_ligand, batch, return_all=False, fix_x) ->
 *      {'x','h'}, UniTransformerO2TwoUpdateGeneral.forward, models/uni_transformer.py:301-328).
 *      d_h [N,128] f32, d_x [N,3] f32, d_mask_ligand [N] uint8, d_node_ptr [B+1] int32.
 *      d_out_nbr ([N,32] int32) and d_out_ew ([N,32] f32, the global edge gate of :312-316) may be NULL; they exist for
 *      the default k = 32 kNN graph only (other graphs: td_graph_build). */
int td_refine_forward(const td_model *m, const float *d_h, const float *d_x, const uint8_t *d_mask_ligand,
                      const int32_t *d_node_ptr, int64_t N, int64_t B, int32_t fix_x, int32_t max_graph_nodes,
                      float *d_out_h, float *d_out_x, int32_t *d_out_nbr, float *d_out_ew,
                      void *d_workspace, size_t workspace_bytes, void *stream);

/* ---- one denoiser evaluation (replaces: ScorePosNet3D.forward, models/molopt_score_model.py:313-368).
 *      Protein / ligand atoms are given un-composed, each sorted by graph:
 *      d_protein_ptr / d_ligand_ptr are [B+1] int32 prefix offsets.  Outputs: pred_ligand_pos [N_l,3],
 *      pred_ligand_v [N_l,C], final_ligand_h [N_l,128]; d_final_h [N,128] may be NULL.
 *      d_ligand_graph_bias (NULL when time_emb_dim == 0, the live configuration): [B][128] fp32, row g is added to the embedding
 *      of every ligand atom of graph g before the bias -- the time-embedding columns of ligand_atom_emb applied to the graph's
 *      time feature (:319-329: `simple` = (t / T) * W[:, C], `sin` = W[:, C:] time_emb(t)); column 127 (the node indicator) is 0.
 *      The caller evaluates that small per-graph term (the mirror does it with torch). */
int td_model_forward(const td_model *m, const float *d_protein_pos, const float *d_protein_v,
                     const int32_t *d_protein_ptr, int64_t N_p, const float *d_ligand_pos,
                     const int64_t *d_ligand_v, const int32_t *d_ligand_ptr, int64_t N_l, int64_t B,
                     int32_t fix_x, int32_t max_graph_nodes, float *d_pred_ligand_pos, float *d_pred_ligand_v,
                     float *d_final_ligand_h, float *d_final_h, void *d_workspace, size_t workspace_bytes,
                     const float *d_ligand_graph_bias, void *stream);

/* ---- posterior update of one reverse-diffusion step (replaces the loop body
 *      models/molopt_score_model.py:673-685: q_pos_posterior :424, extract :706, q_v_posterior :401,
 *      q_v_pred :383, q_v_pred_one_timestep :371, log_add_exp :173, index_to_log_onehot :124,
 *      log_sample_categorical :160).  d_t [B] int32 per-graph timestep; d_noise [N_l,3] ~ N(0,1) and
 *      d_uniform [N_l,C] ~ U[0,1) are supplied by the caller (torch.randn_like / rand_like in the
 *      reference).  d_log_v0 / d_log_post ([N_l,C]) may be NULL. */
int td_posterior_step(const td_model *m, const int32_t *d_t, const int32_t *d_ligand_ptr, int64_t N_l,
                      int64_t B, const float *d_ligand_pos, const int64_t *d_ligand_v,
                      const float *d_pred_pos, const float *d_pred_v, const float *d_noise,
                      const float *d_uniform, float *d_pos_next, int64_t *d_v_next, float *d_log_v0,
                      float *d_log_post, void *stream);

/* ---- standalone EGNN refine net (replaces: models/egnn.py EGNN / EnBaseLayer as get_refine_net('egnn', config) builds
 *      it, models/molopt_score_model.py:34-42: num_r_gaussian = 1, kNN rebuilt per layer, SiLU, no LayerNorm, hidden 128,
 *      4 edge types, k = 32).  `host_weights`: per layer, in this order and as PyTorch stores them: edge_mlp.net.0.{weight
 *      [128,261], bias}, edge_mlp.net.2.{weight, bias}, edge_inf.0.{weight [1,128], bias [1]}, x_mlp.0.{weight, bias},
 *      x_mlp.2.weight [1,128], node_mlp.net.0.{weight [128,256], bias}, node_mlp.net.2.{weight, bias}.
 *      td_egnn_forward = EGNN.forward(h, x, mask_ligand, batch, return_all) (models/egnn.py:121-133): d_all_h [L][N][128] /
 *      d_all_x [L][N][3] (optional) receive the state after every layer (all_h[1:], all_x[1:]). */
typedef struct td_egnn td_egnn;
size_t td_egnn_num_weights(int32_t num_layers);
int td_egnn_create(int32_t num_layers, int32_t hidden_dim, int32_t edge_feat_dim, int32_t knn, const float *host_weights,
                   size_t num_weights, td_egnn **out);
void td_egnn_destroy(td_egnn *m);
size_t td_egnn_workspace_bytes(int64_t N);
int td_egnn_forward(const td_egnn *m, const float *d_h, const float *d_x, const uint8_t *d_mask_ligand,
                    const int32_t *d_node_ptr, int64_t N, int64_t B, int32_t max_graph_nodes, float *d_out_h, float *d_out_x,
                    float *d_all_h, float *d_all_x, void *d_workspace, size_t workspace_bytes, void *stream);

/* ---- other consumers of the denoiser (scripts/likelihood_est_diffusion.py; ScorePosNet3D.forward(return_all=True)).
 * They need the 8th schedule array (alphas_cumprod of the position schedule) at td_model_create.
 *
 * td_perturb: the forward-process sample inside ScorePosNet3D.likelihood_estimation (models/molopt_score_model.py:577-588;
 *   q_v_sample :394-398): pos_t = sqrt(abar_t) pos_0 + sqrt(1 - abar_t) noise; v_t = argmax(gumbel(uniform) + log q(v_t|v_0)).
 * td_likelihood_terms: per-graph kl_pos, kl_v of the same function (:594-613 = q_pos_posterior :424-428, q_v_posterior
 *   :401-409, compute_pos_Lt :463-474, compute_v_Lt :476-483): posterior KL for t > 0, decoder NLL for t == 0, mean over
 *   the graph's ligand atoms.  kl_*: [B].
 * td_likelihood_prior: the time_step == T branch (:569-576 = kl_pos_prior :430-438, kl_v_prior :410-416).  v_index is
 *   what the reference feeds index_to_log_onehot there (it passes batch_ligand, :574).
 * td_embed_ligand / td_v_inference: ligand_atom_emb (+ node indicator, :317,334,338) and v_inference (:307-311) on
 *   free-standing rows -- the block-input entries of layer_pred_ligand_v (:360-367). */
int td_perturb(const td_model *m, const int32_t *d_t, const int32_t *d_ligand_ptr, int64_t N_l, int64_t B,
               const float *d_ligand_pos, const int64_t *d_ligand_v, const float *d_noise, const float *d_uniform,
               float *d_pos_t, int64_t *d_v_t, void *stream);
int td_likelihood_terms(const td_model *m, const int32_t *d_t, const int32_t *d_ligand_ptr, int64_t N_l, int64_t B,
                        const float *d_pos_0, const float *d_pos_t, const int64_t *d_v_0, const int64_t *d_v_t,
                        const float *d_pred_pos, const float *d_pred_v, float *d_kl_pos, float *d_kl_v, void *stream);
int td_likelihood_prior(const td_model *m, const int32_t *d_ligand_ptr, int64_t N_l, int64_t B, const float *d_pos_0,
                        const int64_t *d_v_index, float *d_kl_pos, float *d_kl_v, void *stream);
int td_embed_ligand(const td_model *m, const int64_t *d_ligand_v, int64_t N_l, float *d_h, void *stream);
int td_v_inference(const td_model *m, const float *d_h, int64_t n, float *d_logits, void *stream);

/* ---- centring (replaces: center_pos(mode='protein'), models/molopt_score_model.py:110-120).
 *      offset [B,3] = per-graph protein centroid; positions are shifted in place by -offset (sign = -1)
 *      or +offset (sign = +1, models/molopt_score_model.py:691,695).  d_protein_pos may be NULL to
 *      shift only the ligand with an offset computed earlier (compute_offset = 0). */
int td_center_pos(float *d_protein_pos, const int32_t *d_protein_ptr, float *d_ligand_pos,
                  const int32_t *d_ligand_ptr, int64_t B, float *d_offset, int32_t compute_offset,
                  int32_t sign, void *stream);

/* ---- sampling session (replaces the loop-invariant part of ScorePosNet3D.sample_diffusion,
 *      models/molopt_score_model.py:633-661: protein_pos / protein_v / batch_protein are the same tensors in every
 *      one of the ~1000 forward calls and protein coordinates never move, models/uni_transformer.py:206).
 *      td_session_create takes the CENTRED protein once and precomputes embeddings, protein-only sorted neighbour
 *      lists and -- for protein atoms whose 32-NN row no ligand atom enters -- the edge-gate row and the layer-0 x2h
 *      output.  td_session_forward = ScorePosNet3D.forward for the current ligand state; its results equal
 *      td_model_forward's (same kernels, same per-row arithmetic; neighbour rows bit-identical).  The session owns its
 *      device memory; one forward at a time per session.  The memory is stream-ordered (hipMallocAsync / hipFreeAsync): the
 *      stream of the session's latest call must still exist when td_session_destroy runs (if it does not, the free falls back
 *      to a device synchronisation + hipFree). */
typedef struct td_session td_session;
int td_session_create(const td_model *m, const float *d_protein_pos, const float *d_protein_v,
                      const int32_t *d_protein_ptr, int64_t N_p, const int32_t *d_ligand_ptr, int64_t N_l, int64_t B,
                      int32_t max_graph_nodes, void *stream, td_session **out);
void td_session_destroy(td_session *s);
int td_session_forward(td_session *s, const float *d_ligand_pos, const int64_t *d_ligand_v, float *d_pred_ligand_pos,
                       float *d_pred_ligand_v, float *d_final_ligand_h, const float *d_ligand_graph_bias, void *stream);
/* ---- one reverse-diffusion step as a single replayable unit (replaces the loop body of ScorePosNet3D.sample_diffusion,
 *      models/molopt_score_model.py:650-693: forward, posterior mean / variance + noise :673-679, categorical posterior +
 *      Gumbel-max draw :682-685, trajectory appends :687-693).  = td_session_forward on the current ligand state followed by
 *      td_posterior_step, with everything that differs from step to step held in DEVICE memory, so that the ~50 launches of a
 *      step form one hipGraph that is captured once (second call) and replayed:
 *        d_step[0]   index s of the step to run; incremented on the device when the step's last kernel finishes
 *                    (d_step[1] is scratch of that hand-over and must start as 0).  The caller zeroes both once.
 *        d_t_all     [num_steps][B] the time step of every graph at step s (:649, :652)
 *        d_ligand_pos / d_ligand_v   the CURRENT state x_t / v_t: read by the step, then overwritten with x_{t-1} / v_{t-1}
 *                    (pos_only, :681: the types are left alone)
 *        d_noise [N_l][3], d_uniform [N_l][C]   this step's draws (:677, :161), refilled by the caller before every call
 *        d_pos_traj [num_steps][N_l][3], d_v_traj [num_steps][N_l], d_v0_traj / d_vt_traj [num_steps][N_l][C] (or NULL):
 *                    slot s receives x_{t-1}, v_{t-1}, log v0 (:683) and the log posterior (:684)
 *      use_graph = 0 issues the launches one by one (also the behaviour while the kernel timers or the workgroup trace are
 *      armed, and after a failed capture): the same kernels with the same arguments either way, so the results are the same
 *      bits.  The captured graph belongs to the session and is re-captured if `io` changes.  td_session_step_graph reports
 *      whether the last td_session_step replayed a graph (1) or launched eagerly (0). */
typedef struct td_step_io {
    int32_t *d_step;
    const int32_t *d_t_all;
    int32_t num_steps;
    int32_t pos_only;
    float *d_ligand_pos;
    int64_t *d_ligand_v;
    const float *d_noise;
    const float *d_uniform;
    float *d_pos_traj;
    int64_t *d_v_traj;
    float *d_v0_traj;
    float *d_vt_traj;
    const float *d_ligand_graph_bias;   /* [B][128] or NULL: this step's time-embedding term (see td_model_forward) */
} td_step_io;
int td_session_step(td_session *s, const td_step_io *io, int32_t use_graph, void *stream);
int td_session_step_graph(const td_session *s);
/* rows processed by the last td_session_forward: counts[0] = N, counts[1] = rows recomputed at layer 0 (ligand +
 * displaced protein rows), counts[2 + k] = size of receptive-field level k + 1 of the ligand outputs (level 1 = ligand
 * atoms + their neighbours, level k + 1 = level k + its neighbours; the layer e from the end updates level e + 1 only),
 * -1 for levels the session does not track (k < 4); counts[6] = rows of layer 1 inside the ligand's one-hop forward reach
 * (the others keep the protein-only graph's cached layer-1 output), -1 when off; counts[7] = rows of the session's static tables and
 * counts[8] = distinct protein blocks of the batch when the tables are shared per pocket ("session_share_pockets"), -1 otherwise; synchronises */
int td_session_row_counts(td_session *s, int32_t *host_counts, int32_t n_counts, void *stream);

/* ---- kernel timers (measurement only; process-global: one set of timers for all models and streams of the process.  Launches from
 *      several host threads may record concurrently -- every start / stop pair enters the lists under a lock -- but begin / end
 *      themselves are meant for one controlling thread).  td_profile_begin arms HIP-event
 *      timers around the kernel classes selected by `class_mask` (bit c = class c) on the launch stream;
 *      td_profile_end synchronises the device and returns per-class summed milliseconds and launch counts.
 *      Classes: 0 knn, 1 edge gate, 2 node projections, 3 x2h key pass, 4 x2h value pass, 5 h2x key pass,
 *      6 h2x value pass, 7 compose, 8 head, 9 posterior. */
#define TD_PROFILE_CLASSES 10
int td_profile_begin(uint32_t class_mask);
int td_profile_end(float *ms_out, int32_t *count_out, int32_t num_classes);

/* ---- test hook (not a reference seam): node-side GEMMs of one attention stage of layer `layer`
 *      (stage 0 = x2h: hk/hv/hq, stage 1 = h2x: xk/xv/xq).  d_P [N,512] = [k_i | k_j | v_i | v_j] node
 *      projections of the 340-wide first Linear (h_i part incl. bias) in the form the kernels consume them -- each block centred
 *      over its 128 hidden units and multiplied by the sign of the MLP's LayerNorm weight (the LayerNorm is folded into the two
 *      Linears at pack time) --, d_q [N,128] = MLP_q(h). */
int td_debug_node_stage(const td_model *m, int32_t layer, int32_t stage, const float *d_h, int64_t N, float *d_P,
                        float *d_q, void *stream);

/* ---- test hook: one wave evaluates the cross-lane reduction helpers on 64 inputs; out[6][64] =
 *      {sum over groups of 8, sum over half-waves, sum over the wave, lo+hi half sum, lo/hi half max, other half}. */
int td_debug_reductions(const float *d_in64, float *d_out6x64, void *stream);

/* ---- test hook (fault injection): the nth stream-ordered device allocation the library makes from now on fails with
 *      TD_ENOMEM (0 = off).  Used to check that no entry point leaks the blocks it took before the failing one. */
int td_debug_fail_alloc(int32_t nth);

/* ---- profiling hook: wall clock of every workgroup of the x2h key / x2h value / fused h2x launches.  d_buf [slots][3 passes]
 *      [256 workgroups][8] uint64 in s_memrealtime ticks (the 100 MHz reference clock: comparable across CUs): 0 = the row loop
 *      starts (tables staged), 1 = end, 2 = end of the workgroup's first finished wave, 3 = sum of its waves' ends, 4 = kernel entry;
 *      launch n of a pass writes slot n % slots (pass 0 = key, 1 = value, 2 = h2x).  The caller fills entries 2 with all-ones and
 *      the rest with zero before the step (atomic min / add).  d_buf == NULL switches it off (the default; one scalar test per
 *      kernel). */
int td_debug_wg_trace(uint64_t *d_buf, int32_t slots);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* TARGETDIFF_HIP_H */
