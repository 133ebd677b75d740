/*
 * tmpnn.h — C-ABI of the MI355X-native ThermoMPNN inference engine (libtmpnn.so).
 *
 * Drop-in boundary for the ONE hot path of Kuhlman-Lab/ThermoMPNN:
 *   tied_featurize tensors -> kNN graph -> edge featurizer -> 3x EncLayer -> 3x DecLayer -> ddG head.
 * Each entry point names the reference interface it replaces (file:line under /root/reference).
 *
 * Conventions
 *   - extern "C"; every function returns int (0 = TMPNN_OK, negative = error) unless noted;
 *     no exception crosses the boundary; tmpnn_last_error() returns a thread-local message.
 *   - All data pointers are BORROWED DEVICE pointers (e.g. torch tensor.data_ptr()); the caller owns
 *     every buffer including outputs and workspace. The library never allocates device memory and
 *     never synchronises the stream. `stream` is a hipStream_t passed as void*.
 *   - Ragged batch layout: proteins are packed along one residue axis of length T = sum(L_p);
 *     `offsets[N+1]` (int32, device) gives each protein's [start, end). A padded [B, L] batch of the
 *     reference API is the special case offsets = {0, L, 2L, ...} with mask = 0 on padding.
 *   - All floating-point data is fp32 (the reference computes in fp32 throughout); indices handed
 *     across this boundary are int32 except where an entry point mirrors a reference signature that
 *     takes int64 (gather_nodes / gather_edges).
 *   - Neighbour slots: every residue owns TMPNN_KS = 48 edge slots; slot k >= min(K, L_p) is
 *     invalid (E_idx = -1, h_E row = 0). K must be <= 48 (ThermoMPNN uses the v_48_* weights).
 */
#ifndef TMPNN_H
#define TMPNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TMPNN_VERSION 200
#define TMPNN_HID 128          /* hidden width (transfer_model.py:19) */
#define TMPNN_KS 48            /* neighbour slots per residue */
#define TMPNN_VOCAB 21
#define TMPNN_N_MPNN_TENSORS 118
#define TMPNN_N_TENSORS 130    /* + 12 TransferModel head tensors */

enum {
    TMPNN_OK = 0,
    TMPNN_E_INVALID = -1,      /* bad argument (null pointer, negative size, K out of range ...) */
    TMPNN_E_UNSUPPORTED = -2,  /* valid request outside what this build supports */
    TMPNN_E_LAUNCH = -3,       /* HIP launch/runtime error (message has hipGetErrorString) */
    TMPNN_E_WORKSPACE = -4,    /* workspace / packed buffer too small */
    TMPNN_E_RANGE = -5         /* a result left the finite range (tmpnn_status_error); retry with precision "bf16x3" */
};

/* Device-side status word (int32, caller-owned device memory, optional): kernels OR these bits in; the library never
 * reads it back (no stream sync). The host copies it after synchronising and passes it to tmpnn_status_error(). */
enum {
    TMPNN_STATUS_RANGE = 1,    /* a ddG / log-probability is inf or NaN: an activation or weight overflowed the fp16 range of
                                  the "f16x2" matrix-core path (|x| >= 65504) */
    TMPNN_STATUS_MAXLEN = 2,   /* a protein is longer than the max_len the caller passed: its neighbour rows were left
                                  empty (E_idx = -1) instead of overrunning the kernel's per-row scratch */
    TMPNN_STATUS_SELFTEST = 4  /* tmpnn_selftest: the library was built with flags that break its overflow detection or its
                                  persistent tile loops (-> TMPNN_E_UNSUPPORTED) */
};

typedef struct tmpnn_weights tmpnn_weights_t;   /* opaque; immutable after create */
typedef void *tmpnn_stream_t;                   /* hipStream_t */

int tmpnn_version(void);
const char *tmpnn_last_error(void);
/* Maps a status word (host copy) to an error: 0 -> TMPNN_OK; TMPNN_STATUS_MAXLEN -> TMPNN_E_INVALID;
 * TMPNN_STATUS_RANGE -> TMPNN_E_RANGE (message in tmpnn_last_error()). */
int tmpnn_status_error(int32_t status);
/* Device self-test of the build (one tiny launch on `stream`, no sync): the f16x2 kernels' GELU must propagate NaN (that is how
 * an fp16 overflow reaches TMPNN_STATUS_RANGE) and the kernels' view of gridDim / blockDim must match the launch. ORs
 * TMPNN_STATUS_SELFTEST into the caller-owned device word `status` on failure; hosts run it once per device before trusting a
 * freshly built library (thermompnn_amd/engine.py does, at the first weight handle). */
int tmpnn_selftest(int32_t *status, tmpnn_stream_t stream);
/* "f16x2" (default), "bf16x3" or "fp32": how the per-edge GEMMs (featurizer, message and edge-update kernels) run on
 * the matrix cores. f16x2 = every fp32 operand kept as two fp16 values x = h + l*2^-11 (22 significant bits), three
 * partial products per term on v_mfma_f32_16x16x32_f16 with fp32 accumulation; bf16x3 = exact three-way bf16 split, six
 * partial products on v_mfma_f32_16x16x32_bf16 (full fp32 range); fp32 = v_mfma_f32_16x16x4_f32. All three are in the
 * same accuracy class (see thermompnn_amd/csrc/tmpnn_split.h) and pass the same parity tests.
 * The precision is a property of the WEIGHT HANDLE (tmpnn_weights_create_p); this call returns the default that
 * handles created without an explicit precision get (environment variable TMPNN_PRECISION, else "f16x2"). */
const char *tmpnn_matmul_mode(void);

/* ---- weights ------------------------------------------------------------------------------------
 * Canonical tensor order = the reference state dict: the 118 ProteinMPNN tensors in module
 * registration order (protein_mpnn_utils.py:1184-1215), then light_attention.{feature,attention}
 * _convolution.{weight,bias}, both_out.{1,3,5}.{weight,bias}, ddg_out.{weight,bias}
 * (transfer_model.py:57-73,131-134). tmpnn_tensor_name(i) spells out entry i so a host can check. */
int tmpnn_num_tensors(void);
const char *tmpnn_tensor_name(int index);       /* NULL if out of range */
int64_t tmpnn_tensor_numel(int index);          /* -1 if out of range */

/* Device bytes the library needs for derived data: positional table, per-decoder-layer sequence tables, centre tap of the
 * feature convolution (0.8 MB) and — for "f16x2" handles only — the f16 fragment images of the weight blocks (7.2 MB).
 * tmpnn_weights_packed_bytes() is the upper bound over the precisions; _p gives the size for one precision (NULL = the
 * default; 0 for an unknown name). */
size_t tmpnn_weights_packed_bytes(void);
size_t tmpnn_weights_packed_bytes_p(const char *precision);

/* Replaces: get_protein_mpnn()'s load_state_dict (transfer_model.py:17-37) and the Lightning
 * checkpoint load (analysis/thermompnn_benchmarking.py:78-84) — the host reads the file, this call
 * takes the tensors. `tensors[i]` = device pointer to tensor i (fp32, contiguous). n_tensors is 118
 * (ProteinMPNN only: head calls then fail with TMPNN_E_INVALID) or 130. The raw tensors must stay
 * alive and unmodified for the handle's lifetime. Derived tables are written into `packed` on
 * `stream`. */
int tmpnn_weights_create(tmpnn_weights_t **out, const float *const *tensors, int n_tensors,
                         void *packed, size_t packed_bytes, tmpnn_stream_t stream);
/* Same, with the matrix-core precision of every call made through this handle: "f16x2" | "bf16x3" | "fp32", or NULL
 * for the default. Unknown names return TMPNN_E_INVALID. Several handles (e.g. one per precision) may share the raw
 * tensors; each needs its own `packed` buffer. */
int tmpnn_weights_create_p(tmpnn_weights_t **out, const float *const *tensors, int n_tensors,
                           void *packed, size_t packed_bytes, const char *precision, tmpnn_stream_t stream);
const char *tmpnn_weights_precision(const tmpnn_weights_t *w);
void tmpnn_weights_destroy(tmpnn_weights_t *w);

/* ---- graph construction ------------------------------------------------------------------------
 * Replaces ProteinFeatures._dist + torch.topk (protein_mpnn_utils.py:1101-1109).
 * X [T,4,3] (N,CA,C,O; NaN already zeroed as tied_featurize does, :577), mask [T].
 * E_idx [T,48] receives GLOBAL packed row indices sorted by ascending adjusted distance (ties: lower
 * index first), -1 in invalid slots; D_nb [T,48] the adjusted distances (0 in invalid slots).
 * max_len >= max_p L_p sizes the per-row LDS scratch (1 <= max_len <= 8192). The lengths live in device memory, so the
 * host cannot check them: a protein longer than max_len gets empty rows and TMPNN_STATUS_MAXLEN in *status_opt. */
int tmpnn_knn_topk(const float *X, const float *mask, const int32_t *offsets, int n_proteins,
                   int64_t T, int max_len, int K, int32_t *E_idx, float *D_nb, int32_t *status_opt,
                   tmpnn_stream_t stream);

/* compute_centrality (analysis/thermompnn_benchmarking.py:20-35): out[t] = #{other residues of the same protein
 * with |CA_t - CA_j| < radius}; residues without coordinates (mask 0) get -1 and are never counted. int32 [T]. */
int tmpnn_centrality(const float *X, const float *mask, const int32_t *offsets, int n_proteins, int64_t T,
                     float radius, int32_t *out, tmpnn_stream_t stream);

/* Replaces ProteinFeatures.forward after top-k (protein_mpnn_utils.py:1140-1180: 25 RBF blocks,
 * PositionalEncodings :896-908, edge_embedding, norm_edges) fused with W_e (:1229).
 * residue_idx, chain_enc: int32 [T]. h_E [T,48,128] <- W_e . LN(W_edge . [posenc | RBF]) + b.
 * E_opt (may be NULL) receives the LayerNorm output E [T,48,128] (what ProteinFeatures returns). */
int tmpnn_edge_featurize(const tmpnn_weights_t *w, const float *X, const int32_t *residue_idx,
                         const int32_t *chain_enc, const int32_t *E_idx, const float *D_nb, int64_t T,
                         float *h_E, float *E_opt, tmpnn_stream_t stream);

/* ---- gathers (the "gather HBM GB/s" kernel of BASELINE.json:metric) ------------------------------
 * gather_nodes(nodes[B,N,C], neighbor_idx[B,N,K] int64) -> [B,N,K,C]  (protein_mpnn_utils.py:770-778) */
int tmpnn_gather_nodes(const float *nodes, const int64_t *neighbor_idx, int B, int N, int K, int C,
                       float *out, tmpnn_stream_t stream);
/* Same gather on the engine's own layout: out[r,:] = nodes[idx[r],:], idx int32 global rows
 * (rows with idx < 0 are written as zeros). This is the roofline microbenchmark kernel. */
int tmpnn_gather_rows_i32(const float *nodes, const int32_t *idx, int64_t n_rows, int C, float *out,
                          tmpnn_stream_t stream);
/* gather_edges(edges[B,N,N,C], neighbor_idx[B,N,K] int64) -> [B,N,K,C]  (protein_mpnn_utils.py:763-767) */
int tmpnn_gather_edges(const float *edges, const int64_t *neighbor_idx, int B, int N, int K, int C,
                       float *out, tmpnn_stream_t stream);

/* ---- message-passing layers ---------------------------------------------------------------------
 * Workspace for one layer call / for the fused forward. */
size_t tmpnn_layer_workspace_bytes(int64_t T);
size_t tmpnn_workspace_bytes(int64_t T);

/* EncLayer.forward (protein_mpnn_utils.py:816-839) with mask_attend = mask_i*mask_j (:1232-1233).
 * h_V [T,128] and h_E [T,48,128] are updated in place. */
int tmpnn_enc_layer(const tmpnn_weights_t *w, int layer, float *h_V, float *h_E, const int32_t *E_idx,
                    const float *mask, int64_t T, void *workspace, size_t workspace_bytes,
                    tmpnn_stream_t stream);

/* One decoder step of ProteinMPNN.forward (protein_mpnn_utils.py:1268-1273 + DecLayer.forward
 * :859-880): h_ESV = mask_i * [h_E | W_s[S_j] | h_V_j], no neighbour mask. h_V_out may alias h_V_in. */
int tmpnn_dec_layer(const tmpnn_weights_t *w, int layer, const float *h_V_in, float *h_V_out,
                    const float *h_E, const int32_t *E_idx, const int32_t *S, const float *mask,
                    int64_t T, void *workspace, size_t workspace_bytes, tmpnn_stream_t stream);

/* h_S = W_s[S] (protein_mpnn_utils.py:1238). */
int tmpnn_seq_embed(const tmpnn_weights_t *w, const int32_t *S, int64_t T, float *h_S,
                    tmpnn_stream_t stream);
/* log_softmax(W_out h_V + b) (protein_mpnn_utils.py:1275-1276) -> [T,21]. */
int tmpnn_log_probs(const tmpnn_weights_t *w, const float *h_V, int64_t T, float *log_probs,
                    int32_t *status_opt, tmpnn_stream_t stream);

/* ---- ddG head ------------------------------------------------------------------------------------
 * TransferModel.forward's per-mutation body (transfer_model.py:86-120) evaluated once per POSITION:
 * x = [hV_last | hV_prev | W_s[S]], y = centre tap of LightAttention's feature conv (:148-155 on a
 * length-1 sequence), z = both_out(y) (:67-71). ddg [T,21]: ddg[t,a] = (w z_a + b) - (w z_S[t] + b)
 * (:110-116); z_opt (may be NULL) receives z [T,21]. */
int tmpnn_ddg_head(const tmpnn_weights_t *w, const float *hV_last, const float *hV_prev,
                   const int32_t *S, int64_t T, float *ddg, float *z_opt, int32_t *status_opt,
                   tmpnn_stream_t stream);

/* The same head for ANY configuration the reference constructor accepts (transfer_model.py:45-73): num_final_layers
 * n_final in 0..3 (hidden[0] = last decoder state, hidden[1] the one before, ... as all_mpnn_hid[:n] :84-85), LightAttention
 * on (conv_w [D0, D0, 9] + conv_b [D0], D0 = 128 n_final + 128; only the centre tap acts on a length-1 sequence) or off
 * (conv_w = conv_b = NULL, :106-108), any hidden_dims: dims[0] = D0, dims[1..n_layers-1] = hidden_dims, dims[n_layers] = 21;
 * mlp_w[l] [dims[l+1], dims[l]], mlp_b[l] = both_out's Linear l (ReLU in FRONT of each, :69-71). hidden, mlp_w, mlp_b, dims are
 * HOST arrays (of device pointers / ints); everything they point to, S and the outputs are device memory. ddg [T,21] as
 * tmpnn_ddg_head; z_opt [T,21]. The released configuration runs the specialised kernels behind tmpnn_ddg_head /
 * tmpnn_ssm_forward; this entry keeps retrained heads runnable (fp32 matrix cores, operands from global memory). */
size_t tmpnn_head_generic_workspace_bytes(int64_t T, int n_final, int n_layers, const int32_t *dims);   /* 0 = bad dims */
int tmpnn_ddg_head_generic(const float *const *hidden, int n_final, const float *Ws, const int32_t *S, int64_t T,
                           const float *conv_w, const float *conv_b, int n_layers, const float *const *mlp_w,
                           const float *const *mlp_b, const int32_t *dims, const float *ddg_w, const float *ddg_b, float *ddg,
                           float *z_opt, void *workspace, size_t workspace_bytes, int32_t *status_opt, tmpnn_stream_t stream);

/* ---- the fused path ------------------------------------------------------------------------------
 * Everything TransferModel.forward does on the device for a ragged batch of N proteins
 * (transfer_model.py:75-121 + protein_mpnn_utils.py:1222-1277), one call, 18 launches on `stream` (14 when every workgroup has at most one residue tile).
 * Outputs (each may be NULL; ddg needs a handle with the head tensors): ddg [T,21]; hidden_opt [3,T,128] = decoder states 1..3
 * (the reference returns them reversed, :1277); log_probs_opt [T,21]; E_idx_opt [T,48] global rows.
 * status_opt (device int32, may be NULL) is zeroed on the stream and then receives TMPNN_STATUS_* bits. */
int tmpnn_ssm_forward(const tmpnn_weights_t *w, const float *X, const int32_t *S, const float *mask,
                      const int32_t *residue_idx, const int32_t *chain_enc, const int32_t *offsets,
                      int n_proteins, int64_t T, int max_len, int K, float *ddg, float *hidden_opt,
                      float *log_probs_opt, int32_t *E_idx_opt, int32_t *status_opt, void *workspace,
                      size_t workspace_bytes, tmpnn_stream_t stream);

/* ---- host side: native PDB reader + packer ------------------------------------------
 * Replaces alt_parse_PDB (protein_mpnn_utils.py:183-350) + the packing of tied_featurize (:353-605) for one
 * structure: one pass over the file, all requested chains. `chains` = string of one-letter chain ids in the
 * order to concatenate them ("A", "AB", ...); NULL or "" = every chain present whose id is in the reference's default alphabet
 * (A-Z, a-z, 0-9, in that order; records of other chain ids are ignored, as the reference never reads them).
 * HOST pointers here (the only entry points that are not device-side). */
typedef struct tmpnn_pdb tmpnn_pdb_t;
int tmpnn_pdb_parse(const char *path, const char *chains, tmpnn_pdb_t **out);
/* n files on n_threads host threads; on failure every handle is released, outs[] is NULL and the message names the failing
 * files (up to eight of them). */
int tmpnn_pdb_parse_batch(const char *const *paths, const char *const *chains, int n, int n_threads,
                          tmpnn_pdb_t **outs);
/* The same with a per-file result: status [n] <- TMPNN_OK or the file's error code; a failing file leaves its handle NULL and
 * does NOT void the others (the call returns TMPNN_OK; tmpnn_last_error names the failing files) — a scan can skip or report
 * them (the reference's loop, analysis/SSM.py:105, would stop at the first bad structure). status == NULL: as above. */
int tmpnn_pdb_parse_batch_status(const char *const *paths, const char *const *chains, int n, int n_threads,
                                 tmpnn_pdb_t **outs, int32_t *status);
int64_t tmpnn_pdb_length(const tmpnn_pdb_t *p);      /* total residues L over the concatenated chains */
int tmpnn_pdb_num_chains(const tmpnn_pdb_t *p);
/* Any output may be NULL. X [L,4,3] fp32 with NaN -> 0, S [L] (ALPHABET index, gap -> 20), mask [L] (1 = all
 * four backbone atoms present), residue_idx [L] = 100 (c-1) + position, chain_enc [L] = c (1-based),
 * seq [L+1] = the parser's one-letter sequence ('-' at numbering gaps / unknown residues), NUL-terminated,
 * ca_mask [L] (1 = the CA atom is present: the mask compute_centrality uses, thermompnn_benchmarking.py:20-27). */
int tmpnn_pdb_fill(const tmpnn_pdb_t *p, float *X, int32_t *S, float *mask, int32_t *residue_idx,
                   int32_t *chain_enc, char *seq, float *ca_mask);
void tmpnn_pdb_free(tmpnn_pdb_t *p);
/* The parsed one-letter sequence (NUL-terminated, owned by the handle; same text tmpnn_pdb_fill copies out). */
const char *tmpnn_pdb_seq(const tmpnn_pdb_t *p);
/* n parsed structures -> ONE ragged batch in the caller's HOST buffers (pinned staging memory for an async H2D copy),
 * protein after protein in handle order, on n_threads host threads: offsets [n+1] (always written), then per residue the
 * arrays of tmpnn_pdb_fill (any may be NULL). capacity = residues the buffers hold; TMPNN_E_WORKSPACE if the batch is
 * longer. This is tied_featurize's packing (protein_mpnn_utils.py:353-605) for a whole chunk of a many-PDB scan. */
int tmpnn_pdb_pack_batch(tmpnn_pdb_t *const *handles, int n, int n_threads, int64_t capacity, float *X, int32_t *S,
                         float *mask, int32_t *residue_idx, int32_t *chain_enc, float *ca_mask, int32_t *offsets);

/* ---- host side: columnar result writer ------------------------------------------------
 * Replaces the cell-by-cell pandas frame + DataFrame.to_csv of analysis/SSM.py:102-176 (schema 0: ",WT Seq,Model,
 * Dataset,ddG_pred,position,wildtype,mutation,neighbors,best_AA,pdb") and analysis/custom_inference.py:64,94-111
 * (schema 1: ",Model,Dataset,ddG_pred,position,wildtype,mutation,pdb,chain"), byte for byte what pandas writes: '\n'
 * line ends, running index first, floats as repr(float(x)), empty cells for missing values (a NaN ddG included), minimal
 * quoting — pinned to files made by pandas itself (tests/golden/make_csv_golden.py).
 * HOST pointers; tables are the [T, ld] fp32 ddG tables of tmpnn_ssm_forward copied back (ld >= 20). */
typedef struct tmpnn_csv tmpnn_csv_t;
enum { TMPNN_CSV_PICK_BEST = 1,      /* one row per position carrying best_AA = argmin ddG (SSM.py:32-42,153-162); the schema-0
                                      * frame then has one more column, ",dupe_detector" = pdb + str(position) (:161), never dropped */
       TMPNN_CSV_INCLUDE_CYS = 2,    /* otherwise C is excluded from best_AA / rows mutating to C are dropped (:164-166) */
       TMPNN_CSV_NO_HEADER = 4 };    /* tmpnn_csv_open_ex: a part file of a sharded scan — no header line */
int tmpnn_csv_open(const char *path, int schema, tmpnn_csv_t **out);          /* creates the file, writes the header (of the first
                                                                               * listing written: PICK_BEST adds its column) */
/* The same with the header decided at once: flags = TMPNN_CSV_PICK_BEST (header with dupe_detector) | TMPNN_CSV_NO_HEADER. */
int tmpnn_csv_open_ex(const char *path, int schema, int flags, tmpnn_csv_t **out);
/* No file: the text goes into an anonymous mapping of `capacity` bytes (address space; pages are touched as text arrives), no
 * header line — one rank's share of a sharded scan, kept in memory until the ranks know where each protein's text belongs
 * (TMPNN_E_INVALID "No space left" when the listing outgrows the capacity). tmpnn_csv_mem -> the buffer and its fill; valid until
 * tmpnn_csv_close. */
int tmpnn_csv_open_mem(int schema, int flags, int64_t capacity, tmpnn_csv_t **out);
const char *tmpnn_csv_mem(const tmpnn_csv_t *c, int64_t *bytes_out);
/* The header line of a schema (flags: TMPNN_CSV_PICK_BEST) into buf -> its length, NUL-terminated (cap > length). */
int tmpnn_csv_header(int schema, int flags, char *buf, int cap);
/* Appends the listing of n proteins (may be called once per chunk of a scan; the running index continues). offsets [n+1]
 * index `table`; seqs[i] (length = rows of protein i; '-' positions are skipped) ; names[i] = 'pdb' cell; neighbors (may be
 * NULL) [T] -> 'neighbors' cell; `datasets` (may be NULL) per-protein 'Dataset' cells instead of `dataset`; chain: schema 1;
 * wt_cells (may be NULL) per-protein 'WT Seq' cells instead of seqs[i] — SSM.py:145-149 writes the DATASET's wild-type sequence
 * (dataset.wt_seqs[key]) there, which need not be the parsed structure's. */
int tmpnn_csv_write_ssm(tmpnn_csv_t *c, const float *table, int ld, const int32_t *offsets, int n,
                        const char *const *seqs, const char *const *wt_cells, const char *const *names, const int32_t *neighbors, const char *model,
                        const char *dataset, const char *const *datasets, const char *chain, int flags, int n_threads);
/* One rank's share of a scan that N ranks write into ONE file (the loop of SSM.py:105-176 sharded over GPUs): first_rows (may be
 * NULL) [n] = the running index of each protein's first row in the whole listing (every rank can compute them from the sequences
 * alone), bytes_out (may be NULL) [n] <- bytes of text written per protein; the ranks exchange those, and each places its
 * proteins' text at their offsets of the one output file (thermompnn_amd/dist.py: scan_files_to_csv). */
int tmpnn_csv_write_ssm_ex(tmpnn_csv_t *c, const float *table, int ld, const int32_t *offsets, int n,
                           const char *const *seqs, const char *const *wt_cells, const char *const *names, const int32_t *neighbors, const char *model,
                           const char *dataset, const char *const *datasets, const char *chain, int flags, int n_threads,
                           const int64_t *first_rows, int64_t *bytes_out);
/* Appends an explicit mutation list: triples [m,3] int64 (protein, 0-based position, amino-acid index < 20); schema 0. */
int tmpnn_csv_write_listed(tmpnn_csv_t *c, const float *table, int ld, const int32_t *offsets, int n,
                           const char *const *seqs, const char *const *names, const int32_t *neighbors, const char *model,
                           const char *dataset, const int64_t *triples, int64_t m);
int tmpnn_csv_close(tmpnn_csv_t *c, int64_t *rows_out, int64_t *bytes_out);   /* either may be NULL */
/* repr(float(v)) into buf (>= 32 bytes, NUL-terminated) -> length: the writer's number format, exposed for tests. */
int tmpnn_csv_format_double(double v, char *buf);

#ifdef __cplusplus
}
#endif
#endif /* TMPNN_H */
