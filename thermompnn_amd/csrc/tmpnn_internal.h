// Internal host-side declarations shared by the .hip translation units of libtmpnn.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tmpnn.h"

struct EncW {   // device pointers into the raw state-dict tensors of one EncLayer
    const float *norm1_w, *norm1_b, *norm2_w, *norm2_b, *norm3_w, *norm3_b;
    const float *W1, *b1, *W2, *b2, *W3, *b3, *W11, *b11, *W12, *b12, *W13, *b13;
    const float *Win, *bin, *Wout, *bout;
};
struct DecW {
    const float *norm1_w, *norm1_b, *norm2_w, *norm2_b;
    const float *W1, *b1, *W2, *b2, *W3, *b3;
    const float *Win, *bin, *Wout, *bout;
};
// Pre-built f16x2 MFMA A-fragment image of one 128 x 128 weight block for the 8-wavefront node kernels: 64 KB,
// [wavefront 8][k-step 4][plane 2][lane 64] x 16 B — a wavefront's fragment is four coalesced 1 KB loads per plane
// instead of 16-row gathers of fp32 that are split on the fly (tmpnn_split.hip: node_update8_split_kernel).
#define TM_WIMG_BYTES 65536
#define TM_N_WIMG 110          // enc: W3 + 4 W_in + 4 W_out + W1a W1c W11a W11c + W1e W2 W11e W12 W13 (18) x 3; dec: W3 + 4 + 4 + W1a W1d + W1e W2 (13) x 3; head: 9 blocks of the centre tap + 3 of both_out.1; featurizer: 4 blocks of W_edge[:, 16:416] + W_e
#define TM_N_WIMGP 21          // the message kernels' W1e and W2 (3 encoder + 3 decoder layers) again with the K axis permuted inside every 32-deep step: 12;
                               // + room for the edge update's W11e, W12, W13 (3 layers), which only the debug library builds (tmpnn_edge_wave.hip; the
                               // same struct layout in both libraries: A/B variants link debug objects of a few files with the shipped ones)
                               // (msg8_wave_kernel, tmpnn_msg.hip): element e of lane group q <-> k = 32 c + 16 (e >> 2) + 4 q + (e & 3)
#ifdef TMPNN_DEBUG_BUILD
#define TM_N_WIMGP_BUILT 21    // images a handle of THIS library builds (and its packed buffer has room for)
#else
#define TM_N_WIMGP_BUILT 12
#endif
struct WImg { const float *base; const char *img; };      // base = address of the block's element [0][0] in the raw tensor

struct tmpnn_weights {
    int n_tensors;
    int mode;              // TM_MM_*: matrix-core path of this handle's per-edge GEMMs
    WImg wimg[TM_N_WIMG];  // derived fragment images (in the caller's packed buffer), looked up by block base address
    int n_wimg;
    WImg wimgp[TM_N_WIMGP]; // K-permuted images (few: linear search)
    int n_wimgp;
    const float *t[TMPNN_N_TENSORS];
    // features
    const float *pos_w, *pos_b, *edge_w, *norm_edges_w, *norm_edges_b, *We_w, *We_b, *Ws_w;
    EncW enc[3];
    DecW dec[3];
    const float *Wout_w, *Wout_b;
    // head
    const float *conv_w, *conv_b, *mlp_w[3], *mlp_b[3], *ddg_w, *ddg_b;
    // derived tables (in the caller's packed buffer)
    float *pos_table;      // [66,128]   (W_pos^T + b_pos) . W_edge[:, :16]^T
    float *seq_table[3];   // [21,128]   W_s . W1_dec[l][:, 256:384]^T
    float *conv_center;    // [384,384]  feature_convolution.weight[:, :, 4]
};

// scratch carved out of the caller's workspace for the message-passing layers
struct LayerWs {
    float *P;      // [T,256] node projections (A | C) for the message pass
    float *P2;     // [T,256] node projections for the encoder edge update
    float *Ssum;   // [T,128] sum_k mask_k * m2_k
    float *cnt;    // [T]     sum_k mask_k
};

int tm_set_error(int code, const char *fmt, ...);
int tm_check_launch(const char *what);
void tm_prof_begin(const char *name, hipStream_t st);   // no-ops unless tmpnn_profile_enable(1)
void tm_prof_end(hipStream_t st);

// tmpnn_graph.hip
// Optional extra of the fused forward: the k-NN kernel (one wavefront per residue) also writes the residue's all-zero initial
// node state hV0[t, 0:128] and its message projection P[t] = [ba | 0] (what node_proj of a zero state gives, exactly) —
// two launches (a memset and a fill) fewer per forward.
// fused forward: the k-NN kernel writes the zero state + first projection and ZEROES the caller's status word (workgroup 0; it
// then must not OR into that word itself — rows of an over-long protein are flagged by the last kernel, HeadArgs::maxlen_probe)
struct KnnInit { float *hV0; float *P; const float *ba; int32_t *status_zero; };
int launch_knn(const float *X, const float *mask, const int32_t *offsets, int N, int64_t T, int max_len, int K,
               int32_t *E_idx, float *D_nb, int32_t *status, hipStream_t st, KnnInit init = KnnInit{nullptr, nullptr, nullptr, nullptr});
int launch_centrality(const float *X, const float *mask, const int32_t *offsets, int N, int64_t T, float radius,
                      int32_t *out, hipStream_t st);
// small launches: the k-NN rows computed inside the featurizer launch (E_idx / D_nb are then OUTPUTS of it)
struct KnnFuse { const float *mask; const int32_t *offsets; int N, max_len, K; int32_t *E_idx; float *D_nb; KnnInit init; };
bool featurize_fusable(const tmpnn_weights *w, int64_t T);
int launch_featurize(const tmpnn_weights *w, const float *X, const int32_t *ridx, const int32_t *cenc,
                     const int32_t *E_idx, const float *D_nb, int64_t T, float *h_E, float *E_opt, hipStream_t st,
                     const KnnFuse *knn = nullptr);
int launch_gather_rows(const float *nodes, const void *idx, int idx64, int64_t n_rows, int64_t rows_per_batch,
                       int64_t nodes_per_batch, int C, float *out, hipStream_t st);
int launch_gather_edges(const float *edges, const int64_t *idx, int B, int N, int K, int C, float *out, hipStream_t st);

// tmpnn_layers.hip
int launch_msg(bool dec, const float *W1e, int ld1, const float *W2, const float *b2, const float *P,
               const float *hE, const int32_t *E_idx, const float *mask, int64_t T, float *Ssum, float *cnt, hipStream_t st);
// P [T,256]: P[t, 0:128] = Wa h_t + ba, P[t, 128:256] = Wc h_t (+ add_tab[add_idx[t]] when add_tab is set: the decoder's
// sequence term W1[:, 256:384] W_s[S_t], which rides with the neighbour's projection)
struct NodeProj { const float *Wa; int lda; const float *ba; const float *Wc; int ldc; float *P; const float *add_tab; const int32_t *add_idx; };
// kernel-side argument blocks of node_update (tmpnn_layers.hip: fp32 MFMA; tmpnn_split.hip: f16x2)
struct ProjSpec { const float *Wa; int lda; const float *ba; const float *Wc; int ldc; float *P; const float *add_tab; const int32_t *add_idx; };
struct NodeArgs {
    const float *W3, *b3, *n1w, *n1b, *Win, *bin, *Wout, *bout, *n2w, *n2b;
    const float *h_in, *Ssum, *cnt, *mask;
    float *h_out;
    int T;
    ProjSpec proj[2];       // proj[k].P == nullptr -> not requested
    const char *img[13];    // fragment images of the 13 GEMM units (W3, W_in/W_out chunk pairs, projections) or all null
};
struct HeadArgs;
// head != nullptr (small launches, no projections requested): the ddG head of the same rows MAY run in the same launch (node_head_fused_kernel);
// *head_ran says whether it did — the caller launches the head itself otherwise
int launch_node_update_split(const NodeArgs &a, int64_t T, hipStream_t st, const HeadArgs *head = nullptr, bool *head_ran = nullptr);
bool node_head_fusable(int mode, int64_t T);
int launch_node_proj(const float *h, const NodeProj &np, int64_t T, hipStream_t st);
int launch_node_update(const float *W3, const float *b3, const float *n1w, const float *n1b, const float *Win,
                       const float *bin, const float *Wout, const float *bout, const float *n2w, const float *n2b,
                       const float *h_in, const float *Ssum, const float *cnt, const float *mask, int64_t T,
                       float *h_out, const NodeProj *p0, const NodeProj *p1, hipStream_t st, const HeadArgs *head = nullptr,
                       bool *head_ran = nullptr);
int launch_enc_edge(const EncW &e, const float *P, float *hE, const int32_t *E_idx, int64_t T, hipStream_t st);

// tmpnn_head.hip
int launch_head(const tmpnn_weights *w, const float *hA, const float *hB, const int32_t *S, int64_t T, float *ddg,
                float *z_opt, int32_t *status, hipStream_t st, const int32_t *maxlen_probe = nullptr);
HeadArgs tm_head_args(const tmpnn_weights *w, const float *hA, const float *hB, const int32_t *S, int64_t T, float *ddg, float *z_opt,
                      int32_t *status, const int32_t *maxlen_probe);
int launch_log_probs(const tmpnn_weights *w, const float *h, int64_t T, float *out, int32_t *status, hipStream_t st,
                     const int32_t *maxlen_probe = nullptr);
int launch_seq_embed(const tmpnn_weights *w, const int32_t *S, int64_t T, float *hS, hipStream_t st);
int launch_head_generic(const float *const *hidden, int n_final, const float *Ws, const int32_t *S, int64_t T, const float *conv_w,
                        const float *conv_b, int n_layers, const float *const *mlp_w, const float *const *mlp_b, const int32_t *dims,
                        const float *ddg_w, const float *ddg_b, float *ddg, float *z_opt, float *buf0, float *buf1, int32_t *status,
                        hipStream_t st);
int launch_range_check(const float *x, int64_t n, int32_t *status, hipStream_t st, const int32_t *maxlen_probe = nullptr,
                       int64_t T = 0);   // ORs TMPNN_STATUS_RANGE if any x is inf / NaN (+ the MAXLEN probe of the fused forward)
int launch_prep_tables(tmpnn_weights *w, hipStream_t st);

int launch_clock_probe(int blocks, int iters, unsigned long long *out, float *sink, hipStream_t st);
int launch_clock_monitor(int iters, unsigned long long *out, hipStream_t st);
// tmpnn_split.hip (mode = TM_MM_F16X2 | TM_MM_BF16X3)
int launch_enc_edge_split(int mode, const EncW &e, const float *P, float *hE, const int32_t *E_idx, int64_t T, hipStream_t st);
int launch_msg_split(int mode, bool dec, const float *W1e, int ld1, const float *W2, const float *b2, const float *P,
                     const float *hE, const int32_t *E_idx, const float *mask, int64_t T, float *Ssum, float *cnt, hipStream_t st);
int launch_gemm_probe(int mode, const float *X, const float *W, float *Y, int64_t T, int reps, hipStream_t st);
// small launches (one tile per workgroup): edge update of layer l + message pass of the next layer as one launch (bit-identical)
bool edge_msg_fusable(int mode, int64_t T);
int launch_edge_msg_fused(const EncW &e, const float *P_edge, float *hE, const int32_t *E_idx, bool dec, const float *W1e, int ld1,
                          const float *W2, const float *b2, const float *P_msg, const float *mask, int64_t T, float *Ssum, float *cnt,
                          hipStream_t st);
int launch_selftest(int32_t *status, hipStream_t st);
// tmpnn_edge_wave.hip: f16x2 edge update of large launches, one wavefront per 16-row block (same bits as the 8-wavefront forms)
struct EdgeArgsB;
bool enc_edge_wave_wanted(int64_t T);
int launch_enc_edge_wave(const EdgeArgsB &a, int64_t T, hipStream_t st);

int tm_num_cus();
// Kernel-form switches (TMPNN_KNN_REG, TMPNN_NODE_DEEP, TMPNN_FEAT_SPLIT ...) and the per-phase timers (TMPNN_*_PROF, which
// hipMalloc a scratch buffer, copy it back with a blocking hipMemcpy and print) exist ONLY in the debug variant of the library
//   python -m thermompnn_amd.build --variant debug -DTMPNN_DEBUG_BUILD      (-> libtmpnn_debug.so; select with TMPNN_LIB)
// The shipped libtmpnn.so picks every kernel form from the launch size and the handle's precision alone: its launchers never
// read the environment, never allocate device memory and never synchronise (the contract of include/tmpnn.h).
#ifdef TMPNN_DEBUG_BUILD
#include <stdlib.h>
#define TM_DBG_FLAG(name, dflt) ([] { const char *e_ = getenv(name); return e_ ? e_[0] != '0' : (bool)(dflt); }())
#define TM_DBG_INT(name, dflt) ([] { const char *e_ = getenv(name); return e_ ? atoi(e_) : (int)(dflt); }())
#else
#define TM_DBG_FLAG(name, dflt) ((bool)(dflt))
#define TM_DBG_INT(name, dflt) ((int)(dflt))
#endif
// matrix-core path of the per-edge GEMMs (tmpnn_split.h): a property of the weight handle (tmpnn_weights_create_p);
// TMPNN_PRECISION = f16x2 (default) | bf16x3 | fp32 only picks the default of handles created without one.
enum { TM_MM_FP32 = 0, TM_MM_BF16X3 = 1, TM_MM_F16X2 = 2 };
int tm_matmul_mode();                 // mode of the API call in progress on this thread (set from the handle)
const tmpnn_weights *tm_cur_weights();   // handle of the API call in progress (nullptr outside one)
struct TmModeScope {                  // entry points that take a handle open one of these
    int saved;
    const tmpnn_weights *saved_w;
    explicit TmModeScope(const tmpnn_weights *w);
    ~TmModeScope();
};
int launch_prep_wimg(const float *W, int ld, char *dst, hipStream_t st, int n_rows = 128, int k_valid = 128, int k_wrap = 0, bool perm = false);       // tmpnn_split.hip
const char *tm_find_wimg(const float *base);                                   // nullptr if no image (or no handle in scope)
const char *tm_find_wimgp(const float *base);                                  // ... the K-permuted image of a full 128 x 128 block
// Non-finite tests under -fno-honor-nans. The kernels are built with relaxed NaN semantics, so hipcc may fold a NaN test
// on the RESULT of floating-point arithmetic (measured: both the sum test and the exponent-bit test on a computed value
// were compiled away; only the inf half survives). Tests are therefore made on raw bits LOADED FROM MEMORY, before any
// arithmetic touches them: tm_nonfinite_bits on integer loads of the same addresses.
__device__ __forceinline__ bool tm_nonfinite_bits(unsigned b) { return (b & 0x7f800000u) == 0x7f800000u; }
__device__ __forceinline__ bool tm_nonfinite(float x) { return tm_nonfinite_bits(__float_as_uint(x)); }   // inf only is guaranteed
// |x| >= 65504 (the largest fp16), inf and NaN included: a value the f16x2 split cannot carry. On raw bits, like the above.
__device__ __forceinline__ bool tm_f16_range_bits(unsigned b) { return (b & 0x7fffffffu) >= 0x477fe000u; }
// The same test on a COMPUTED value: the value is laundered through an empty asm first, so that hipcc (-fno-honor-nans) cannot
// reason "the result of fp arithmetic is never NaN" and fold the NaN half of the integer comparison away.
__device__ __forceinline__ bool tm_f16_range_computed(float x) {
    asm volatile("" : "+v"(x));
    return tm_f16_range_bits(__float_as_uint(x));
}
