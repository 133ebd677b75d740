// k-NN graph construction, the fused edge featurizer and the standalone gathers (gfx950).
//
// Reference semantics: ProteinFeatures (/root/reference/protein_mpnn_utils.py:1084-1180),
// PositionalEncodings (:896-908), gather_edges (:763-767), gather_nodes (:770-778).
// Nothing here materialises an L x L matrix: distances are computed per row (kNN) or per edge (RBF).
#include <stdio.h>
#include <stdlib.h>

#include "tmpnn_common.h"
#include "tmpnn_internal.h"
#include "tmpnn_split.h"

// ------------------------------------------------------------------------------------------------
// knn_topk: one wavefront per residue row. The row's adjusted distances live in LDS; selection is K rounds of a
// wavefront-wide 32-bit min over per-lane cached minima (distance bit patterns; a ballot resolves the owner lane, a
// second min only on exact ties), after which the owner lane retires the element and rescans its stripe.
//   D = m_i m_j sqrt(|Ca_i - Ca_j|^2 + 1e-6);  D_adj = D + (1 - m_i m_j) max_j D      (:1101-1106)
// ------------------------------------------------------------------------------------------------
// Wavefront-wide unsigned min through DPP (no LDS crossbar traffic, ~8 cycles per step instead of a ds_bpermute round
// trip): prefix-min inside each row of 16 lanes (row_shr 1/2/4/8; min is idempotent, so overlapping windows are fine),
// then row_bcast:15 / row_bcast:31 carry the row results to lane 63, which is read back as a wavefront-uniform value.
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#define TM_DPP_MIN(ctrl, row_mask)                                                                           \
    {                                                                                                        \
        const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, ctrl, row_mask, 0xf, false); \
        v = o < v ? o : v;                                                                                   \
    }
    TM_DPP_MIN(0x111, 0xf)   // row_shr:1
    TM_DPP_MIN(0x112, 0xf)   // row_shr:2
    TM_DPP_MIN(0x114, 0xf)   // row_shr:4
    TM_DPP_MIN(0x118, 0xf)   // row_shr:8   -> lane 15 of every row holds the row minimum
    TM_DPP_MIN(0x142, 0xa)   // row_bcast:15 into rows 1 and 3
    TM_DPP_MIN(0x143, 0xc)   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wavefront minimum
#undef TM_DPP_MIN
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }

// One row of the k-NN graph by one wavefront. LONG rows (L > 512) pad the LDS row by one float per 64 elements (knn_slot)
// so that a lane's stripe (j = lane, lane + 64, ...) sits in consecutive banks, and re-scan the retired element's stripe
// with the WHOLE wavefront (one conflict-free read, one DPP min); short rows let the owner lane re-scan its <= 8 elements.
template <bool LONG>
__device__ __forceinline__ void knn_row(float *d, const float *__restrict__ X, const float *__restrict__ mask, int i, int s,
                                        int L, int Keff, int lane, int32_t *__restrict__ E_idx, float *__restrict__ D_nb) {
    auto slot = [](int j) { return LONG ? j + (j >> 6) : j; };
    const float xi = X[(size_t)i * 12 + 3], yi = X[(size_t)i * 12 + 4], zi = X[(size_t)i * 12 + 5];
    const float mi = mask[i];

    float dmax = 0.f;
#pragma unroll 4
    for (int j = lane; j < L; j += 64) {
        const float *c = X + (size_t)(s + j) * 12 + 3;
        const float dx = c[0] - xi, dy = c[1] - yi, dz = c[2] - zi;
        const float s2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        const float D = __fmul_rn(mi * mask[s + j], sqrtf(__fadd_rn(s2, 1e-6f)));
        d[slot(j)] = D;
        dmax = fmaxf(dmax, D);
    }
    dmax = wave_max_f32(dmax);
    // per-lane running minimum over its stripe (j = lane, lane + 64, ...): (distance bits, index), lowest index on ties.
    // Distances are >= +0, so their bit patterns order like the values and a 32-bit unsigned min suffices.
    unsigned best_d = 0xffffffffu, best_j = 0xffffffffu;
#pragma unroll 4
    for (int j = lane; j < L; j += 64) {
        const float m2 = mi * mask[s + j];
        const float Da = __fadd_rn(d[slot(j)], __fmul_rn(1.0f - m2, dmax));
        d[slot(j)] = Da;
        const unsigned bits = __float_as_uint(Da);
        if (bits < best_d) { best_d = bits; best_j = (unsigned)j; }
    }
    wave_lds_fence();
    const int stripe_len = (L + 63) >> 6;
    int out_j = -1;                                              // lane t keeps the t-th neighbour: one coalesced store per row
    float out_d = 0.f;
    for (int t = 0; t < Keff; ++t) {
        const unsigned g = wave_min_u32(best_d);
        const unsigned long long tied = __ballot(best_d == g);
        unsigned j;
        if (__popcll(tied) == 1) {                               // the common case: one lane holds the minimum
            j = __builtin_amdgcn_readlane(best_j, (int)__ffsll((long long)tied) - 1);
        } else {                                                 // exact tie between lanes: lowest index wins
            j = wave_min_u32(best_d == g ? best_j : 0xffffffffu);
        }
        if (lane == t) {
            out_j = s + (int)j;
            out_d = __uint_as_float(g);
        }
        const unsigned o = j & 63u;                              // owner lane of j's stripe
        if (!LONG) {
            if (lane == (int)o) {                                // owner lane retires j and rescans its stripe
                d[j] = __uint_as_float(0x7f800000u);
                best_d = 0xffffffffu;
                best_j = 0xffffffffu;
#pragma unroll 2
                for (int jj = lane; jj < L; jj += 64) {
                    const unsigned bits = __float_as_uint(d[jj]);
                    if (bits != 0x7f800000u && bits < best_d) { best_d = bits; best_j = (unsigned)jj; }
                }
            }
        } else {
            if (lane == 0) d[slot((int)j)] = __uint_as_float(0x7f800000u);
            wave_lds_fence();
            unsigned nb = 0xffffffffu, nj = 0xffffffffu;
            for (int k0 = 0; k0 < stripe_len; k0 += 64) {        // one pass for L <= 4096
                const int jj = (int)o + 64 * (k0 + lane);
                unsigned bits = 0xffffffffu;
                if (k0 + lane < stripe_len && jj < L) {
                    bits = __float_as_uint(d[slot(jj)]);
                    if (bits == 0x7f800000u) bits = 0xffffffffu; // retired
                }
                const unsigned m = wave_min_u32(bits);
                if (m < nb) {                                    // wavefront-uniform
                    nb = m;
                    const unsigned long long hit = __ballot(bits == m);
                    nj = o + 64u * (unsigned)(k0 + (int)__ffsll((long long)hit) - 1);
                }
            }
            if (lane == (int)o) {
                best_d = nb;
                best_j = nb == 0xffffffffu ? 0xffffffffu : nj;
            }
        }
    }
    if (lane < TM_KS) {                                          // slots >= Keff keep (-1, 0)
        E_idx[(size_t)i * TM_KS + lane] = out_j;
        D_nb[(size_t)i * TM_KS + lane] = out_d;
    }
    wave_lds_fence();
}

// Rows of at most 64 NS residues (NS = 4 or 8): the whole row lives in registers, NS candidates per lane (j = lane + 64 k).
// Each lane sorts its own stripe once (odd-even transposition with a strict compare: stable, so equal distances keep their
// index order), after which a selection round is a wavefront min over the lanes' heads and a register shift in the owner
// lane — no LDS, no divergent rescan. Same arithmetic, same (distance, index) order as knn_row.
template <int NS>
__device__ __forceinline__ void knn_row_reg(const float *__restrict__ X, const float *__restrict__ mask, int i, int s, int L,
                                            int Keff, int lane, int32_t *__restrict__ E_idx, float *__restrict__ D_nb) {
    const float xi = X[(size_t)i * 12 + 3], yi = X[(size_t)i * 12 + 4], zi = X[(size_t)i * 12 + 5];
    const float mi = mask[i];
    float D[NS], m2[NS];
    float dmax = 0.f;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int j = lane + 64 * k, jc = j < L ? j : 0;        // (a valid row for the loads of the empty slots)
        const float *c = X + (size_t)(s + jc) * 12 + 3;
        const float dx = c[0] - xi, dy = c[1] - yi, dz = c[2] - zi;
        const float s2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        m2[k] = mi * mask[s + jc];
        D[k] = __fmul_rn(m2[k], sqrtf(__fadd_rn(s2, 1e-6f)));
        if (j < L) dmax = fmaxf(dmax, D[k]);
    }
    dmax = wave_max_f32(dmax);
    unsigned v[NS], id[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const float Da = __fadd_rn(D[k], __fmul_rn(1.0f - m2[k], dmax));
        v[k] = lane + 64 * k < L ? __float_as_uint(Da) : 0xffffffffu;    // distances are >= +0: their bits order like the values
        id[k] = (unsigned)k;
    }
#pragma unroll
    for (int r = 0; r < NS; ++r)
#pragma unroll
        for (int k = r & 1; k + 1 < NS; k += 2) {
            const bool sw = v[k] > v[k + 1];
            const unsigned a = v[k], b = v[k + 1], ia = id[k], ib = id[k + 1];
            v[k] = sw ? b : a;
            v[k + 1] = sw ? a : b;
            id[k] = sw ? ib : ia;
            id[k + 1] = sw ? ia : ib;
        }
    int out_j = -1;                                              // lane t keeps the t-th neighbour: one coalesced store per row
    float out_d = 0.f;
    for (int t = 0; t < Keff; ++t) {
        const unsigned g = wave_min_u32(v[0]);
        const unsigned long long tied = __ballot(v[0] == g);
        unsigned j;
        if (__popcll(tied) == 1) {                               // the common case: one lane holds the minimum
            const int o = (int)__ffsll((long long)tied) - 1;
            j = ((unsigned)__builtin_amdgcn_readlane((int)id[0], o) << 6) | (unsigned)o;
        } else {                                                 // exact tie between lanes: lowest index wins
            j = wave_min_u32(v[0] == g ? (id[0] << 6) | (unsigned)lane : 0xffffffffu);
        }
        if (lane == t) {
            out_j = s + (int)j;
            out_d = __uint_as_float(g);
        }
        const bool mine = lane == (int)(j & 63u);                // the owner lane drops its head
#pragma unroll
        for (int k = 0; k + 1 < NS; ++k) {
            v[k] = mine ? v[k + 1] : v[k];
            id[k] = mine ? id[k + 1] : id[k];
        }
        v[NS - 1] = mine ? 0xffffffffu : v[NS - 1];
    }
    if (lane < TM_KS) {                                          // slots >= Keff keep (-1, 0)
        E_idx[(size_t)i * TM_KS + lane] = out_j;
        D_nb[(size_t)i * TM_KS + lane] = out_d;
    }
}

// Threshold form of the register rows (round 3; default for NS = 4 / 8): instead of Keff dependent extract-min rounds (~30
// instructions each) the wavefront
//   1. finds a distance t with Keff <= #{candidates <= t} <= 64 — counting is 1 v_cmp + 1 s_bcnt1 per candidate register, every
//      value wave-uniform; the search is a regula falsi on r^3 (neighbour counts grow like a volume) that falls back to bisecting
//      the bit patterns every other step: 2-4 probes for a protein-like row;
//   2. compacts those <= 64 candidates to one per lane (v_mbcnt of the ballots gives each its slot; 512 B of LDS per wavefront);
//   3. sorts the 64 (distance bits, index) pairs with a bitonic network over the lanes (21 compare-exchange stages; partners
//      fetched through the LDS crossbar, the min / max choice of a stage is a compile-time lane mask XOR-ed into the compare mask
//      on the scalar unit), and lane t keeps the t-th neighbour.
// Keys are (distance bits, index) exactly as in knn_row / knn_row_reg, so the neighbour list is the same list in the same order,
// ties included. Rows where no such t exists (an exact tie straddling the 48th..64th place: masked-out rows, fewer than Keff
// unmasked candidates) take the extract-min form.
template <int J>
__device__ __forceinline__ unsigned knn_xor_lane(unsigned x, int lane) {
    return (unsigned)__builtin_amdgcn_ds_bpermute((lane ^ J) << 2, (int)x);
}
template <int K2, int J>
__device__ __forceinline__ void knn_bitonic_stage(unsigned &d, unsigned &j, int lane) {
    // lanes that keep the LARGER of the pair: (lane & K2 != 0) xor (lane & J != 0)   (K2 = 64: ascending everywhere)
    constexpr unsigned long long bitK = K2 >= 64 ? 0ull : ~0ull / ((1ull << (K2 & 63)) + 1ull) << (K2 & 63);           // lanes with bit log2(K2) set
    constexpr unsigned long long bitJ = ~0ull / ((1ull << J) + 1ull) << J;                                  // lanes with bit log2(J) set
    constexpr unsigned long long keep_max = bitK ^ bitJ;
    const unsigned pd = knn_xor_lane<J>(d, lane), pj = knn_xor_lane<J>(j, lane);
    const unsigned long long less = __ballot(pd < d) | (__ballot(pd == d) & __ballot(pj < j));              // partner's key is smaller
    const unsigned long long take = less ^ keep_max;          // keep-min lanes take a smaller partner, keep-max lanes a larger one (keys are distinct)
    unsigned nd, nj;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(nd) : "v"(d), "v"(pd), "s"(take));
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(nj) : "v"(j), "v"(pj), "s"(take));
    d = nd;
    j = nj;
}
template <int NS>
__device__ __forceinline__ bool knn_row_sel(unsigned (*sel)[2], const float *__restrict__ X, const float *__restrict__ mask, int i, int s, int L,
                                            int Keff, int lane, int32_t *__restrict__ E_idx, float *__restrict__ D_nb) {
    const float xi = X[(size_t)i * 12 + 3], yi = X[(size_t)i * 12 + 4], zi = X[(size_t)i * 12 + 5];
    const float mi = mask[i];
    float D[NS], m2[NS];
    float dmax = 0.f;
#pragma unroll
    for (int k = 0; k < NS; ++k) {                              // (identical arithmetic to knn_row_reg / knn_row)
        const int j = lane + 64 * k, jc = j < L ? j : 0;
        const float *c = X + (size_t)(s + jc) * 12 + 3;
        const float dx = c[0] - xi, dy = c[1] - yi, dz = c[2] - zi;
        const float s2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        m2[k] = mi * mask[s + jc];
        D[k] = __fmul_rn(m2[k], sqrtf(__fadd_rn(s2, 1e-6f)));
        if (j < L) dmax = fmaxf(dmax, D[k]);
    }
    dmax = wave_max_f32(dmax);
    unsigned v[NS];
    unsigned vmax = 0u;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const float Da = __fadd_rn(D[k], __fmul_rn(1.0f - m2[k], dmax));
        const bool ok = lane + 64 * k < L;
        v[k] = ok ? __float_as_uint(Da) : 0xffffffffu;           // distances are >= +0: their bits order like the values
        if (ok) vmax = v[k] > vmax ? v[k] : vmax;
    }
    auto count = [&](unsigned tb) {                              // #{candidates <= tb}: wave-uniform
        int c = 0;
#pragma unroll
        for (int k = 0; k < NS; ++k) c += (int)__popcll(__ballot(v[k] <= tb));
        return c;
    };
    unsigned tb;
    if (L <= 64) {
        tb = 0xfffffffeu;                                        // every real candidate (the empty slots hold 0xffffffff)
    } else {
        vmax = __float_as_uint(wave_max_f32(__uint_as_float(vmax)));           // (bit patterns of non-negative floats order like the floats)
        unsigned lo = 0u, hi = vmax;                             // count(lo - 1) < Keff (nothing below +0), count(hi) = L > 64
        int clo = 0, chi = L;
        const float aim = 0.5f * (float)(Keff + 64);
        bool found = false;
        for (int it = 0; it < 14; ++it) {
            if (hi - lo <= 1u) break;                            // no pattern strictly between: an exact tie straddles the window
            unsigned t;
            if (it & 1) {
                t = lo + ((hi - lo) >> 1);                       // bisection step (bit patterns)
            } else {                                             // regula falsi on r^3
                const float rl = __uint_as_float(lo), rh = __uint_as_float(hi);
                const float x = (aim - (float)clo) / (float)(chi - clo);
                const float c3 = rl * rl * rl + x * (rh * rh * rh - rl * rl * rl);
                const float r = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(fmaxf(c3, 1e-30f)) * (1.0f / 3.0f));
                t = __float_as_uint(r);
                if (!(t > lo && t < hi)) t = lo + ((hi - lo) >> 1);
            }
            const int c = count(t);
            if (c >= Keff && c <= 64) { tb = t; found = true; break; }
            if (c < Keff) { lo = t; clo = c; } else { hi = t; chi = c; }
        }
        if (!found) return false;
    }
    // compaction: candidate (lane, k) with v <= tb goes to slot (#selected before it in k-major, lane-minor order)
    int base = 0;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const bool in = v[k] <= tb;
        const unsigned long long mk = __ballot(in);
        const int pos = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, (unsigned)base));
        if (in) { sel[pos][0] = v[k]; sel[pos][1] = (unsigned)(lane + 64 * k); }
        base += (int)__popcll(mk);
    }
    wave_lds_fence();
    unsigned d = lane < base ? sel[lane][0] : 0xffffffffu, j = lane < base ? sel[lane][1] : 0xffffffffu;
    wave_lds_fence();
    knn_bitonic_stage<2, 1>(d, j, lane);
    knn_bitonic_stage<4, 2>(d, j, lane);  knn_bitonic_stage<4, 1>(d, j, lane);
    knn_bitonic_stage<8, 4>(d, j, lane);  knn_bitonic_stage<8, 2>(d, j, lane);  knn_bitonic_stage<8, 1>(d, j, lane);
    knn_bitonic_stage<16, 8>(d, j, lane); knn_bitonic_stage<16, 4>(d, j, lane); knn_bitonic_stage<16, 2>(d, j, lane);
    knn_bitonic_stage<16, 1>(d, j, lane);
    knn_bitonic_stage<32, 16>(d, j, lane); knn_bitonic_stage<32, 8>(d, j, lane); knn_bitonic_stage<32, 4>(d, j, lane);
    knn_bitonic_stage<32, 2>(d, j, lane);  knn_bitonic_stage<32, 1>(d, j, lane);
    knn_bitonic_stage<64, 32>(d, j, lane); knn_bitonic_stage<64, 16>(d, j, lane); knn_bitonic_stage<64, 8>(d, j, lane);
    knn_bitonic_stage<64, 4>(d, j, lane);  knn_bitonic_stage<64, 2>(d, j, lane);  knn_bitonic_stage<64, 1>(d, j, lane);
    if (lane < TM_KS) {                                          // slots >= Keff keep (-1, 0)
        E_idx[(size_t)i * TM_KS + lane] = lane < Keff ? s + (int)j : -1;
        D_nb[(size_t)i * TM_KS + lane] = lane < Keff ? __uint_as_float(d) : 0.f;
    }
    return true;
}

// One residue's neighbour row by ONE wavefront (the body of knn_kernel's loop; the small-launch featurizer calls it too, so both
// produce the same list, bit for bit): the zero state + first projection of the fused forward, the protein bounds, the max_len
// guard, then the row in registers (NS candidates per lane) or through the LDS row `d` (NS = 0).
template <int NS>
__device__ __forceinline__ void knn_residue(float *d, unsigned (*sel)[2], const float *__restrict__ X, const float *__restrict__ mask,
                                            const int32_t *__restrict__ offsets, int N, int max_len, int K, int32_t *__restrict__ E_idx,
                                            float *__restrict__ D_nb, int32_t *__restrict__ status, const KnnInit &init, int sel_rows,
                                            int i, int lane) {
    if (init.hV0) {      // the fused forward: this residue's all-zero initial state and its projection (W . 0 + b = b exactly)
        const f4 z = f4{0.f, 0.f, 0.f, 0.f};
        if (lane < 32) st4(init.hV0 + (size_t)i * TM_H + 4 * lane, z);
        st4(init.P + (size_t)i * 256 + 4 * lane, lane < 32 ? ld4(init.ba + 4 * lane) : z);
    }
    int lo = 0, hi = N;                      // protein p with offsets[p] <= i < offsets[p+1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offsets[mid] <= i) lo = mid; else hi = mid;
    }
    const int s = offsets[lo], L = offsets[lo + 1] - s;
    if (L > max_len) {      // the caller's max_len sized this wavefront's LDS row: a longer protein would overrun it
        if (lane < TM_KS) { E_idx[(size_t)i * TM_KS + lane] = -1; D_nb[(size_t)i * TM_KS + lane] = 0.f; }
        if (lane == 0 && status) atomicOr(status, TMPNN_STATUS_MAXLEN);
        return;
    }
    const int Keff = K < L ? K : L;
    if constexpr (NS > 0) {
        if (!sel_rows || !knn_row_sel<NS>(sel, X, mask, i, s, L, Keff, lane, E_idx, D_nb))
            knn_row_reg<NS>(X, mask, i, s, L, Keff, lane, E_idx, D_nb);
    } else if (L > 512) knn_row<true>(d, X, mask, i, s, L, Keff, lane, E_idx, D_nb);
    else knn_row<false>(d, X, mask, i, s, L, Keff, lane, E_idx, D_nb);
}

// NS = 0: rows of any length through LDS (knn_row); NS = 4 / 8: every row of the batch has at most 256 / 512 residues
// (max_len says so) and runs in registers (knn_row_reg) — no dynamic LDS at all.
template <int NS>
__global__ __launch_bounds__(TM_THREADS, NS == 8 ? 4 : 8) void knn_kernel(const float *__restrict__ X, const float *__restrict__ mask,
                                                            const int32_t *__restrict__ offsets, int N, int T, int max_len,
                                                            int K, int32_t *__restrict__ E_idx, float *__restrict__ D_nb,
                                                            int32_t *__restrict__ status, KnnInit init, int sel_rows) {
    extern __shared__ __attribute__((aligned(16))) float knn_lds[];
    __shared__ unsigned s_sel[NS > 0 ? 4 : 1][64][2];        // compaction slots of knn_row_sel, one set per wavefront
    const int lane = tm_tid() & 63, wv = tm_wave(tm_tid());
    float *d = knn_lds + (size_t)wv * (max_len + (max_len >> 6) + 1);
    if (init.status_zero && tm_bid() == 0 && tm_tid() == 0) *init.status_zero = 0;   // (nothing in this launch ORs into it: status == nullptr)

    for (int i = tm_bid() * 4 + wv; i < T; i += tm_nblk() * 4)
        knn_residue<NS>(d, s_sel[wv], X, mask, offsets, N, max_len, K, E_idx, D_nb, status, init, sel_rows, i, lane);
}

// ------------------------------------------------------------------------------------------------
// edge_featurize: one residue (48 edge slots) per workgroup iteration.
//   25 atom-pair distances -> 400 RBFs in LDS -> [48x400]x[400x128] on the matrix cores (+ folded
//   positional table) -> LayerNorm -> W_e -> h_E.   W_edge's RBF columns live in VGPRs (200/lane).
// ------------------------------------------------------------------------------------------------
#define RBF_RS 448   // padded row length (floats) of the swizzled RBF tile: 100 chunks used of 112

__constant__ int c_pair_a[25] = {1, 0, 2, 3, 4, 1, 1, 1, 1, 0, 0, 0, 4, 4, 3, 0, 2, 3, 4, 2, 3, 4, 2, 3, 2};
__constant__ int c_pair_b[25] = {1, 0, 2, 3, 4, 0, 2, 3, 4, 2, 3, 4, 2, 3, 2, 1, 1, 1, 1, 0, 0, 0, 4, 4, 3};

struct FeatArgs {
    const float *edge_w;     // [128,416]
    const float *pos_table;  // [66,128]
    const float *pos_w, *pos_b;   // features.embeddings.linear [16,66], [16]
    const float *ln_w, *ln_b, *We_w, *We_b;
    const float *X;          // [T,4,3]
    const int32_t *ridx, *cenc, *E_idx;
    const float *D_nb;
    float *hE, *E_opt;
    int T;
    float mu[16];            // torch.linspace(2, 22, 16)
    const char *img_e[4];    // fragment images of edge_embedding.weight[:, 16:416] (four 128-column blocks, the last zero-padded) ...
    const char *img_we;      // ... and of W_e (f16x2 handles; null otherwise)
};

__device__ __forceinline__ void atoms5(const float *__restrict__ x, float *out /*[15]*/) {
    float n[3], ca[3], c[3], o[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { n[k] = x[k]; ca[k] = x[3 + k]; c[k] = x[6 + k]; o[k] = x[9 + k]; }
    float b[3], cc[3], a[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { b[k] = ca[k] - n[k]; cc[k] = c[k] - ca[k]; }
    a[0] = b[1] * cc[2] - b[2] * cc[1];
    a[1] = b[2] * cc[0] - b[0] * cc[2];
    a[2] = b[0] * cc[1] - b[1] * cc[0];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        out[k] = n[k]; out[3 + k] = ca[k]; out[6 + k] = c[k]; out[9 + k] = o[k];
        out[12 + k] = -0.58273431f * a[k] + 0.56802827f * b[k] - 0.54067466f * cc[k] + ca[k];   // virtual Cb (:1134)
    }
}

// NW wavefronts per workgroup (1 workgroup per CU): NW = 8 puts two wavefronts on every SIMD, each owning ONE
// 16-column block (132 weight VGPRs). The matrix-pipe work per SIMD is unchanged, but the serial phases in front of
// the GEMM (atom gather -> 1200 distances -> 19200 Gaussians) and the LayerNorm / store phases run on twice the
// threads with twice the latency hiding — they were 50 % of this kernel's time with 4 wavefronts.
template <int NW>
__global__ __launch_bounds__(64 * NW, (NW == 8 ? 2 : 1)) void featurize_kernel(FeatArgs a) {
    constexpr int NT = 64 * NW, NCB = 8 / NW, RPW = TM_TILE / NW;    // threads, column blocks / wave, rows / wave
    __shared__ __attribute__((aligned(16))) float rbf[TM_TILE * RBF_RS];
    __shared__ __attribute__((aligned(16))) float tA[TM_TILE * TM_H];
    __shared__ __attribute__((aligned(16))) float tB[TM_TILE * TM_H];
    __shared__ float s_atoms[TM_TILE][16];
    __shared__ float s_self[16];
    __shared__ float s_dist[TM_TILE][28];
    __shared__ int s_idx[TM_TILE];
    __shared__ int s_dpos[TM_TILE];
    const int tid = tm_tid(), lane = tid & 63, wv = tid >> 6, m = lane & 15, q = lane >> 4;
    const int c32 = lane & 31;
    const int col0 = (TM_H / NW) * wv, chunk0 = (32 / NW) * wv;

    float wedge[NCB][100], we[NCB][32];
    f4 be[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const int n0 = col0 + 16 * cb;
        load_wfrag<25>(a.edge_w, 416, n0, 16, TM_H, wedge[cb], lane);
        load_wfrag<8>(a.We_w, TM_H, n0, 0, TM_H, we[cb], lane);
        be[cb] = ld4(a.We_b + n0 + 4 * q);
    }
    const f4 g4 = ld4(a.ln_w + 4 * c32), b4 = ld4(a.ln_b + 4 * c32);

    const TileRange tr = xcd_tile_range(a.T);
    for (int i = tr.begin; i < tr.end; i += tr.step) {
        if (tid < TM_TILE) {
            const int j = a.E_idx[(size_t)i * TM_KS + tid];
            s_idx[tid] = j;
            const int jj = j < 0 ? i : j;
            float at[15];
            atoms5(a.X + (size_t)jj * 12, at);
#pragma unroll
            for (int k = 0; k < 15; ++k) s_atoms[tid][k] = at[k];
            // PositionalEncodings index (:903-905, :1170-1175)
            const int off = a.ridx[i] - a.ridx[jj];
            const int same = a.cenc[i] == a.cenc[jj];
            s_dpos[tid] = same ? min(max(off + 32, 0), 64) : 65;
        } else if (tid == 64) {
            float at[15];
            atoms5(a.X + (size_t)i * 12, at);
#pragma unroll
            for (int k = 0; k < 15; ++k) s_self[k] = at[k];
        }
        __syncthreads();
        for (int e = tid; e < TM_TILE * 25; e += NT) {
            const int mm = e / 25, p = e - mm * 25;
            float D;
            if (p == 0) {
                D = a.D_nb[(size_t)i * TM_KS + mm];            // masked Ca-Ca distance from _dist (:1142)
            } else {
                const float *A = s_self + 3 * c_pair_a[p];
                const float *B = s_atoms[mm] + 3 * c_pair_b[p];
                const float dx = A[0] - B[0], dy = A[1] - B[1], dz = A[2] - B[2];
                D = sqrtf(dx * dx + dy * dy + dz * dz + 1e-6f);  // _get_rbf (:1122)
            }
            s_dist[mm][p] = D;
        }
        __syncthreads();
        for (int e = tid; e < TM_TILE * 100; e += NT) {          // 16 Gaussians per pair, 4 per thread (:1111-1119)
            const int mm = e / 100, c = e - mm * 100;
            const float D = s_dist[mm][c >> 2];
            const int r0 = (c & 3) * 4;
            f4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = (D - a.mu[r0 + r]) * 0.8f;         // / sigma, sigma = 1.25
                v[r] = __expf(-(t * t));
            }
            st4(rbf + chunk_off<RBF_RS>(mm, c), v);
        }
        __syncthreads();

        f4 acc[3][NCB];
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {
            const float *pt = a.pos_table + s_dpos[16 * rb + m] * TM_H + col0 + 4 * q;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = ld4(pt + 16 * cb);
        }
        mma_tile<25, NCB, RBF_RS>(rbf, wedge, acc, lane);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) st4(tA + chunk_off(16 * rb + m, chunk0 + 4 * cb + q), acc[rb][cb]);
        __syncthreads();
#pragma unroll
        for (int it = 0; it < RPW / 2; ++it) {                   // norm_edges (:1179)
            const int row = RPW * wv + 2 * it + (lane >> 5);
            float *p = tA + chunk_off(row, c32);
            const f4 y = layer_norm_row(ld4(p), g4, b4);
            st4(p, y);
            if (a.E_opt) {
                const bool ok = s_idx[row] >= 0;
                st4(a.E_opt + ((size_t)i * TM_KS + row) * TM_H + 4 * c32, ok ? y : f4{0.f, 0.f, 0.f, 0.f});
            }
        }
        __syncthreads();
#pragma unroll
        for (int rb = 0; rb < 3; ++rb)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = be[cb];
        mma_tile<8, NCB>(tA, we, acc, lane);                      // W_e (:1229)
#pragma unroll
        for (int rb = 0; rb < 3; ++rb)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) st4(tB + chunk_off(16 * rb + m, chunk0 + 4 * cb + q), acc[rb][cb]);
        __syncthreads();
#pragma unroll
        for (int it = 0; it < RPW / 2; ++it) {
            const int row = RPW * wv + 2 * it + (lane >> 5);
            const f4 y = s_idx[row] >= 0 ? ld4(tB + chunk_off(row, c32)) : f4{0.f, 0.f, 0.f, 0.f};
            st4(a.hE + ((size_t)i * TM_KS + row) * TM_H + 4 * c32, y);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Split-precision featurizer (default, f16x2): same pipeline, both GEMMs on the 16-bit matrix cores (tmpnn_split.h).
// The Gaussians are split into planes as they are generated (they never exist as an fp32 tile); the RBF planes have a
// 1024-byte row pitch (416 values used) with the usual 16-byte-chunk XOR swizzle, which is conflict-free for the
// ds_read_b128 lane groups of gfx950 (a plain 848-byte pitch measured 42 % conflict cycles). Columns 400..415 are zero K-padding. LayerNorm statistics are taken in the
// GEMM-1 epilogue (values stay in registers), its output goes straight into the GEMM-2 input planes, which — like the
// fp32 output tile — are aliased on the dead RBF planes.
// ------------------------------------------------------------------------------------------------
#define RBFP_ROWB 1024
#ifndef TM_FEAT_PF
#define TM_FEAT_PF 1      // B-fragment prefetch distance of GEMM 1 (mma_tile_split): 1 = 0.500 ms with 38 spilled VGPRs (reloaded around the GEMM, not in it) against 0.524 at 0 (13 spilled), 0.527 at 2
#endif
#ifndef TM_FEAT_DIST2
#define TM_FEAT_DIST2 1
#endif
#ifndef TM_FEAT_DMA
#define TM_FEAT_DMA 1     // 1: the next tile's rows go global -> LDS by LDS-DMA, issued before the Gaussians (see the kernel); 0: through 18
                          // VGPRs of every wavefront, issued after them (round 2)
#endif
// one LDS-DMA piece: lane l's 16 (4) bytes at `src` land at LDS byte address lds + 16 (4) * l (wave-uniform base in M0; retired
// through vmcnt). Inline asm ON PURPOSE: with the builtin, hipcc orders every later ds_read of the wavefront behind the piece
// (s_waitcnt vmcnt(0) in front of the first LDS read — here the Gaussians), which exposes exactly the latency the piece is
// issued early to hide. The compiler does not count these in its vmcnt bookkeeping, so its own waits can only over-wait;
// the consumer waits explicitly (`publish`).
__device__ __forceinline__ unsigned tm_lds_addr(const void *p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void *)p;
}
__device__ __forceinline__ void tm_glds16(const void *src, unsigned lds) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ void tm_glds4(const void *src, unsigned lds) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(src), "s"(lds) : "memory", "m0");
}
// PROF: phase timing (s_memtime deltas of thread 0 of workgroup 0, summed over its tiles) into prof[0..8] — TMPNN_FEAT_PROF=1
// KNN (small launches, one tile per workgroup): the neighbour row of the workgroup's residue is computed HERE by wavefront 0
// (knn_residue<4>: T <= #CUs <= 256, so every protein fits the register form) while the other seven load their weight fragments — one
// launch less in front of a single protein (2.5 us of dispatch + the start-up latency of a kernel, tools/gap_probe.py).
struct KnnFuseArgs { const float *mask; const int32_t *offsets; int N, max_len, K; int32_t *E_idx; float *D_nb; KnnInit init; int sel_rows; };

template <typename SP, bool PROF = false, bool IMG = false, bool KNN = false>
__global__ __launch_bounds__(512, 2) void featurize_split_kernel(FeatArgs a, unsigned long long *prof = nullptr, KnnFuseArgs kf = KnnFuseArgs{}) {
    unsigned long long t_last = 0;
    auto mark = [&](int k) {
        if (PROF && tm_bid() == 0 && tm_tid() == 0) {
            const unsigned long long t = __builtin_readcyclecounter();
            if (k >= 0) prof[k] += t - t_last;
            t_last = t;
        }
    };
    constexpr int PLB = TM_TILE * RBFP_ROWB;                 // bytes per RBF plane
    constexpr int TILEB = SP::NP * SPLIT_PLANE_BYTES;
    __shared__ __attribute__((aligned(16))) char rbf[SP::NP * PLB];
    // GEMM 2's operand planes: a tile of their own (until round 6 aliased on the RBF planes, with an fp32 output tile here — see the stores)
    __shared__ __attribute__((aligned(16))) char s_tA[TILEB];
    __shared__ __attribute__((aligned(16))) float s_stat[TM_TILE][TM_STAT8_LD];
    __shared__ float s_atoms[TM_TILE][16];
    __shared__ float s_self[16];
    __shared__ float s_dist[TM_TILE][28];
    __shared__ int s_ix[2][2][TM_TILE];                      // [buffer][neighbour index | positional index][neighbour]: ONE array, one lane base
    __shared__ __attribute__((aligned(16))) float s_const[3][TM_H];   // W_e bias, LayerNorm gain / bias: read where used, not held (12 VGPRs)
    char *tAp = s_tA;
    const int tid = tm_tid(), lane = tid & 63, wv = tid >> 6, m = lane & 15, q = lane >> 4;
    const int ncol = 16 * wv + 4 * q, c4 = 4 * wv + q;

    // Weight fragments, resident for the whole launch (136 VGPRs). With fragment images (f16x2 handles) they arrive as coalesced
    // 1 KB loads instead of 16-row fp32 gathers split on the fly (the prologue is what a single protein pays for). Re-reading
    // W_e per tile instead of keeping it (to free 32 VGPRs for the GEMM-1 pipeline) was measured twice: 64 KB per tile through
    // the CU's 64 B/clk return path costs more than the pipeline gains (0.546 vs 0.534 ms).
    WFragS<SP> wedge[1][13], we[1][4];
    if constexpr (IMG) {
#pragma unroll
        for (int st = 0; st < 13; ++st) {
            const char *p = a.img_e[st >> 2] + (size_t)wv * 8192 + (st & 3) * 2048 + lane * 16;
            wedge[0][st].p[0] = *reinterpret_cast<const u4 *>(p);
            wedge[0][st].p[1] = *reinterpret_cast<const u4 *>(p + 1024);
        }
        const char *pw = a.img_we + (size_t)wv * 8192 + lane * 16;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            we[0][c].p[0] = *reinterpret_cast<const u4 *>(pw + 2048 * c);
            we[0][c].p[1] = *reinterpret_cast<const u4 *>(pw + 2048 * c + 1024);
        }
    } else {
        load_wfrag_split<SP, 13>(a.edge_w, 416, 16 * wv, 16, 400, wedge[0], lane, 16);
        load_wfrag_split<SP, 4>(a.We_w, TM_H, 16 * wv, 0, TM_H, we[0], lane);
    }
    if (tid < 3 * TM_H / 4) {
        const int w = tid >> 5, c = 4 * (tid & 31);
        st4(&s_const[w][c], ld4((w == 0 ? a.We_b : w == 1 ? a.ln_w : a.ln_b) + c));   // visible after the prologue's barriers
    }
#if TM_FEAT_DMA
    // PositionalEncodings (:896-908) = one_hot(d) . W_pos^T + b_pos: 16 values per edge, a 66-row table in LDS. They ride GEMM 1
    // in its K padding (columns 400..415 against edge_embedding.weight[:, 0:16]) instead of arriving as accumulator rows from
    // a [66,128] global table: no global load between the tile loop's barriers at all
    __shared__ __attribute__((aligned(16))) float s_pos[66][16];
    for (int e = tid; e < 66 * 16; e += 512) {
        const int d = e >> 4, pp = e & 15;
        s_pos[d][pp] = a.pos_w[pp * 66 + d] + a.pos_b[pp];
    }
#endif
    f4 mu4;                                                   // this thread's 4 Gaussian centres: (tid & 3) is fixed
#pragma unroll
    for (int r = 0; r < 4; ++r) mu4[r] = a.mu[(tid & 3) * 4 + r];
    // consumed HERE so that the wait for this load sits in the prologue: left pending, the compiler's scoreboard puts a
    // vmcnt(N) into the Gaussian loop of every tile, which (vmcnt retires in order) waits for the DMA pieces issued in front of it
    asm volatile("" ::"v"(mu4.x), "v"(mu4.y), "v"(mu4.z), "v"(mu4.w));

    // Per-tile inputs (neighbour list, 5 atoms of every neighbour, positional index, Ca-Ca distance), one tile ahead.
    // TM_FEAT_DMA = 1 (round 3): the phase profile showed GEMM 1 at 1.85 x its matrix time only because wavefront 0 met the
    // dependent global loads (E_idx -> X[j]) in front of it — vmcnt retires in order, so the accumulator rows issued behind
    // them waited for both latencies (2 600 of the tile's 17 300 cycles), and the 18 registers holding the rows across the
    // GEMM were spilled and reloaded in `publish` (900 cycles, serial). Now: the list entry of tile i+2 is an ordinary load
    // (1 VGPR, consumed one tile later); the rows of tile i+1 go global -> LDS by LDS-DMA, issued by wavefront 0 (neighbours)
    // and 1 (the residue itself) BEFORE the Gaussians, which outlast their latency; this tile's positional rows are loaded
    // into the accumulators at the same point. `publish` reads the raw rows back, adds the virtual Cb and writes the tables
    // `distances` uses. (All 64 lanes of both wavefronts issue: lanes >= 48 duplicate neighbour 47, so no EXEC-masked DMA.)
    // TM_FEAT_DMA = 0: registers, issued after the Gaussians; a two-deep variant of that form was measured in round 2: no gain.
#if TM_FEAT_DMA
    __shared__ __attribute__((aligned(16))) float s_raw[3][64][4];   // neighbours' rows: [16-byte piece of the 48-byte row][lane]
    __shared__ __attribute__((aligned(16))) float s_sraw[64];        // the residue's own row, one word per lane (12 used)
    __shared__ int s_misc[3][64];                            // per neighbour: masked Ca-Ca distance, residue_idx[j], chain[j]; lane 48 of the last two: [i]
    __shared__ int s_list[2][64];                            // neighbour lists, two tiles ahead: [tile parity][neighbour]
    const int wu = __builtin_amdgcn_readfirstlane(wv);       // scalar branches around the pieces
#else
    float g_at[15];
    float g_d0 = 0.f;
    int g_idx = -1, g_dpos = 0;
#endif
    // (lane offsets laundered through an empty asm: otherwise the compiler hoists base + lane as 64-bit pairs out of the tile
    //  loop, spills them, and the reload's vmcnt(0) — in order behind the DMA pieces — waits for the pieces)
#if TM_FEAT_DMA
    // the 8 pieces of one tile, ONE per wavefront (a piece costs its wavefront several hundred cycles of issue): rows of tile ii
    // through its list in s_list[lb] (landed a tile ago), and the list of tile i2 (two ahead; < 0: none) into s_list[lb ^ 1].
    // Lanes 48..63 of the per-neighbour pieces address the residue itself: lane 48 of residue_idx / chain is what `publish`
    // compares against, no separate piece and no scalar load for them.
    auto fetch = [&](int ii, int lb, int i2) {
        int nb = lane < TM_TILE ? lane : TM_TILE - 1;
        asm volatile("" : "+v"(nb));
        const int jn = lane < TM_TILE ? s_list[lb][nb] : -1;
        const int jj = jn < 0 ? ii : jn;
        const float *x = a.X + (size_t)jj * 12;
        if (wu == 0) {
            tm_glds16(x, tm_lds_addr(&s_raw[0][0][0]));
        } else if (wu == 1) {
            tm_glds16(x + 4, tm_lds_addr(&s_raw[1][0][0]));
        } else if (wu == 2) {
            tm_glds16(x + 8, tm_lds_addr(&s_raw[2][0][0]));
        } else if (wu == 3) {
            tm_glds4(a.X + (size_t)ii * 12 + (lane < 12 ? lane : 11), tm_lds_addr(s_sraw));
        } else if (wu == 4) {
            tm_glds4(a.ridx + jj, tm_lds_addr(s_misc[1]));
        } else if (wu == 5) {
            tm_glds4(a.cenc + jj, tm_lds_addr(s_misc[2]));
        } else if (wu == 6) {
            tm_glds4((a.D_nb + (size_t)ii * TM_KS) + nb, tm_lds_addr(s_misc[0]));   // masked Ca-Ca distance from _dist (:1142)
        } else if (i2 >= 0) {
            tm_glds4((a.E_idx + (size_t)i2 * TM_KS) + nb, tm_lds_addr(s_list[lb ^ 1]));
        }
    };
    auto landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };   // in front of the barrier that publishes the pieces
#else
    auto fetch = [&](int ii) {
        if (tid < TM_TILE) {
            const int j = a.E_idx[(size_t)ii * TM_KS + tid];
            g_idx = j;
            const int jj = j < 0 ? ii : j;
            atoms5(a.X + (size_t)jj * 12, g_at);
            const int off = a.ridx[ii] - a.ridx[jj];              // PositionalEncodings index (:903-905, :1170-1175)
            const int same = a.cenc[ii] == a.cenc[jj];
            g_dpos = same ? min(max(off + 32, 0), 64) : 65;
            g_d0 = a.D_nb[(size_t)ii * TM_KS + tid];              // masked Ca-Ca distance from _dist (:1142)
        } else if (tid == 64) {
            atoms5(a.X + (size_t)ii * 12, g_at);
        }
    };
#endif
    auto publish = [&](int buf, int lb) {
#if TM_FEAT_DMA
        // (everything local: a value assigned under a wavefront test and declared outside the tile loop is carried through
        //  it as a phi in every wavefront — 17 VGPRs of pressure in the round-2 form)
        if (wu < 2) {                                           // (the pieces landed before the last barrier)
            float x[12], at[15];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const f4 v = wu == 0 ? ld4(&s_raw[c][lane][0]) : ld4(&s_sraw[4 * c]);
                x[4 * c] = v.x; x[4 * c + 1] = v.y; x[4 * c + 2] = v.z; x[4 * c + 3] = v.w;
            }
            atoms5(x, at);
            if (wu == 1) {
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < 15; ++k) s_self[k] = at[k];
                }
            } else if (lane < TM_TILE) {
                const int off = s_misc[1][TM_TILE] - s_misc[1][lane];   // PositionalEncodings index (:903-905, :1170-1175)
                const int same = s_misc[2][TM_TILE] == s_misc[2][lane];
                s_ix[buf][0][lane] = s_list[lb][lane];
                s_ix[buf][1][lane] = same ? min(max(off + 32, 0), 64) : 65;
#pragma unroll
                for (int k = 0; k < 15; ++k) s_atoms[lane][k] = at[k];
                s_atoms[lane][15] = __builtin_bit_cast(float, s_misc[0][lane]);
            }
        }
#else
        if (tid < TM_TILE) {
            s_ix[buf][0][tid] = g_idx;
            s_ix[buf][1][tid] = g_dpos;
#pragma unroll
            for (int k = 0; k < 15; ++k) s_atoms[tid][k] = g_at[k];
            s_atoms[tid][15] = g_d0;
        } else if (tid == 64) {
#pragma unroll
            for (int k = 0; k < 15; ++k) s_self[k] = g_at[k];
        }
#endif
    };
    // 25 atom-pair distances of the 48 neighbours: 1 200 values over 512 threads
    auto dist_one = [&](int e) {
        const int mm = e / 25, p = e - mm * 25;
        // atom indices of pair p (c_pair_a / c_pair_b, 3 bits each) from immediates: no per-lane constant-memory load
        const int sh = 3 * (p & 15);
        const int ia = (int)(((p < 16 ? 0xe400124c681ull : 0x26a351aull) >> sh) & 7ull);
        const int ib = (int)(((p < 16 ? 0x29a8d4684681ull : 0x3900049ull) >> sh) & 7ull);
        const float *A = s_self + 3 * ia;
        const float *B = s_atoms[mm] + 3 * ib;
        const float d0 = s_atoms[mm][15];                         // pair 0: the masked Ca-Ca distance of _dist
        const float dx = A[0] - B[0], dy = A[1] - B[1], dz = A[2] - B[2];
        const float D = __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz + 1e-6f);  // _get_rbf (:1122); v_sqrt_f32, 1 ulp
        return p == 0 ? d0 : D;
    };
    auto distances = [&]() {
#if TM_FEAT_DIST2
        // the two full rounds as one straight-line block (their LDS reads and square roots overlap), then the round of 176
        const float D0 = dist_one(tid), D1 = dist_one(tid + 512);
        const int m0 = tid / 25, m1 = (tid + 512) / 25;
        s_dist[m0][tid - 25 * m0] = D0;
        s_dist[m1][tid + 512 - 25 * m1] = D1;
        if (tid < TM_TILE * 25 - 1024) {
            const int e = tid + 1024, mm = e / 25;
            s_dist[mm][e - 25 * mm] = dist_one(e);
        }
#else
        for (int e = tid; e < TM_TILE * 25; e += 512) {
            const int mm = e / 25;
            s_dist[mm][e - 25 * mm] = dist_one(e);
        }
#endif
    };

    const TileRange tr = xcd_tile_range(a.T);
    int i = tr.begin, cur = 0;
    if constexpr (KNN) {
#if TM_FEAT_DMA
        __shared__ unsigned s_selk[64][2];
        if (tm_bid() == 0 && tid == 0 && kf.init.status_zero) *kf.init.status_zero = 0;     // (as knn_kernel: nothing in this launch ORs into it)
        if (wu == 0)
            for (int r = tr.begin; r < tr.end; r += tr.step)
                knn_residue<4>(nullptr, s_selk, a.X, kf.mask, kf.offsets, kf.N, kf.max_len, kf.K, kf.E_idx, kf.D_nb, nullptr, kf.init, kf.sel_rows, r, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the row is in L2 before any wavefront of this workgroup asks for it
        __syncthreads();
#endif
    }
    if (i < tr.end) {
#if TM_FEAT_DMA
        if (wu == 7) {
            const int nb = lane < TM_TILE ? lane : TM_TILE - 1;
            tm_glds4((a.E_idx + (size_t)i * TM_KS) + nb, tm_lds_addr(s_list[0]));
            landed();
        }
        __syncthreads();
        fetch(i, 0, i + tr.step < tr.end ? i + tr.step : -1);
        landed();
        __syncthreads();
#else
        fetch(i);
#endif
        publish(0, 0);
        __syncthreads();
        distances();
        __syncthreads();
    }
    mark(-1);
    for (; i < tr.end; i += tr.step) {
        const int inext = i + tr.step;
        const bool has_next = inext < tr.end;
#if TM_FEAT_DMA
        if (tid < TM_TILE * 4) {                                // K columns 400..415: this tile's positional features
            const int row = tid >> 2, c = tid & 3;
            store_split<SP, TM_TILE, RBFP_ROWB>(rbf, row, 100 + c, ld4(&s_pos[s_ix[cur][1][row]][4 * c]));
        }
        f4 acc[3][1];
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) acc[rb][0] = f4{0.f, 0.f, 0.f, 0.f};
        if (has_next) fetch(inext, cur ^ 1, inext + tr.step < tr.end ? inext + tr.step : -1);   // tile i's list is in s_list[cur]
#else
        if (tid < TM_TILE * 2 * SP::NP) {                       // zero the K padding (columns 400..415) of every plane row
            const int p = tid / (TM_TILE * 2), rem = tid - p * (TM_TILE * 2);
            *reinterpret_cast<u4 *>(rbf + plane_off8<TM_TILE, RBFP_ROWB>(p, rem >> 1, 50 + (rem & 1))) = u4{0u, 0u, 0u, 0u};
        }
        f4 acc[3][1];
#endif
        // 16 Gaussians per pair, 4 per thread-iteration (:1111-1119). (Measured and dropped in the spill-free kernel: stepping
        // (row, quad) instead of dividing by 100 — 8 simple ops for 3 integer multiplies, +1.6 %; the v_fma_mix split, nil.)
#if TM_ABL_NOGAUSS
        if (false)
#endif
        // (round 6: a thread keeps ONE column group c = tid % 100 — c & 3 == tid & 3: its four centres — and walks the rows tid / 100, + 5, ...:
        //  the plane address and the distance address advance by constants, only the row's swizzle term is recomputed; with e = tid + 512 k
        //  every iteration divided by 100 and rebuilt both addresses: 15 of its 39 vector instructions. Threads 500..511 idle: 2.3 %.)
        int tl = tid;                                          // (laundered: hoisted out of the tile loop, c and its addresses are three VGPRs
        asm volatile("" : "+v"(tl));                           //  the kernel, at 256, does not have — 12 bytes of scratch)
        // (the distance of the NEXT iteration requested one iteration ahead — the read is waited for right where it is issued — measured:
        //  nil, 2.385 against 2.374 of the message kernel's time; the other wavefront of the SIMD covers the LDS latency)
        for (int mm = tl / 100, c = tl - 100 * (tl / 100); mm < TM_TILE && tl < 500; mm += 5) {
            const float D = s_dist[mm][c >> 2];
            // exp(-((D - mu) / 1.25)^2) = exp2(-(t t)), t = (D - mu) * 0.8 sqrt(log2 e): packed fp32, two centres per op
            const f2 t01 = (f2{D, D} - f2{mu4.x, mu4.y}) * 0.96089795f, t23 = (f2{D, D} - f2{mu4.z, mu4.w}) * 0.96089795f;
            const f2 e01 = -(t01 * t01), e23 = -(t23 * t23);
            unsigned lo2[SP::NP], hi2[SP::NP];
            SP::split2(f2{__builtin_amdgcn_exp2f(e01.x), __builtin_amdgcn_exp2f(e01.y)}, lo2);
            SP::split2(f2{__builtin_amdgcn_exp2f(e23.x), __builtin_amdgcn_exp2f(e23.y)}, hi2);
#pragma unroll
            for (int p = 0; p < SP::NP; ++p)
                *reinterpret_cast<u2 *>(rbf + plane_off4<TM_TILE, RBFP_ROWB>(p, mm, c)) = u2{lo2[p], hi2[p]};
        }
        mark(0);
#if TM_FEAT_DMA
        landed();
#endif
        __syncthreads();                                       // RBF planes complete; s_dist / s_atoms consumed; DMA pieces in LDS
        mark(1);

#if !TM_FEAT_DMA
        if (has_next) fetch(inext);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) acc[rb][0] = ld4(a.pos_table + s_ix[cur][1][16 * rb + m] * TM_H + ncol);
#endif
        mma_tile_split<SP, 13, 1, 3, TM_TILE, RBFP_ROWB, 13, 0, true, TM_FEAT_PF>(rbf, wedge, acc, lane);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) row_stats_partial1b(acc[rb][0], &s_stat[16 * rb + m][2 * wv], q);
        mark(2);
        if (has_next) publish(cur ^ 1, cur ^ 1);
        mark(3);
        __syncthreads();                                       // RBF planes dead, statistics + next tile's atoms complete
        mark(4);
        const f4 g4 = ld4(&s_const[1][ncol]), b4 = ld4(&s_const[2][ncol]);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {                        // norm_edges (:1179)
            const int row = 16 * rb + m;
            float mean, rstd;
            row_stats_finish8b(&s_stat[row][0], mean, rstd);
            const f4 y = (acc[rb][0] - mean) * rstd * g4 + b4;
            store_split<SP>(tAp, row, c4, y);
            if (a.E_opt) st4(a.E_opt + ((size_t)i * TM_KS + row) * TM_H + ncol, s_ix[cur][0][row] >= 0 ? y : f4{0.f, 0.f, 0.f, 0.f});
        }
        mark(5);
        if (has_next) distances();
        mark(6);
        __syncthreads();
        mark(7);
        const f4 be = ld4(&s_const[0][ncol]);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) acc[rb][0] = be;
        mma_tile_split<SP, 4, 1>(tAp, we, acc, lane);           // W_e (:1229)
        // round 6: h_E rows leave from the accumulators — a wavefront's store covers 16 rows x 64 B (its 16 columns) — instead of crossing
        // LDS into full 512-byte rows (an fp32 tile, one more barrier, 3 ds_read_b128 + 3 stores per lane): -2.3 %, profiles/r06_ab_feat_store.txt
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) {
            const int row = 16 * rb + m;
            st4(a.hE + ((size_t)i * TM_KS + row) * TM_H + ncol, s_ix[cur][0][row] >= 0 ? acc[rb][0] : f4{0.f, 0.f, 0.f, 0.f});
        }
        mark(8);
        cur ^= 1;
        // no barrier behind GEMM 2: the next tile's Gaussians and positional columns go into the RBF planes (last read by GEMM 1, two barriers
        // ago), its DMA pieces into buffers publish() consumed before the last-but-one barrier; s_tA is rewritten after two more barriers,
        // s_ix[cur ^ 1] (this tile's, read by the stores above) by publish() after one
    }
}

// ------------------------------------------------------------------------------------------------
// gathers
// ------------------------------------------------------------------------------------------------
// out[r, :] = nodes[base(r) + idx[r], :] with C % 4 == 0; one 16-byte chunk per thread-iteration.
template <typename IdxT>
__global__ __launch_bounds__(TM_THREADS) void gather_rows_kernel(const float *__restrict__ nodes,
                                                                 const IdxT *__restrict__ idx, int64_t n_rows,
                                                                 int64_t rows_per_batch, int64_t nodes_per_batch,
                                                                 int C4, float *__restrict__ out) {
    const int64_t total = n_rows * C4;
    // XCD-aware split (see xcd_tile_range): each XCD streams one contiguous eighth of the output, so the table
    // rows it gathers (neighbours are local) stay in its own L2
    int64_t g0 = (int64_t)tm_bid() * TM_THREADS + tm_tid(), g1 = total, stride = (int64_t)tm_nblk() * TM_THREADS;
    if ((tm_nblk() & 7) == 0 && total >= 8 * stride) {
        const int x = tm_bid() & 7;
        const int64_t s = total / 8 * x;
        g1 = x == 7 ? total : total / 8 * (x + 1);
        g0 = s + (int64_t)(tm_bid() >> 3) * TM_THREADS + tm_tid();
        stride >>= 3;
    }
    for (int64_t g = g0; g < g1; g += stride) {
        const int64_t r = g / C4;
        const int c = (int)(g - r * C4);
        const int64_t j = (int64_t)idx[r];
        f4 v = f4{0.f, 0.f, 0.f, 0.f};
        if (j >= 0) {
            const int64_t base = rows_per_batch > 0 ? (r / rows_per_batch) * nodes_per_batch : 0;
            v = ld4(nodes + ((base + j) * C4 + c) * 4);
        }
        __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(out) + g);
    }
}

// scalar fallback for C % 4 != 0 (e.g. gather_nodes(mask.unsqueeze(-1), E_idx), :1232)
template <typename IdxT>
__global__ __launch_bounds__(TM_THREADS) void gather_rows_scalar_kernel(const float *__restrict__ nodes,
                                                                        const IdxT *__restrict__ idx, int64_t n_rows,
                                                                        int64_t rows_per_batch, int64_t nodes_per_batch,
                                                                        int C, float *__restrict__ out) {
    const int64_t total = n_rows * C;
    const int64_t stride = (int64_t)tm_nblk() * TM_THREADS;
    for (int64_t g = (int64_t)tm_bid() * TM_THREADS + tm_tid(); g < total; g += stride) {
        const int64_t r = g / C;
        const int c = (int)(g - r * C);
        const int64_t j = (int64_t)idx[r];
        const int64_t base = rows_per_batch > 0 ? (r / rows_per_batch) * nodes_per_batch : 0;
        out[g] = j >= 0 ? nodes[(base + j) * C + c] : 0.f;
    }
}

// out[b,i,k,:] = edges[b,i,idx[b,i,k],:]
__global__ __launch_bounds__(TM_THREADS) void gather_edges_kernel(const float *__restrict__ edges,
                                                                  const int64_t *__restrict__ idx, int64_t n_rows,
                                                                  int N, int K, int C, float *__restrict__ out) {
    const int64_t total = n_rows * C;
    const int64_t stride = (int64_t)tm_nblk() * TM_THREADS;
    for (int64_t g = (int64_t)tm_bid() * TM_THREADS + tm_tid(); g < total; g += stride) {
        const int64_t r = g / C;            // r = (b*N + i)*K + k
        const int c = (int)(g - r * C);
        const int64_t bi = r / K;
        out[g] = edges[(bi * N + idx[r]) * C + c];
    }
}

// centrality: number of OTHER residues of the same protein whose CA lies within `radius` of this residue's CA
// (compute_centrality, analysis/thermompnn_benchmarking.py:20-35: cdist, NaN -> 2*radius, count < radius, minus 1).
// A residue without coordinates (mask 0) is at distance 2*radius from everything, itself included -> -1.
__global__ __launch_bounds__(TM_THREADS) void centrality_kernel(const float *__restrict__ X, const float *__restrict__ mask,
                                                                const int32_t *__restrict__ offsets, int N, int T,
                                                                float radius, int32_t *__restrict__ out) {
    const int lane = tm_tid() & 63, wv = tm_wave(tm_tid());
    for (int i = tm_bid() * 4 + wv; i < T; i += tm_nblk() * 4) {
        int lo = 0, hi = N;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (offsets[mid] <= i) lo = mid; else hi = mid;
        }
        const int s = offsets[lo], L = offsets[lo + 1] - s;
        const float xi = X[(size_t)i * 12 + 3], yi = X[(size_t)i * 12 + 4], zi = X[(size_t)i * 12 + 5];
        const bool vi = mask[i] > 0.f;
        int cnt = 0;
        for (int j = lane; j < L; j += 64) {
            const float *c = X + (size_t)(s + j) * 12 + 3;
            const float dx = c[0] - xi, dy = c[1] - yi, dz = c[2] - zi;
            const float d = sqrtf(dx * dx + dy * dy + dz * dz);
            cnt += (vi && mask[s + j] > 0.f && d < radius) ? 1 : 0;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
        if (lane == 0) out[i] = cnt - 1;
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
int launch_centrality(const float *X, const float *mask, const int32_t *offsets, int N, int64_t T, float radius,
                      int32_t *out, hipStream_t st) {
    const int64_t blocks = (T + 3) / 4, cap = (int64_t)tm_num_cus() * 8;
    { tm_prof_begin("centrality", st); centrality_kernel<<<(int)(blocks < cap ? blocks : cap), TM_THREADS, 0, st>>>(X, mask, offsets, N, (int)T, radius, out); tm_prof_end(st); }
    return tm_check_launch("centrality");
}

int launch_knn(const float *X, const float *mask, const int32_t *offsets, int N, int64_t T, int max_len, int K,
               int32_t *E_idx, float *D_nb, int32_t *status, hipStream_t st, KnnInit init) {
    const size_t lds = (size_t)4 * (max_len + (max_len >> 6) + 1) * sizeof(float);   // knn_slot padding
    if (lds > 160 * 1024) return tm_set_error(TMPNN_E_UNSUPPORTED, "knn_topk: max_len %d needs %zu B of LDS", max_len, lds);
    const int64_t blocks = (T + 3) / 4;
    const int64_t cap = (int64_t)tm_num_cus() * 8;
    const int grid = (int)(blocks < cap ? blocks : cap);
    static const bool reg_rows = TM_DBG_FLAG("TMPNN_KNN_REG", true);
    static const int sel_rows = TM_DBG_FLAG("TMPNN_KNN_SEL", true) ? 1 : 0;   // 0: extract-min rounds only (A/B, debug build)
    tm_prof_begin("knn", st);
    if (reg_rows && max_len <= 256) knn_kernel<4><<<grid, TM_THREADS, 0, st>>>(X, mask, offsets, N, (int)T, max_len, K, E_idx, D_nb, status, init, sel_rows);
    else if (reg_rows && max_len <= 512) knn_kernel<8><<<grid, TM_THREADS, 0, st>>>(X, mask, offsets, N, (int)T, max_len, K, E_idx, D_nb, status, init, sel_rows);
    else {
        // per call, not cached: the attribute is per device and one process may drive several GPUs
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(knn_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        knn_kernel<0><<<grid, TM_THREADS, lds, st>>>(X, mask, offsets, N, (int)T, max_len, K, E_idx, D_nb, status, init, 0);
    }
    tm_prof_end(st);
    return tm_check_launch("knn_topk");
}

// kf != nullptr: the k-NN rows are computed inside the launch (featurize_fusable says when that form exists)
bool featurize_fusable(const tmpnn_weights *w, int64_t T) {
    if (tm_matmul_mode() != TM_MM_F16X2 || T <= 0 || T > (int64_t)tm_num_cus() || !TM_FEAT_DMA) return false;
    for (int b = 0; b < 4; ++b) if (!tm_find_wimg(w->edge_w + 16 + 128 * b)) return false;
    return tm_find_wimg(w->We_w) != nullptr;
}

int launch_featurize(const tmpnn_weights *w, const float *X, const int32_t *ridx, const int32_t *cenc,
                     const int32_t *E_idx, const float *D_nb, int64_t T, float *h_E, float *E_opt, hipStream_t st,
                     const KnnFuse *knn) {
    FeatArgs a;
    a.edge_w = w->edge_w; a.pos_table = w->pos_table; a.pos_w = w->pos_w; a.pos_b = w->pos_b; a.ln_w = w->norm_edges_w; a.ln_b = w->norm_edges_b;
    a.We_w = w->We_w; a.We_b = w->We_b; a.X = X; a.ridx = ridx; a.cenc = cenc; a.E_idx = E_idx; a.D_nb = D_nb;
    a.hE = h_E; a.E_opt = E_opt; a.T = (int)T;
    for (int i = 0; i < 16; ++i)   // torch.linspace(2, 22, 16): double arithmetic, symmetric halves, cast to fp32
        a.mu[i] = i < 8 ? (float)(2.0 + (20.0 / 15.0) * i) : (float)(22.0 - (20.0 / 15.0) * (15 - i));
    bool img = tm_matmul_mode() == TM_MM_F16X2;
    for (int b = 0; b < 4; ++b) { a.img_e[b] = img ? tm_find_wimg(w->edge_w + 16 + 128 * b) : nullptr; img = img && a.img_e[b]; }
    a.img_we = img ? tm_find_wimg(w->We_w) : nullptr;
    img = img && a.img_we;
    static const bool img_on = TM_DBG_FLAG("TMPNN_FEAT_IMG", true);
    if (!img_on) img = false;
    const int64_t cap = tm_num_cus();
    static const int nw = TM_DBG_INT("TMPNN_FEAT_WAVES", 8);
    // (two split-precision bf16x3 forms of this kernel — half-width tiles, and one 126 KB single-pass plane tile — were
    //  measured and dropped: generating + splitting the 19 200 Gaussians into three planes and the extra LDS traffic cost as
    //  much as the shorter 400->128 GEMM saved: 1.07-1.08 ms vs 1.08 ms)
    tm_prof_begin("featurize", st);
    static const bool split_ok = TM_DBG_FLAG("TMPNN_FEAT_SPLIT", true);
#ifdef TMPNN_DEBUG_BUILD
    static const bool feat_prof = TM_DBG_FLAG("TMPNN_FEAT_PROF", false);
    if (tm_matmul_mode() == TM_MM_F16X2 && split_ok && feat_prof) {       // debug build: phase timing of workgroup 0 (synchronises!)
        static unsigned long long *d_prof = nullptr;
        if (!d_prof) (void)hipMalloc(&d_prof, 16 * sizeof(unsigned long long));
        (void)hipMemsetAsync(d_prof, 0, 16 * sizeof(unsigned long long), st);
        if (img) featurize_split_kernel<SplitH2, true, true><<<(int)(T < cap ? T : cap), 512, 0, st>>>(a, d_prof);
        else featurize_split_kernel<SplitH2, true><<<(int)(T < cap ? T : cap), 512, 0, st>>>(a, d_prof);
        unsigned long long h[16];
        (void)hipMemcpy(h, d_prof, sizeof(h), hipMemcpyDeviceToHost);
        fprintf(stderr, "featurize phases (cycles, wg 0): gauss %llu bar %llu gemm1+stats %llu publish %llu bar %llu ln+split %llu dist %llu bar %llu gemm2+store %llu\n",
                h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8]);
        tm_prof_end(st);
        return tm_check_launch("edge_featurize");
    }
#endif
    if (knn && !(tm_matmul_mode() == TM_MM_F16X2 && split_ok && img && T <= cap)) {
        // (only reachable in the debug library, whose switches can take the image / split form away after featurize_fusable said yes)
        tm_prof_end(st);
        const int rc = launch_knn(X, knn->mask, knn->offsets, knn->N, T, knn->max_len, knn->K, knn->E_idx, knn->D_nb, nullptr, st, knn->init);
        if (rc != TMPNN_OK) return rc;
        return launch_featurize(w, X, ridx, cenc, E_idx, D_nb, T, h_E, E_opt, st, nullptr);
    }
    if (knn) {                                                   // small launch: k-NN + featurizer in one (the caller asked featurize_fusable)
        static const bool sel_rows = TM_DBG_FLAG("TMPNN_KNN_SEL", true);
        KnnFuseArgs kf{knn->mask, knn->offsets, knn->N, knn->max_len, knn->K, knn->E_idx, knn->D_nb, knn->init, sel_rows ? 1 : 0};
        featurize_split_kernel<SplitH2, false, true, true><<<(int)T, 512, 0, st>>>(a, nullptr, kf);
        tm_prof_end(st);
        return tm_check_launch("knn_featurize_fused");
    }
    if (tm_matmul_mode() == TM_MM_F16X2 && split_ok) {
        if (img) featurize_split_kernel<SplitH2, false, true><<<(int)(T < cap ? T : cap), 512, 0, st>>>(a);
        else featurize_split_kernel<SplitH2><<<(int)(T < cap ? T : cap), 512, 0, st>>>(a);
        tm_prof_end(st);
        return tm_check_launch("edge_featurize");
    }
    if (nw == 4) featurize_kernel<4><<<(int)(T < cap ? T : cap), 256, 0, st>>>(a);
    else featurize_kernel<8><<<(int)(T < cap ? T : cap), 512, 0, st>>>(a);
    tm_prof_end(st);
    return tm_check_launch("edge_featurize");
}

int launch_gather_rows(const float *nodes, const void *idx, int idx64, int64_t n_rows, int64_t rows_per_batch,
                       int64_t nodes_per_batch, int C, float *out, hipStream_t st) {
    if (n_rows == 0) return TMPNN_OK;
    const int64_t cap = (int64_t)tm_num_cus() * 16;
    if (C % 4 == 0) {
        const int64_t blocks = (n_rows * (C / 4) + TM_THREADS - 1) / TM_THREADS;
        const int grid = (int)(blocks < cap ? blocks : cap);
        if (idx64) { tm_prof_begin("gather_rows", st); gather_rows_kernel<int64_t><<<grid, TM_THREADS, 0, st>>>(nodes, (const int64_t *)idx, n_rows, rows_per_batch, nodes_per_batch, C / 4, out); tm_prof_end(st); }
        else { tm_prof_begin("gather_rows", st); gather_rows_kernel<int32_t><<<grid, TM_THREADS, 0, st>>>(nodes, (const int32_t *)idx, n_rows, rows_per_batch, nodes_per_batch, C / 4, out); tm_prof_end(st); }
    } else {
        const int64_t blocks = (n_rows * C + TM_THREADS - 1) / TM_THREADS;
        const int grid = (int)(blocks < cap ? blocks : cap);
        if (idx64) { tm_prof_begin("gather_rows_scalar", st); gather_rows_scalar_kernel<int64_t><<<grid, TM_THREADS, 0, st>>>(nodes, (const int64_t *)idx, n_rows, rows_per_batch, nodes_per_batch, C, out); tm_prof_end(st); }
        else { tm_prof_begin("gather_rows_scalar", st); gather_rows_scalar_kernel<int32_t><<<grid, TM_THREADS, 0, st>>>(nodes, (const int32_t *)idx, n_rows, rows_per_batch, nodes_per_batch, C, out); tm_prof_end(st); }
    }
    return tm_check_launch("gather_rows");
}

int launch_gather_edges(const float *edges, const int64_t *idx, int B, int N, int K, int C, float *out, hipStream_t st) {
    const int64_t n_rows = (int64_t)B * N * K;
    if (n_rows == 0) return TMPNN_OK;
    const int64_t blocks = (n_rows * C + TM_THREADS - 1) / TM_THREADS;
    const int64_t cap = (int64_t)tm_num_cus() * 16;
    { tm_prof_begin("gather_edges", st); gather_edges_kernel<<<(int)(blocks < cap ? blocks : cap), TM_THREADS, 0, st>>>(edges, idx, n_rows, N, K, C, out); tm_prof_end(st); }
    return tm_check_launch("gather_edges");
}
