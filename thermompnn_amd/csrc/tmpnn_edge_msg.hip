// Small launches: the edge update of layer l and the message pass of layer l + 1 as one launch (f16x2).
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "tmpnn_split.h"
#include "tmpnn_internal.h"

// ------------------------------------------------------------------------------------------------
// Small launches (T <= #CUs: one tile per workgroup — a single protein, a handful of short ones): the edge update of
// encoder layer l and the message pass of the NEXT layer (encoder l+1, or decoder 0 after the last encoder layer) as ONE
// launch. Both need only this residue's edge tile plus node projections that node_update(l) has already written, so there is
// no grid-wide dependency between them; a launch costs 2.5 us of dispatch + a prologue even when it does nothing
// (tools/gap_probe.py), and the fresh LayerNorm'd tile is in registers in exactly the row layout the message pass splits from.
// The five weight fragments do not have to be resident together here (nothing persists across tiles): the message weights are
// loaded into the registers the edge weights leave. Arithmetic = enc_edge8_rp_kernel followed by msg8_rp_kernel, operation for
// operation (the same GEMM step order, the same epilogue expressions): results are BIT-IDENTICAL to the two-launch path, so a
// protein's numbers do not depend on the batch it is in (tests: test_small_launch_fused_forms_are_bit_identical).
// ------------------------------------------------------------------------------------------------
template <typename SP, bool DEC>
__global__ __launch_bounds__(512, 2) void edge_msg_fused_kernel(EdgeArgsB a, MsgArgsB b) {
    constexpr int TILEB = SP::NP * SPLIT_PLANE_BYTES;
    static_assert(TILEB >= TM_TILE * TM_H * 4, "the fp32 LayerNorm tile is aliased on the x planes");
    __shared__ __attribute__((aligned(16))) char tE[TILEB];
    __shared__ __attribute__((aligned(16))) char tX[TILEB];              // x planes; the fp32 LayerNorm input; the message pass's tA
    // GEMM 2's output planes live where the e planes were: GEMM 1 was their last reader (every wavefront is past the barrier behind
    // it), the next tile's e planes are written only behind the barrier that follows GEMM 3. Two plane tiles instead of three:
    // 53 KB of LDS, every LDS offset below 64 KB (an offset above costs an address VGPR + a v_or each: 10 VALU per tile).
    char *const tY = tE;
    __shared__ __attribute__((aligned(16))) float s_stat[TM_TILE][TM_STAT_LD];
    __shared__ int s_idx[TM_TILE];
    __shared__ float s_ma[TM_TILE];
    float *tO = reinterpret_cast<float *>(tX);
    const int tid = tm_tid(), lane = tid & 63, wv = tid >> 6, m = lane & 15, q = lane >> 4;
    const int ncol = 16 * wv + 4 * q, c4 = 4 * wv + q;
    const int c32 = lane & 31;
    const unsigned ucol = (unsigned)ncol;
    const unsigned eoff = (unsigned)(m * TM_H + ncol);
    const unsigned soff = (unsigned)((6 * wv + (lane >> 5)) * TM_H + 4 * c32);
    const unsigned foff = (unsigned)(wv * 8192 + lane * 16);              // this lane's 16 bytes inside a fragment image (+ 2048 step + 1024 plane)

    for (int i = tm_bid(); i < a.T; i += tm_nblk()) {
        // ---- edge update of this tile (enc_edge8_rp_kernel) ------------------------------------------
        f4 e_cur[3], yrow[3];
        f4 g0, gj[3];                                    // the message pass's node terms: requested with the edge update's (same list)
        float mi, nma = 0.f;
        // The five weight fragments are a software pipeline through TWO register sets (64 VGPRs), each requested one GEMM phase
        // ahead of its use, into the set the previous GEMM has just finished with: fa = W11 -> W13 -> W2, fb = W12 -> W1.
        // (All five resident, or the message pair requested early, spills — and a scratch reload's vmcnt wait drains every
        //  prefetch in flight: 18.4 us per launch against 15.8.)
        WFragS<SP> fa[1][4], fb[1][4];
        f4 bias2;
        // round 6: the fragment a GEMM phase later needs is requested piece by piece behind the MFMA steps of the GEMM that runs on the OTHER
        // register set (eight global_loads in a row cost the wavefront ~85 cycles of issue each); the images exist for every f16x2 handle
        auto ride_into = [&](const char *img, WFragS<SP> (&dst)[1][4], auto S) {
            constexpr int s = decltype(S)::value;
            static_for<0, 8>([&](auto K) {                                       // (scalar image base + ONE 32-bit lane offset for every image)
                constexpr int k = decltype(K)::value;
                if constexpr ((k * 12) / 8 == s) dst[0][k >> 1].p[k & 1] = *reinterpret_cast<const u4 *>(img + (foff + 2048u * (k >> 1) + 1024u * (k & 1)));
            });
        };
        {
            load_wfrag_auto<SP>(a.img11, a.W11e, 384, wv, lane, fa[0]);
            load_wfrag_auto<SP>(a.img12, a.W12, TM_H, wv, lane, fb[0]);
            const f4 b12 = ld4(a.b12 + ncol), b13 = ld4(a.b13 + ncol);
            const f4 g4 = ld4(a.g3 + 4 * c32), be4 = ld4(a.be3 + 4 * c32);
            if (tid < TM_TILE) s_idx[tid] = a.E_idx[(size_t)i * TM_KS + tid];
            float *tile_g = a.hE + (size_t)i * TM_KS * TM_H;
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) e_cur[rb] = ld4(tile_g + (eoff + 16 * rb * TM_H));
            __syncthreads();
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) store_split<SP>(tE, 16 * rb + m, c4, e_cur[rb]);
            f4 gai = ld4(a.P + (size_t)i * 256 + ucol), gcj[3];
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                const int j = s_idx[16 * rb + m];
                gcj[rb] = ld4(a.P + (size_t)(j < 0 ? i : j) * 256 + 128 + ncol);
            }
            g0 = ld4(b.P + (size_t)i * 256 + ucol);
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                const int j0 = s_idx[16 * rb + m];
                gj[rb] = ld4(b.P + (size_t)(j0 < 0 ? i : j0) * 256 + 128 + ncol);
            }
            mi = b.mask[i];
            if (tid < TM_TILE) {                          // (only REQUESTED here; the product is formed in the message phase — a
                const int j = s_idx[tid];                 //  use here would wait for every load above, in front of GEMM 1)
                nma = b.mask[j < 0 ? i : j];
            }
            __syncthreads();
            f4 acc[3][1];
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) acc[rb][0] = gai + gcj[rb];
            mma_tile_split<SP, 4, 1, 3, TM_TILE, 256, 4, 0, true, TM_EDGE_PF>(tE, fa, acc, lane);
            {
                f4 g[3];
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) g[rb] = gelu4(acc[rb][0]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) store_split<SP>(tX, 16 * rb + m, c4, g[rb]);
            }
            __syncthreads();
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) acc[rb][0] = b12;
            mma_tile_split_ride<SP, 4, 3, TM_EDGE_PF>(tX, fb, acc, lane, [&](auto S) { ride_into(a.img13, fa, S); });   // W11 is done with: W13 for GEMM 3
            {
                f4 g[3];
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) g[rb] = gelu4(acc[rb][0]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rb = 0; rb < 3; ++rb) store_split<SP>(tY, 16 * rb + m, c4, g[rb]);
            }
            __syncthreads();
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) acc[rb][0] = b13;
            mma_tile_split_ride<SP, 4, 3, TM_EDGE_PF>(tY, fa, acc, lane, [&](auto S) { ride_into(b.imgp1, fb, S); });   // W12 is done with: the message pass's W1 (its K order: perm_c4)
            bias2 = ld4(b.b2 + ncol);
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                const f4 v = e_cur[rb] + acc[rb][0];                             // residual on the fp32 tile
                st4(tO + chunk_off(16 * rb + m, c4), v);
                row_stats_partial16(v, &s_stat[16 * rb + m][2 * wv], q);
            }
            __syncthreads();                                                     // tE free, tO + stats complete
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const int row = 6 * wv + 2 * it + (lane >> 5);
                float mean = 0.f, rstd = 1.f;
                row_stats_finish8d(&s_stat[row][0], lane, mean, rstd);
                const f4 x4 = ld4(tO + chunk_off(row, c32));
                const f2 s01 = f2{g4.x, g4.y} * rstd, s23 = f2{g4.z, g4.w} * rstd;
                const f2 t01 = __builtin_elementwise_fma(f2{-mean, -mean}, s01, f2{be4.x, be4.y});
                const f2 t23 = __builtin_elementwise_fma(f2{-mean, -mean}, s23, f2{be4.z, be4.w});
                const f2 y01 = __builtin_elementwise_fma(f2{x4.x, x4.y}, s01, t01), y23 = __builtin_elementwise_fma(f2{x4.z, x4.w}, s23, t23);
                const f4 y = f4{y01.x, y01.y, y23.x, y23.y};
                yrow[it] = s_idx[row] >= 0 ? y : f4{0.f, 0.f, 0.f, 0.f};         // rows without a neighbour stay zero
                st4(tile_g + (soff + 2 * it * TM_H), yrow[it]);                  // the later kernels read the updated tile from HBM
            }
        }
        // ---- message pass of the next layer on the SAME tile (msg8_rp_kernel) ------------------------------
        {
            if (tid < TM_TILE) s_ma[tid] = s_idx[tid] < 0 ? 0.f : (DEC ? 1.f : mi * nma);
            const int prow = 6 * wv + (lane >> 5), pc = lane & 31;              // the row layout the tile was just produced in
#pragma unroll
            for (int it = 0; it < 3; ++it) store_split<SP>(tE, prow + 2 * it, perm_c4(pc), yrow[it]);       // the message pass's K order
            __syncthreads();                                                     // e planes + s_ma complete; tO (= tA) consumed
            f4 acc[3][1];
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) acc[rb][0] = DEC ? gj[rb] : g0 + gj[rb];
            mma_tile_split_ride<SP, 4, 3, TM_MSG_PF>(tE, fb, acc, lane, [&](auto S) { ride_into(b.imgp2, fa, S); });     // W13 is done with: the message pass's W2
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                f4 v = acc[rb][0];
                if (DEC) v = g0 + mi * v;
                store_split<SP>(tX, 16 * rb + m, perm_c4(c4), gelu4(v));
            }
            __syncthreads();
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) acc[rb][0] = bias2;
            mma_tile_split<SP, 4, 1, 3, TM_TILE, 256, 4, 0, true, TM_MSG_PF>(tX, fa, acc, lane);
            f4 tot = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                const float ma = s_ma[16 * rb + m];
                const f4 g = gelu4(acc[rb][0]);
                tot = f4{__builtin_fmaf(g.x, ma, tot.x), __builtin_fmaf(g.y, ma, tot.y), __builtin_fmaf(g.z, ma, tot.z), __builtin_fmaf(g.w, ma, tot.w)};
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float x = tot[c];
                x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111, 0xf, 0xf, true));
                x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x112, 0xf, 0xf, true));
                x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x114, 0xf, 0xf, true));
                x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x118, 0xf, 0xf, true));
                tot[c] = x;
            }
            if (m == 15) st4(b.Ssum + (size_t)i * TM_H + ucol, tot);
            if (wv == 2) {
                float c = lane < TM_TILE ? s_ma[lane] : 0.f;
#define TM_DPP_ADD(ctrl, row_mask, bc)                                                                  \
                c += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c), ctrl, row_mask, 0xf, bc));
                TM_DPP_ADD(0x111, 0xf, true)
                TM_DPP_ADD(0x112, 0xf, true)
                TM_DPP_ADD(0x114, 0xf, true)
                TM_DPP_ADD(0x118, 0xf, true)
                TM_DPP_ADD(0x142, 0xa, false)
                TM_DPP_ADD(0x143, 0xc, false)
#undef TM_DPP_ADD
                if (lane == 63) b.cnt[i] = c;
            }
            __syncthreads();                                                     // (a further tile of this workgroup reuses every buffer)
        }
    }
}

// The fused form is used when every workgroup has at most one tile and the fragment images exist (f16x2 handles).
bool edge_msg_fusable(int mode, int64_t T) { return mode == TM_MM_F16X2 && T > 0 && T <= (int64_t)tm_num_cus(); }

int launch_edge_msg_fused(const EncW &e, const float *P_edge, float *hE, const int32_t *E_idx, bool dec, const float *W1e, int ld1,
                          const float *W2, const float *b2, const float *P_msg, const float *mask, int64_t T, float *Ssum, float *cnt,
                          hipStream_t st) {
    EdgeArgsB a{e.W11 + 128, e.W12, e.b12, e.W13, e.b13, e.norm3_w, e.norm3_b, P_edge, hE, E_idx, (int)T,
                tm_find_wimg(e.W11 + 128), tm_find_wimg(e.W12), tm_find_wimg(e.W13),
                tm_find_wimgp(e.W11 + 128), tm_find_wimgp(e.W12), tm_find_wimgp(e.W13)};
    MsgArgsB b{W1e, ld1, W2, b2, P_msg, hE, E_idx, mask, Ssum, cnt, (int)T, tm_find_wimg(W1e), tm_find_wimg(W2), tm_find_wimgp(W1e), tm_find_wimgp(W2), 0};
    const int64_t cap = tm_num_cus();
    const int grid = (int)(T < cap ? T : cap);
    tm_prof_begin("edge_msg_fused", st);
    if (dec) edge_msg_fused_kernel<SplitH2, true><<<grid, 512, 0, st>>>(a, b);
    else edge_msg_fused_kernel<SplitH2, false><<<grid, 512, 0, st>>>(a, b);
    tm_prof_end(st);
    return tm_check_launch("edge_msg_fused");
}
