// corr_fast_body.h — one 128 x 128 tile chain of the fp32 patch correlation + arg-max (K4 + K5), as a device function.
//
// Two kernels run it: corr_argmax_fast_kernel (corr_argmax.hip: the whole correlation, or only the q-tiles of one class) and
// attention_overlap_kernel (attention.hip: the q-tiles WITHOUT masked positions, side by side with the serial recurrence of
// the masked ones).  Same code, same arithmetic, same partial (max, arg-max) layout in both.
#pragma once
#include "ipsr_common.h"

namespace ipsr {
namespace corr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;   // k rows (patches) per workgroup tile
constexpr int BN = 128;   // q columns (reference positions) per workgroup tile
constexpr int BK = 16;    // channels per LDS stage (generic kernel)
constexpr int FBK = 16;   // channels per LDS stage (fast kernel: direct-to-LDS loads)
constexpr int FNBUF = 4;  // LDS ring slots; the DMA of stage s+3 is issued while stage s computes
constexpr int NTHREADS = 256;
constexpr int FAST_LDS_FLOATS = FNBUF * 2 * FBK * BM + 2 * 2 * 32 * 2;   // operand ring + the final arg-max exchange

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) float* ldsf_t;      // the body addresses its tiles as LDS, whoever declared the array
typedef __attribute__((address_space(3))) int* ldsi_t;

// Which q-tiles (128 reference positions each) a launch works on.  The masked positions are only known on the device, so every
// workgroup classifies its own tile: pass 0 = every tile, 1 = only tiles that hold a masked position, 2 = only tiles that hold
// none.  The coherent-attention recurrence needs the arg-max of the masked positions alone (models/IPSRFunction.py:105-125):
// pass 1 runs first, pass 2 shares a launch with the recurrence (attention.hip).
struct TileFilter {
    int pass;
    const int32_t* mpi;       // masked positions [M], or one row of `stride` per sample
    int stride;
    const int32_t* mcount;    // per-sample counts (NULL: M for every sample)
    int M;
};

// true when this workgroup's tile belongs to the launch's class.  Uniform for the workgroup; contains a barrier.
__device__ __forceinline__ bool tile_selected(const TileFilter& tf, int b, int qt, int N)
{
    if (tf.pass == 0) return true;
    int M = tf.M;
    if (tf.mcount) { M = tf.mcount[b]; M = M < 0 ? 0 : (M > tf.M ? tf.M : M); }
    const int32_t* mpi = tf.mpi + (size_t)b * tf.stride;
    int hit = 0;
    for (int l = threadIdx.x; l < M; l += NTHREADS) hit |= (mpi_at(mpi, l, N) / BN) == qt;
    const bool masked = __syncthreads_or(hit) != 0;
    return masked == (tf.pass == 1);
}

// ---------------------------------------------------------------------------------------------------
// Fast path (N % 128 == 0, C % 16 == 0, 16-byte aligned operands): same tile, same arithmetic (every S[k][q] is
// still ONE fmaf chain over ascending channels), shaped by what was measured on gfx950 with this kernel
// (IPSR_CLOCK_PROBE build, s_memtime around the K loop, clock 2.39 GHz):
//   * the MFMA groups alone run at 64.4 cycles per v_mfma_f32_32x32x2_f32 (the 64-cycle peak);
//   * every other vector-memory/LDS instruction is ADDED to that at the SIMD level, co-resident waves do not hide it:
//     +7 cycles per LDS fragment instruction, +56 per global_load_lds piece (two workgroups per CU took exactly twice
//     the cycles of one; a dedicated loader wave only concentrated the DMA cost on one SIMD and lost 6 %).
// So the instruction stream is kept minimal and evenly spread:
//   * operand tiles go HBM/L2 -> LDS directly (global_load_lds_dwordx4: 1 KiB = two 512-byte tile rows per
//     wave-instruction, no staging VGPRs, no ds_write pass), 4 pieces per wave per stage, issued one at a time
//     between MFMA groups; 4-slot LDS ring, the DMA of stage s+3 is issued during stage s and the end-of-stage
//     wait is a COUNTED vmcnt (only stage s+1 must have landed) before a raw s_barrier — a __syncthreads() would
//     drain the ring (cdna_hip_programming.md §5 "Pipelining across barriers");
//   * a wave owns the 32x32 sub-tiles {wm*32, wm*32+64} x {wn*32, wn*32+64}: its two A (and two B) fragments of a
//     k-step are 256 bytes apart in LDS = ONE ds_read2st64_b32 with immediate offsets each, i.e. 2 LDS instructions
//     and no address arithmetic per 4 MFMAs; three fragment register sets in rotation, the reads of k-step kk+1
//     pinned in front of the MFMAs of kk.
//
// RAGGED: the operands are [C][ld] with ld a multiple of 128 and the columns [N, ld) zero (shift_sz > 1 window grids);
// patches k >= N are kept out of the arg-max and columns q >= N are not stored.
template <bool WRITE_S, bool RAGGED>
__device__ __forceinline__ void fast_tile_chain(ldsf_t lds,                  // the workgroup's __shared__ array, FAST_LDS_FLOATS
                                                const float* __restrict__ xn, const float* __restrict__ ref, int C, int N, int ld,
                                                int b, int qt, int ks, int ksplit, int ktiles, int kt_per_wg,
                                                float* __restrict__ S_out, float* __restrict__ pval, int32_t* __restrict__ pidx)
{
    const ldsf_t tiles = lds;                                  // [slot][A|B][FBK][128]
    const ldsf_t red_v = lds + FNBUF * 2 * FBK * BM;           // [wn][jn][32]
    const ldsi_t red_i = (ldsi_t)(red_v + 2 * 2 * 32);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    const int q0 = qt * BN;

    const float* A = xn + (size_t)b * C * ld;
    const float* R = ref + (size_t)b * C * ld;

    const int nstage = C / FBK;
    const int kt_lo = ks * kt_per_wg, kt_hi = min(ktiles, kt_lo + kt_per_wg);
    // start from the first row this lane will see, value -inf (see the generic kernel)
    float best[2] = {-INFINITY, -INFINITY};
    int bidx[2] = {kt_lo * BM + wm * 32 + 4 * h, kt_lo * BM + wm * 32 + 4 * h};
    if (RAGGED && bidx[0] >= N) bidx[0] = bidx[1] = kt_lo * BM;

    // LDS-DMA: a stage = 2 operands x FBK rows x 512 B = 2*FBK/2 = 16 pieces of 1 KiB (2 rows); wave w owns row
    // pairs w and w+4 of A and of B -> NP = 4 pieces per wave per stage.  Lane -> (row parity, 16-byte column).
    constexpr int NP = FBK / 4;
    const int dma_row = lane >> 5, dma_col = (lane & 31) * 4;

    for (int kt = kt_lo; kt < kt_hi; ++kt) {
        const int k0 = kt * BM;
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

        // piece p: p < NP/2 -> A rows, else B rows (index the __shared__ array directly: the builtin needs a pointer the compiler
        // KNOWS is LDS).  One pointer per piece, advanced by a constant per stage; the prefetch is UNCONDITIONAL — past the last
        // stage it re-reads the last one into a slot nobody reads again — so the loop body has no branch and the number of DMAs
        // in flight is the same in every iteration: one constant vmcnt (measured on the Winograd GEMM, the same pipeline: -8 %).
        const float* gp[NP];
        int loff[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const bool isA = p < NP / 2;
            const int pair = wave + 4 * (isA ? p : p - NP / 2);
            loff[p] = (isA ? 0 : FBK * BM) + pair * 2 * BM;
            const size_t row = (size_t)2 * pair + dma_row;
            gp[p] = isA ? (A + row * ld + k0 + dma_col) : (R + row * ld + q0 + dma_col);
        }
        const size_t stage_stride = (size_t)FBK * ld;
        auto dma_piece = [&](int p, int slot, bool more) {
            __builtin_amdgcn_global_load_lds((gptr_t)gp[p], (lptr_t)&lds[slot * (2 * FBK * BM) + loff[p]], 16, 0, 0);
            gp[p] += more ? stage_stride : 0;
        };
        constexpr int AHEAD = FNBUF - 1;
        int issued = 0, pf_slot = 0;
        // prologue: stages 0,1,2 in flight; stage 0 must have landed before the first compute
#pragma unroll
        for (int a2 = 0; a2 < AHEAD; ++a2) {
#pragma unroll
            for (int p = 0; p < NP; ++p) dma_piece(p, pf_slot, issued + 1 < nstage);
            ++issued;
            pf_slot = (pf_slot + 1) & (FNBUF - 1);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * NP) : "memory");
        __builtin_amdgcn_s_barrier();

        for (int s = 0; s < nstage; ++s) {
            const int cur = s & (FNBUF - 1);
            // slot (s+3)%4 was last read in stage s-1, which every wave has left (barrier); its pieces are spread
            // over this stage's k-steps
            const bool more = issued + 1 < nstage;
            const ldsf_t ta = tiles + (size_t)cur * (2 * FBK * BM) + h * BM + wm * 32 + r;
            const ldsf_t tb = tiles + (size_t)cur * (2 * FBK * BM) + FBK * BM + h * BM + wn * 32 + r;
            float fa0[3], fa1[3], fb0[3], fb1[3];
            fa0[0] = ta[0]; fa1[0] = ta[64]; fb0[0] = tb[0]; fb1[0] = tb[64];
#pragma unroll
            for (int kk = 0; kk < FBK / 2; ++kk) {
                // THREE fragment sets in rotation: the set refilled now (for kk+1) was last read by the MFMAs of kk-2
                const int cs = kk % 3, ns = (kk + 1) % 3;
                if (kk + 1 < FBK / 2) {
                    const int ro = (kk + 1) * 2 * BM;
                    fa0[ns] = ta[ro]; fa1[ns] = ta[ro + 64]; fb0[ns] = tb[ro]; fb1[ns] = tb[ro + 64];
                }
                if ((kk % ((FBK / 2) / NP)) == 0) dma_piece(kk / ((FBK / 2) / NP), pf_slot, more);
                __builtin_amdgcn_sched_barrier(0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[cs], fb0[cs], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[cs], fb1[cs], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[cs], fb0[cs], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[cs], fb1[cs], acc[1][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            ++issued;
            pf_slot = (pf_slot + 1) & (FNBUF - 1);
            // stage s+1 has landed once all but the two youngest stages' pieces are done
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * NP) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        // the redundant tail prefetches wrote slots that the next k-tile's prologue refills: they must have landed first
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        // epilogue of this k-tile: fold 2x16 rows into the running (max, argmax) of the lane's 2 columns.
        // Rows are visited in ascending k and only a strictly larger value replaces -> lowest k on ties.
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            const int q = q0 + wn * 32 + jn * 64 + r;
#pragma unroll
            for (int im = 0; im < 2; ++im) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int k = k0 + wm * 32 + im * 64 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    const float v = acc[im][jn][e];
                    if (!RAGGED || k < N) {
                        if (takes_over(v, best[jn])) { best[jn] = v; bidx[jn] = k; }
                        if (WRITE_S) { if (!RAGGED || q < N) S_out[((size_t)b * N + k) * N + q] = v; }
                    }
                }
            }
        }
    }

    // merge the two lane halves (rows 4h..), then the two waves stacked along k (lexicographic (value, idx))
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
        const float ov = __shfl_xor(best[jn], 32);
        const int oi = __shfl_xor(bidx[jn], 32);
        if (better(ov, oi, best[jn], bidx[jn])) { best[jn] = ov; bidx[jn] = oi; }
    }
    if (wm == 1 && h == 0) {
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) { red_v[(wn * 2 + jn) * 32 + r] = best[jn]; red_i[(wn * 2 + jn) * 32 + r] = bidx[jn]; }
    }
    __syncthreads();
    if (wm == 0 && h == 0) {
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            const float ov = red_v[(wn * 2 + jn) * 32 + r];
            const int oi = red_i[(wn * 2 + jn) * 32 + r];
            if (better(ov, oi, best[jn], bidx[jn])) { best[jn] = ov; bidx[jn] = oi; }
            const int q = q0 + wn * 32 + jn * 64 + r;
            if (!RAGGED || q < N) {
                pval[((size_t)b * ksplit + ks) * N + q] = best[jn];
                pidx[((size_t)b * ksplit + ks) * N + q] = bidx[jn];
            }
        }
    }
}


}  // namespace corr
}  // namespace ipsr
