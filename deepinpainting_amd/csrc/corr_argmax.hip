// corr_argmax.hip — K4 + K5: patch x reference cross-correlation fused with the arg-max over patches.
//
// Reference: tmp1 = conv_enc(ref.relu4_3) (models/IPSRFunction.py:59) followed by MaxCoord.update_output
// (util/MaxCoord.py:16-28):
//      S[k][q] = sum_c xn[c][k] * ref[c][q]        (N x N, reduction over C)
//      ind[q]  = argmax_k S[k][q]  (lowest k on ties),   vmax[q] = max_k S[k][q]
// The reference materialises S (N*N fp32 per sample) and re-reads it for the max; here S lives only in
// MFMA accumulators and the running (max, argmax) is folded in the epilogue of every 128-row k-tile.
//
// This is the one genuine dense contraction of the layer (2*N*N*C flop per sample) and runs on the
// fp32-input matrix cores: v_mfma_f32_32x32x2_f32, exact fp32, 64 FLOP/clk/SIMD (157 TF peak).  Because
// that instruction is bit-for-bit an fmaf chain over its 2 k-steps (cdna_hip_programming.md §3), walking
// C in ascending order makes every S[k][q] the same single fmaf chain the oracle computes, so the
// arg-max indices match the CPU restatement exactly, not just within a tolerance.
//
// Tiling (64-wide waves, one wave per SIMD):
//   workgroup = 256 threads = 4 waves in a 2x2 grid; workgroup tile 128 (k) x 128 (q); each wave owns
//   64x64 = 2x2 MFMA tiles of 32x32 (64 accumulator VGPRs); BK = 16 channels per LDS stage.
//   Both operands are channel-major in HBM ([C][N]), which is exactly the MFMA operand order
//   (A[i][kk]: lane = i + 32*kk), so the LDS image is a plain copy of the global tile: 512-byte
//   coalesced row segments in, conflict-free ds_read_b32 out (32 consecutive floats per half-wave).
//   Double-buffered LDS, global loads for stage s+1 are in flight while stage s feeds the MFMAs.
//   A 32x32 accumulator tile has its q column on the lane and 16 k rows in registers, so the arg-max
//   over k is 16 in-lane compares per tile, one cross-half exchange and one LDS exchange at the very end.
//
// Grid: (sample, q-tile, k-split) with an XCD-aware remap so that all workgroups of a sample — which
// share its xn and ref tiles — sit on one XCD's L2.  k-splits keep >= 2 workgroups per CU busy at the
// batch-8 / N=1024 size; their partial (max, argmax) are merged by a tiny second kernel.
#include "ipsr_common.h"

namespace ipsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;   // k rows (patches) per workgroup tile
constexpr int BN = 128;   // q columns (reference positions) per workgroup tile
constexpr int BK = 16;    // channels per LDS stage (generic kernel)
constexpr int FBK = 16;   // channels per LDS stage (fast kernel: direct-to-LDS loads)
constexpr int FNBUF = 4;  // LDS ring slots; the DMA of stage s+3 is issued while stage s computes
constexpr int NTHREADS = 256;

#ifdef IPSR_CLOCK_PROBE
// Diagnostic build only (never shipped): per-workgroup shader-clock / 100 MHz real-time stamps around the K loop,
// to read the clock the chip actually holds (MI355X_MICROARCH.md "DVFS give-back" item 6).
__device__ unsigned long long g_probe[2 * 8192];
#endif

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <bool FAST, bool WRITE_S>
__global__ void __launch_bounds__(NTHREADS, 2)
corr_argmax_kernel(const float* __restrict__ xn, const float* __restrict__ ref, int C, int N, int ld,
                   int qtiles, int ksplit, int ktiles, int kt_per_wg,
                   float* __restrict__ S_out, float* __restrict__ pval, int32_t* __restrict__ pidx)
{
    __shared__ __attribute__((aligned(16))) float As[2][BK][BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];
    __shared__ float red_v[2][2][32];
    __shared__ int red_i[2][2][32];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int ks = L % ksplit;
    const int qt = (L / ksplit) % qtiles;
    const int b = L / (ksplit * qtiles);
    const int q0 = qt * BN;

    const float* A = xn + (size_t)b * C * ld;
    const float* R = ref + (size_t)b * C * ld;

    // staging assignment: 16 rows x 32 float4 per operand tile = 512 float4, 2 per thread
    const int ld_row = tid >> 5;          // 0..7  (+8 for the second)
    const int ld_c4 = (tid & 31) * 4;     // column offset in floats

    const int nstage = (C + BK - 1) / BK;
    const int kt_lo = ks * kt_per_wg, kt_hi = min(ktiles, kt_lo + kt_per_wg);
    // start from the first row this lane will see (always < N: row 4h of a tile that exists), value -inf: a column whose
    // correlations are all -inf (or that this lane never beats) still reports an in-range index
    float best[2] = {-INFINITY, -INFINITY};
    int bidx[2] = {kt_lo * BM + wm * 64 + 4 * h, kt_lo * BM + wm * 64 + 4 * h};
    if (bidx[0] >= N) bidx[0] = bidx[1] = kt_lo * BM;

    for (int kt = kt_lo; kt < kt_hi; ++kt) {
        const int k0 = kt * BM;
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

        float4 ra[2], rb[2];
        auto gload = [&](int s) {
            const int c0 = s * BK;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = c0 + ld_row + 8 * i;
                if (FAST) {
                    ra[i] = *reinterpret_cast<const float4*>(A + (size_t)c * ld + k0 + ld_c4);
                    rb[i] = *reinterpret_cast<const float4*>(R + (size_t)c * ld + q0 + ld_c4);
                } else {
                    float va[4], vb[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int kc = k0 + ld_c4 + e, qc = q0 + ld_c4 + e;
                        va[e] = (c < C && kc < N) ? A[(size_t)c * ld + kc] : 0.0f;
                        vb[e] = (c < C && qc < N) ? R[(size_t)c * ld + qc] : 0.0f;
                    }
                    ra[i] = make_float4(va[0], va[1], va[2], va[3]);
                    rb[i] = make_float4(vb[0], vb[1], vb[2], vb[3]);
                }
            }
        };
        auto sstore = [&](int buf) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                *reinterpret_cast<float4*>(&As[buf][ld_row + 8 * i][ld_c4]) = ra[i];
                *reinterpret_cast<float4*>(&Bs[buf][ld_row + 8 * i][ld_c4]) = rb[i];
            }
        };

        gload(0);
        sstore(0);
        __syncthreads();
        for (int s = 0; s < nstage; ++s) {
            const int cur = s & 1;
            if (s + 1 < nstage) gload(s + 1);
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                const float a0 = As[cur][kk * 2 + h][wm * 64 + r];
                const float a1 = As[cur][kk * 2 + h][wm * 64 + 32 + r];
                const float b0 = Bs[cur][kk * 2 + h][wn * 64 + r];
                const float b1 = Bs[cur][kk * 2 + h][wn * 64 + 32 + r];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
            if (s + 1 < nstage) sstore(cur ^ 1);
            __syncthreads();
        }

        // epilogue of this k-tile: fold 2x16 rows into the running (max, argmax) of the lane's 2 columns.
        // Rows are visited in ascending k and only a strictly larger value replaces -> lowest k on ties.
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            const int q = q0 + wn * 64 + jn * 32 + r;
#pragma unroll
            for (int im = 0; im < 2; ++im) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int k = k0 + wm * 64 + im * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    const float v = acc[im][jn][e];
                    if (FAST || k < N) {
                        if (takes_over(v, best[jn])) { best[jn] = v; bidx[jn] = k; }
                        if (WRITE_S) { if (FAST || q < N) S_out[((size_t)b * N + k) * N + q] = v; }
                    }
                }
            }
        }
    }

    // merge the two lane halves (rows 4h..), then the two waves stacked along k
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
        const float ov = __shfl_xor(best[jn], 32);
        const int oi = __shfl_xor(bidx[jn], 32);
        if (better(ov, oi, best[jn], bidx[jn])) { best[jn] = ov; bidx[jn] = oi; }
    }
    if (wm == 1 && h == 0) {
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) { red_v[wn][jn][r] = best[jn]; red_i[wn][jn][r] = bidx[jn]; }
    }
    __syncthreads();
    if (wm == 0 && h == 0) {
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            const float ov = red_v[wn][jn][r];
            const int oi = red_i[wn][jn][r];
            if (better(ov, oi, best[jn], bidx[jn])) { best[jn] = ov; bidx[jn] = oi; }
            const int q = q0 + wn * 64 + jn * 32 + r;
            if (q < N) {
                pval[((size_t)b * ksplit + ks) * N + q] = best[jn];
                pidx[((size_t)b * ksplit + ks) * N + q] = bidx[jn];
            }
        }
    }
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
__global__ void __launch_bounds__(NTHREADS, 2)
corr_argmax_fast_kernel(const float* __restrict__ xn, const float* __restrict__ ref, int C, int N, int ld,
                        int qtiles, int ksplit, int ktiles, int kt_per_wg,
                        float* __restrict__ S_out, float* __restrict__ pval, int32_t* __restrict__ pidx)
{
    // one array (a second __shared__ object next to an LDS-DMA target can make hipcc drain vmcnt early)
    __shared__ __attribute__((aligned(16))) float lds[FNBUF * 2 * FBK * BM + 2 * 2 * 32 * 2];
    float* const tiles = lds;                                  // [slot][A|B][FBK][128]
    float* const red_v = lds + FNBUF * 2 * FBK * BM;           // [wn][jn][32]
    int* const red_i = reinterpret_cast<int*>(red_v + 2 * 2 * 32);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int ks = L % ksplit;
    const int qt = (L / ksplit) % qtiles;
    const int b = L / (ksplit * qtiles);
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

#ifdef IPSR_CLOCK_PROBE
    const unsigned long long pt0 = __builtin_amdgcn_s_memtime(), pr0 = __builtin_amdgcn_s_memrealtime();
#endif
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
            const float* ta = tiles + (size_t)cur * (2 * FBK * BM) + h * BM + wm * 32 + r;
            const float* tb = tiles + (size_t)cur * (2 * FBK * BM) + FBK * BM + h * BM + wn * 32 + r;
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
#ifdef IPSR_CLOCK_PROBE
    if (tid == 0 && blockIdx.x < 8192) {
        g_probe[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - pt0;
        g_probe[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - pr0;
    }
#endif

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

// ---------------------------------------------------------------------------------------------------
// bf16 variant (BASELINE config 5: "CDNA4 bf16 MFMA for patch-corr"): the same contraction, arg-max and k-split merge
// on v_mfma_f32_32x32x16_bf16 (bf16 operands, fp32 accumulate).  Opt-in only (ipsr_forward_bf16corr /
// ipsr_corr_argmax_bf16): the operands are ROUNDED to bf16, so an arg-max can move where the two best patches are
// closer than the rounding error — the measured agreement with the fp32 kernel is reported by bench.py and asserted in
// tests/test_gpu_parity.py; the fp32 kernel stays the default and the only one the parity contract is stated on.
//
// Operand layout.  The instruction wants, per lane, 8 CONSECUTIVE reduction elements of one row (A[i][8g..8g+7],
// lane = i + 32g).  Channel-major [C][N] has them N apart, so both operands are first re-packed (pack_bf16_k8_kernel)
// into [C/8][ld][8] bf16: the 8 channels of a group interleaved per position.  A lane's fragment is then ONE 16-byte
// load, a half-wave reads 512 contiguous bytes, and the data is half the size of the fp32 operands.
//
// The MFMA part is 16x shorter than in fp32 (C=512: 128 instructions of 32 cycles per 128x128 tile and wave), so the
// kernel is bound by operand delivery, not by the matrix pipe.  It therefore skips LDS altogether: every wave streams
// its own A and B fragments from L2 straight into registers through a PF-deep software pipeline (no barrier, no LDS
// image, no DMA bookkeeping); the 2x re-read of a tile by the two waves that share it is served by L1/L2.
constexpr int PF = 4;     // k-steps (16 channels each) of fragment loads in flight per wave

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(256) pack_bf16_k8_kernel(const float* __restrict__ src, int C, int ld, int C8, uint4* __restrict__ dst)
{
    // one thread per (channel group, position): 8 strided fp32 reads (coalesced across positions), one 16-byte store
    const int n = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y, b = blockIdx.z;
    if (n >= ld) return;
    const float* s = src + ((size_t)b * C + (size_t)g * 8) * ld + n;
    unsigned short h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = (g * 8 + e < C) ? s[(size_t)e * ld] : 0.0f;
        h[e] = __builtin_bit_cast(unsigned short, (__bf16)v);          // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
    }
    uint4 o;
    o.x = h[0] | ((unsigned)h[1] << 16); o.y = h[2] | ((unsigned)h[3] << 16);
    o.z = h[4] | ((unsigned)h[5] << 16); o.w = h[6] | ((unsigned)h[7] << 16);
    dst[((size_t)b * C8 + g) * ld + n] = o;
}

template <bool RAGGED>
__global__ void __launch_bounds__(NTHREADS, 2)
corr_argmax_bf16_kernel(const uint4* __restrict__ xnp, const uint4* __restrict__ refp, int C8, int N, int ld,
                        int qtiles, int ksplit, int ktiles, int kt_per_wg, float* __restrict__ pval, int32_t* __restrict__ pidx)
{
    __shared__ float red_v[2 * 2 * 32];
    __shared__ int red_i[2 * 2 * 32];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int ks = L % ksplit;
    const int qt = (L / ksplit) % qtiles;
    const int b = L / (ksplit * qtiles);
    const int q0 = qt * BN;

    const uint4* A = xnp + (size_t)b * C8 * ld;
    const uint4* R = refp + (size_t)b * C8 * ld;
    const int nks = C8 / 2;                                  // k-steps of 16 channels; a multiple of PF (host-checked)
    const int kt_lo = ks * kt_per_wg, kt_hi = min(ktiles, kt_lo + kt_per_wg);
    float best[2] = {-INFINITY, -INFINITY};
    int bidx[2] = {kt_lo * BM + wm * 64 + 4 * h, kt_lo * BM + wm * 64 + 4 * h};
    if (RAGGED && bidx[0] >= N) bidx[0] = bidx[1] = kt_lo * BM;

    // B fragments do not depend on the k-tile: positions q0 + wn*64 + {0,32} + r, channel group 2s + h
    const uint4* rb = R + (size_t)h * ld + q0 + wn * 64 + r;
    for (int kt = kt_lo; kt < kt_hi; ++kt) {
        const int k0 = kt * BM;
        const uint4* ra = A + (size_t)h * ld + k0 + wm * 64 + r;
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
        uint4 fa[PF][2], fb[PF][2];
        auto load = [&](int d, int s) {
            const size_t o = (size_t)2 * s * ld;
            fa[d][0] = ra[o]; fa[d][1] = ra[o + 32];
            fb[d][0] = rb[o]; fb[d][1] = rb[o + 32];
        };
#pragma unroll
        for (int d = 0; d < PF; ++d) load(d, d);
        for (int s0 = 0; s0 < nks; s0 += PF) {
#pragma unroll
            for (int d = 0; d < PF; ++d) {
                const bf16x8 a0 = __builtin_bit_cast(bf16x8, fa[d][0]), a1 = __builtin_bit_cast(bf16x8, fa[d][1]);
                const bf16x8 b0 = __builtin_bit_cast(bf16x8, fb[d][0]), b1 = __builtin_bit_cast(bf16x8, fb[d][1]);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
                load(d, min(s0 + d + PF, nks - 1));          // branch-free refill (the tail re-reads the last step, unused)
            }
        }
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
#pragma unroll
            for (int im = 0; im < 2; ++im) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int k = k0 + wm * 64 + im * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    const float v = acc[im][jn][e];
                    if (!RAGGED || k < N) {
                        if (takes_over(v, best[jn])) { best[jn] = v; bidx[jn] = k; }
                    }
                }
            }
        }
    }
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
            const int q = q0 + wn * 64 + jn * 32 + r;
            if (!RAGGED || q < N) {
                pval[((size_t)b * ksplit + ks) * N + q] = best[jn];
                pidx[((size_t)b * ksplit + ks) * N + q] = bidx[jn];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// shift_sz > 1: correlation of p x p windows as sums of shifted diagonals of the 1x1 correlation R = x^T ref
//     S[k'][q'] = inv[k'] * sum_{dy,dx} R[(ky+dy)*w + kx+dx][(qy+dy)*w + qx+dx]
// (oracle: window_corr_argmax).  R [B][hw][hw] comes from corr_argmax_*_kernel<WRITE_S> on the RAW features: 2*N*N*C flop
// instead of the 2*N'*N'*C*p*p of contracting unfolded patches (8x fewer at p = 3), and nothing is unfolded.  This kernel is
// the p*p-tap stencil + running arg-max: thread = window q', loop over a range of k' (k-split for parallelism; the partials go
// to the same merge as the correlation kernel's).  The taps of one (k', q') are p*p coalesced loads (neighbouring q' read
// neighbouring addresses); every R element is touched p*p times in all, from L2 / Infinity Cache.
__global__ void __launch_bounds__(256) window_corr_argmax_kernel(const float* __restrict__ R, const float* __restrict__ inv, int hw, int w,
                                                                 int nW, int Np, int patch, int ksplit, int kper,
                                                                 float* __restrict__ pval, int32_t* __restrict__ pidx)
{
    const int q = blockIdx.x * 256 + threadIdx.x, ks = blockIdx.y, b = blockIdx.z;
    if (q >= Np) return;
    const float* Rb = R + (size_t)b * hw * hw + (size_t)(q / nW) * w + q % nW;      // column base of this window
    const float* invb = inv + (size_t)b * Np;
    const int k_lo = ks * kper, k_hi = min(Np, k_lo + kper);
    float best = -INFINITY;
    int bidx = k_lo;
    int ky = k_lo / nW, kx = k_lo - ky * nW;
    for (int k = k_lo; k < k_hi; ++k) {
        const float* Rk = Rb + (size_t)(ky * w + kx) * hw;
        float acc = 0.0f;
        bool first = true;
        for (int dy = 0; dy < patch; ++dy)
            for (int dx = 0; dx < patch; ++dx) {
                const int d = dy * w + dx;
                const float v = Rk[(size_t)d * hw + d];
                acc = first ? v : acc + v;
                first = false;
            }
        const float sv = invb[k] * acc;
        if (takes_over(sv, best)) { best = sv; bidx = k; }
        if (++kx == nW) { kx = 0; ++ky; }
    }
    pval[((size_t)b * ksplit + ks) * Np + q] = best;
    pidx[((size_t)b * ksplit + ks) * Np + q] = bidx;
}

__global__ void argmax_merge_kernel(const float* __restrict__ pval, const int32_t* __restrict__ pidx, int B, int N, int ksplit,
                                    int32_t* __restrict__ ind, float* __restrict__ vmax);

// partial buffers of the window arg-max: [B][ksplit][N'] values + indices
static void window_plan(int B, int Np, int* ksplit, int* kper)
{
    const int qblocks = cdiv(Np, 256) * B;
    int ks = 1;
    while (qblocks * ks < 1024 && ks < 64 && cdiv(Np, ks * 2) >= 32) ks *= 2;
    *kper = cdiv(Np, ks);
    *ksplit = cdiv(Np, *kper);
}

size_t window_corr_ws_bytes(int B, int Np)
{
    int ks, kper;
    window_plan(B, Np, &ks, &kper);
    return 2 * align_up((size_t)B * ks * Np * 4, 256) + 2 * align_up((size_t)B * Np * 4, 256) + 256;
}

int launch_window_corr_argmax(const float* R, const float* inv, int B, int h, int w, int patch, void* ws, size_t ws_bytes, hipStream_t st,
                              CorrPartials* partials)
{
    const int nW = w - patch + 1, Np = (h - patch + 1) * nW;
    int ks, kper;
    window_plan(B, Np, &ks, &kper);
    if (ws_bytes < window_corr_ws_bytes(B, Np)) return fail(IPSR_ERR_WORKSPACE, "window correlation: workspace %zu < %zu", ws_bytes, window_corr_ws_bytes(B, Np));
    Carver cv(ws, ws_bytes);
    float* pval = cv.take<float>((size_t)B * ks * Np);
    int32_t* pidx = cv.take<int32_t>((size_t)B * ks * Np);
    float* mval = cv.take<float>((size_t)B * Np);
    int32_t* midx = cv.take<int32_t>((size_t)B * Np);
    window_corr_argmax_kernel<<<dim3(cdiv(Np, 256), ks, B), 256, 0, st>>>(R, inv, h * w, w, nW, Np, patch, ks, kper, pval, pidx);
    if (int rc = check_launch("window_corr_argmax_kernel")) return rc;
    // fold the k-splits here, once: the stage kernel's consumers (one merge per 32x32 gather tile over K = C*p*p channels)
    // would otherwise re-fold the same 16 partials thousands of times
    argmax_merge_kernel<<<cdiv(B * Np, 256), 256, 0, st>>>(pval, pidx, B, Np, ks, midx, mval);
    partials->pval = mval;
    partials->pidx = midx;
    partials->ksplit = 1;
    return check_launch("argmax_merge_kernel");
}

// merge the k-split partials in ascending k order
__global__ void __launch_bounds__(256) argmax_merge_kernel(const float* __restrict__ pval, const int32_t* __restrict__ pidx,
                                                           int B, int N, int ksplit, int32_t* __restrict__ ind,
                                                           float* __restrict__ vmax)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * N) return;
    const int b = i / N, q = i - b * N;
    float bv = pval[((size_t)b * ksplit) * N + q];
    int bi = pidx[((size_t)b * ksplit) * N + q];
    for (int s = 1; s < ksplit; ++s) {
        const float v = pval[((size_t)b * ksplit + s) * N + q];
        const int ii = pidx[((size_t)b * ksplit + s) * N + q];
        if (better(v, ii, bv, bi)) { bv = v; bi = ii; }
    }
    ind[i] = bi;
    vmax[i] = bv;
}

static void plan(int B, int N, int* qtiles, int* ktiles, int* ksplit, int* kt_per_wg)
{
    *qtiles = cdiv(N, BN);
    *ktiles = cdiv(N, BM);
    // The chip runs 512 workgroups at a time (256 CUs x 2).  Pick the k-tiles per workgroup that minimises
    // (rounds of 512 workgroups) x (k-tiles per workgroup + a fixed per-workgroup cost of about a quarter tile); on
    // ties the larger one (fewer partials).  N=1024, B=8: 8 splits of 1 tile (512 workgroups).  The 3x3-patch window
    // grid N'=3844, B=4: 31 q-tiles x 4 splits of 8,8,8,7 tiles = 496 workgroups in ONE round (5 splits of 7 would be
    // 620 workgroups = two rounds, the second one a fifth full).
    const int base = B * *qtiles;
    long best_cost = -1;
    int best = *ktiles;
    for (int kpw = *ktiles; kpw >= 1; --kpw) {
        const int ks = cdiv(*ktiles, kpw);
        const long rounds = cdiv(base * ks, 512);
        const long cost = rounds * (4L * kpw + 1);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = kpw; }
    }
    *kt_per_wg = best;
    *ksplit = cdiv(*ktiles, best);
}

// bf16 path: the packed operands ([C/8][ld][8] bf16 each) live in the workspace behind the partials
static size_t bf16_pack_bytes(int B, int C, int ld) { return align_up((size_t)B * ((C + 7) / 8) * ld * 16, 256); }

bool corr_bf16_supported(int C, int ld) { return C % (16 * PF) == 0 && ld % BM == 0; }

size_t corr_argmax_bf16_ws_bytes(int B, int C, int N, int ld)
{
    if (ld <= 0) ld = N;
    return corr_argmax_ws_bytes(B, C, N) + 2 * bf16_pack_bytes(B, C, ld);
}

size_t corr_argmax_ws_bytes(int B, int C, int N)
{
    (void)C;
    int qt, kt, ks, kpw;
    plan(B, N, &qt, &kt, &ks, &kpw);
    return 2 * align_up((size_t)B * ks * N * 4, 256) + 256;
}

int launch_corr_argmax(const float* xn, const float* ref, int B, int C, int N, int32_t* ind, float* vmax,
                       float* S_out, void* ws, size_t ws_bytes, hipStream_t st, CorrPartials* partials, int ld)
{
    if (ld <= 0) ld = N;
    int qt, kt, ks, kpw;
    plan(B, N, &qt, &kt, &ks, &kpw);
    if (ws_bytes < corr_argmax_ws_bytes(B, C, N))
        return fail(IPSR_ERR_WORKSPACE, "ipsr_corr_argmax: workspace %zu < %zu", ws_bytes, corr_argmax_ws_bytes(B, C, N));
    Carver cv(ws, ws_bytes);
    float* pval = cv.take<float>((size_t)B * ks * N);
    int32_t* pidx = cv.take<int32_t>((size_t)B * ks * N);
    // fast path: whole 128-column tiles in memory (ld), whole 16-channel stages, 16-byte aligned rows
    const bool fast = (ld % BM == 0) && (ld >= qt * BN) && (C % FBK == 0) &&
                      ((reinterpret_cast<uintptr_t>(xn) | reinterpret_cast<uintptr_t>(ref)) & 15u) == 0;
    const int grid = B * qt * ks;
    profile_mark_start(st);
    if (fast && ld == N) {
        if (S_out) corr_argmax_fast_kernel<true, false><<<grid, NTHREADS, 0, st>>>(xn, ref, C, N, ld, qt, ks, kt, kpw, S_out, pval, pidx);
        else corr_argmax_fast_kernel<false, false><<<grid, NTHREADS, 0, st>>>(xn, ref, C, N, ld, qt, ks, kt, kpw, S_out, pval, pidx);
    } else if (fast) {
        if (S_out) corr_argmax_fast_kernel<true, true><<<grid, NTHREADS, 0, st>>>(xn, ref, C, N, ld, qt, ks, kt, kpw, S_out, pval, pidx);
        else corr_argmax_fast_kernel<false, true><<<grid, NTHREADS, 0, st>>>(xn, ref, C, N, ld, qt, ks, kt, kpw, S_out, pval, pidx);
    } else {
        if (S_out) corr_argmax_kernel<false, true><<<grid, NTHREADS, 0, st>>>(xn, ref, C, N, ld, qt, ks, kt, kpw, S_out, pval, pidx);
        else corr_argmax_kernel<false, false><<<grid, NTHREADS, 0, st>>>(xn, ref, C, N, ld, qt, ks, kt, kpw, S_out, pval, pidx);
    }
    profile_mark_stop(st);
    if (int rc = check_launch("corr_argmax_kernel")) return rc;
    if (partials) {
        partials->pval = pval;
        partials->pidx = pidx;
        partials->ksplit = ks;
        return IPSR_OK;
    }
    argmax_merge_kernel<<<cdiv(B * N, 256), 256, 0, st>>>(pval, pidx, B, N, ks, ind, vmax);
    return check_launch("argmax_merge_kernel");
}

// bf16 MFMA variant of launch_corr_argmax: xn / ref are the SAME fp32 operands; they are packed to bf16 here.
int launch_corr_argmax_bf16(const float* xn, const float* ref, int B, int C, int N, int32_t* ind, float* vmax,
                            void* ws, size_t ws_bytes, hipStream_t st, CorrPartials* partials, int ld)
{
    if (ld <= 0) ld = N;
    if (!corr_bf16_supported(C, ld))
        return fail(IPSR_ERR_UNSUPPORTED, "bf16 correlation: needs C %% %d == 0 and a row stride that is a multiple of %d (got C=%d, ld=%d)",
                    16 * PF, BM, C, ld);
    int qt, kt, ks, kpw;
    plan(B, N, &qt, &kt, &ks, &kpw);
    if (ld < qt * BN) return fail(IPSR_ERR_UNSUPPORTED, "bf16 correlation: row stride %d < %d", ld, qt * BN);
    if (ws_bytes < corr_argmax_bf16_ws_bytes(B, C, N, ld))
        return fail(IPSR_ERR_WORKSPACE, "ipsr_corr_argmax_bf16: workspace %zu < %zu", ws_bytes, corr_argmax_bf16_ws_bytes(B, C, N, ld));
    Carver cv(ws, ws_bytes);
    float* pval = cv.take<float>((size_t)B * ks * N);
    int32_t* pidx = cv.take<int32_t>((size_t)B * ks * N);
    const int C8 = C / 8;
    uint4* xp = cv.take<uint4>((size_t)B * C8 * ld);
    uint4* rp = cv.take<uint4>((size_t)B * C8 * ld);
    const dim3 pg(cdiv(ld, 256), C8, B);
    pack_bf16_k8_kernel<<<pg, 256, 0, st>>>(xn, C, ld, C8, xp);
    pack_bf16_k8_kernel<<<pg, 256, 0, st>>>(ref, C, ld, C8, rp);
    if (int rc = check_launch("pack_bf16_k8_kernel")) return rc;
    const int grid = B * qt * ks;
    profile_mark_start(st);
    if (ld == N) corr_argmax_bf16_kernel<false><<<grid, NTHREADS, 0, st>>>(xp, rp, C8, N, ld, qt, ks, kt, kpw, pval, pidx);
    else corr_argmax_bf16_kernel<true><<<grid, NTHREADS, 0, st>>>(xp, rp, C8, N, ld, qt, ks, kt, kpw, pval, pidx);
    profile_mark_stop(st);
    if (int rc = check_launch("corr_argmax_bf16_kernel")) return rc;
    if (partials) {
        partials->pval = pval;
        partials->pidx = pidx;
        partials->ksplit = ks;
        return IPSR_OK;
    }
    argmax_merge_kernel<<<cdiv(B * N, 256), 256, 0, st>>>(pval, pidx, B, N, ks, ind, vmax);
    return check_launch("argmax_merge_kernel");
}

}  // namespace ipsr

#ifdef IPSR_CLOCK_PROBE
extern "C" int ipsr_debug_read_probe(unsigned long long* host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ipsr::g_probe), sizeof(unsigned long long) * 2 * n);
}
#endif
