// attention.hip — K6 coherent-attention recurrence, K7 shift-fold reconstruction and the sparse form of
// trunc(kbar) that the backward needs.
//
// Reference: models/IPSRFunction.py:70-134.  For every sample the reference walks all N positions in
// raster order in Python; masked position l (q = mask_point_idx[l], kq = ind[q]) does
//      at = <P[q]/(||P[q]||+1e-8), o_{l-1}>;  s = at + vmax[q];  wn = at/s;  wo = vmax[q]/s
//      o_l = wn*o_{l-1} + wo*P[kq];          a_l = wn*a_{l-1};  a_l[kq] += wo           (:105-125)
// (o_0 = P[kq], a_0 = onehot(kq)), fills the dense N x N matrix kbar column by column and finally
// multiplies it with the raw patches (conv_transpose2d, :130-133).  Here:
//   * recurrence_kernel  — the only truly serial part: one 64-lane wave per sample keeps o_l in
//     registers (8 channels per lane per 512), reads patch rows from the patch-major copy xT through a
//     4-deep register prefetch ring, reduces the dot with DPP + readlane (no LDS, no barrier) and emits
//     only the scalars (wn_l, wo_l).  Latency-bound by construction: M dependent steps.
//   * attn_compress_kernel — a_l[k] is a scalar recurrence per k, independent across k, and non-zero only in
//     the columns {kq_l}: one thread per ACTIVE column replays (wn_l, wo_l) from LDS (see "Compressed attention").
//   * recon_gather_kernel — non-masked q: kbar column is one-hot, so out[:,q] = P[ind[q]] is a row
//     gather (LDS-transposed so that both the read of xT rows and the write of out rows are coalesced).
//   * recon_masked_kernel — masked q: out[:,q_l] = sum_k a_l[k] * P[k,:], the dense part of the
//     reference's second GEMM, on fp32 MFMA over the active columns in ascending k (one fmaf chain per
//     output; the skipped terms are exact zeros, so the bits equal the oracle's dense chain).
//   * attn_prepare / index_scan / csr_fill — kbar is kept by the reference in a LongTensor (:36,134), i.e.
//     truncated toward zero.  What survives is stored as a CSR over the patch index k: the non-masked q
//     with ind[q] == k (weight 1, ascending q) followed by the masked rows whose |a_l[k]| >= 1 (weight
//     trunc(a_l[k]), ascending l).  That is all the backward needs.
#include <cstdlib>

#include "ipsr_common.h"

namespace ipsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------
// wave-wide sum with the canonical tree: xor-butterfly 1,2,4,8 inside each row of 16 lanes (DPP), then
// (r0 + r1) + (r2 + r3) over the four row sums.  Every lane returns the same bits.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum_canonical(float p)
{
    p = p + dpp_mov<0xB1>(p);    // quad_perm [1,0,3,2]  : lane ^ 1
    p = p + dpp_mov<0x4E>(p);    // quad_perm [2,3,0,1]  : lane ^ 2
    p = p + dpp_mov<0x141>(p);   // row_half_mirror      : the other quad of the 8-group (all its lanes hold the same sum)
    p = p + dpp_mov<0x140>(p);   // row_mirror           : the other 8-group of the row
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p), 48));
    return (r0 + r1) + (r2 + r3);
}

// ---------------------------------------------------------------------------------------------------
// K6 serial part.  NCH = ceil(Cp / 512): lane j owns the 8-channel chunks j, j+64, ...
//
// Latency engineering (this kernel is M dependent steps on ONE wave per sample, nothing else matters):
//   * the index chain mpi -> ind -> (inv, vmax) is resolved for all steps up front into LDS, stored by step
//     so that the 4 steps of a ring turn are one aligned ds_read_b128 per array;
//   * patch rows come through a RING-deep register ring; the steady-state loop is branch-free, so the
//     compiler can wait with a counted vmcnt for exactly the rows of the current step while the rows of the
//     next RING-1 steps stay in flight (a conditional in the loop degrades that to vmcnt(0) = one full memory
//     round trip per step);
//   * (wn_l, wo_l) are parked in the lane l%64 and written 64 at a time, so the loop has no stores either.
constexpr int RING = 4;

template <int NCH>
struct RowRegs { float v[NCH][8]; };

// Branch-free row load: lanes whose chunk lies past Cp read chunk 0 instead (a valid address); the loaded
// registers are NOT touched here (any use would force the compiler to wait for the load right away) — the
// consumer zeroes such chunks with `live_mask` when it finally reads them.
template <int NCH>
__device__ __forceinline__ void load_row(RowRegs<NCH>& dst, const float* __restrict__ xTb, int row, int Cp, int lane)
{
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int base = (lane + 64 * i) * 8;
        const float* p = xTb + (size_t)row * Cp + (base < Cp ? base : 0);
        const float4 v0 = *reinterpret_cast<const float4*>(p);
        const float4 v1 = *reinterpret_cast<const float4*>(p + 4);
        dst.v[i][0] = v0.x; dst.v[i][1] = v0.y; dst.v[i][2] = v0.z; dst.v[i][3] = v0.w;
        dst.v[i][4] = v1.x; dst.v[i][5] = v1.y; dst.v[i][6] = v1.z; dst.v[i][7] = v1.w;
    }
}

// one step of IPSRFunction.py:105-125 on the wave-distributed state o.  FULL: Cp == 512*NCH, every lane chunk is
// live and no masking is needed; otherwise chunks past Cp contribute exact zeros.
template <int NCH, bool FULL>
__device__ __forceinline__ void rec_step(RowRegs<NCH>& o, const RowRegs<NCH>& pu, const RowRegs<NCH>& pk, float iq, float v,
                                         int Cp, int lane, float& wn, float& wo)
{
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const bool live = FULL || ((lane + 64 * i) * 8 < Cp);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float u = (live ? pu.v[i][e] : 0.0f) * iq;                                 // u = P[q]*inv  (:109)
            acc = __builtin_fmaf(u, o.v[i][e], acc);
        }
    }
    const float at = wave_sum_canonical(acc);                                                // (:116)
    const float s = at + v;
    wn = at / s;                                                                             // (:120)
    wo = v / s;                                                                              // (:121)
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const bool live = FULL || ((lane + 64 * i) * 8 < Cp);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float t0 = wn * o.v[i][e];
            const float t1 = wo * (live ? pk.v[i][e] : 0.0f);
            o.v[i][e] = t0 + t1;                                                             // (:122)
        }
    }
}

template <int NCH, bool FULL>
__global__ void __launch_bounds__(64) recurrence_kernel(const float* __restrict__ xT, const float* __restrict__ inv,
                                                        const int32_t* __restrict__ ind, const float* __restrict__ vmax,
                                                        const int32_t* __restrict__ mpi, int Cp, int N, int M,
                                                        float* __restrict__ wn_out, float* __restrict__ wo_out,
                                                        int32_t* __restrict__ kq_out)
{
    // step-indexed LDS arrays: entry s describes step l = s + 1 (so a ring turn l = 1+4j.. is 16-byte aligned);
    // padded by 3*RING entries that repeat a valid row index so that run-ahead prefetches stay in bounds
    extern __shared__ __attribute__((aligned(16))) int lds_raw[];
    const int Mp = ((M + 3) & ~3) + 3 * RING;
    int* q_s = lds_raw;
    int* kq_s = q_s + Mp;
    float* iv_s = reinterpret_cast<float*>(kq_s + Mp);
    float* vm_s = iv_s + Mp;
    float* wn_s = vm_s + Mp;      // (wn_l, wo_l) indexed by l, copied out in one coalesced pass at the end
    float* wo_s = wn_s + Mp;

    const int b = blockIdx.x, lane = threadIdx.x;
    const float* xTb = xT + (size_t)b * N * Cp;
    const int32_t* indb = ind + (size_t)b * N;

    for (int l = lane; l < M; l += 64) {
        const int q = mpi[l];
        const int kq = indb[q];
        kq_out[(size_t)b * M + l] = kq;
        if (l >= 1) {
            q_s[l - 1] = q;
            kq_s[l - 1] = kq;
            iv_s[l - 1] = inv[(size_t)b * N + q];
            vm_s[l - 1] = vmax[(size_t)b * N + q];
        }
    }
    const int q0 = mpi[0];
    const int kq0 = indb[q0];
    for (int s = M - 1 + lane; s < Mp; s += 64) { q_s[s] = q0; kq_s[s] = kq0; iv_s[s] = 0.0f; vm_s[s] = 1.0f; }
    if (lane == 0) { wn_s[0] = 0.0f; wo_s[0] = 1.0f; }      // step 0: (wn, wo) = (0, 1) makes a_0 = onehot(kq_0)
    __syncthreads();

    // Two register sets of RING slots: set A serves even ring turns, set B odd ones.  A slot is refilled right
    // after it is consumed with the row of the step TWO turns ahead, so every load has a full turn (4 steps)
    // of compute to land in, wherever the scheduler places it inside the turn.
    RowRegs<NCH> o, puA[RING], pkA[RING], puB[RING], pkB[RING];
    load_row<NCH>(o, xTb, kq0, Cp, lane);              // step 0: o_0 = P[kq_0]   (IPSRFunction.py:98-101)
    if (!FULL) {
#pragma unroll
        for (int i = 0; i < NCH; ++i)
            if ((lane + 64 * i) * 8 >= Cp) {
#pragma unroll
                for (int e = 0; e < 8; ++e) o.v[i][e] = 0.0f;
            }
    }
#pragma unroll
    for (int d = 0; d < RING; ++d) {
        load_row<NCH>(puA[d], xTb, q_s[d], Cp, lane);        load_row<NCH>(pkA[d], xTb, kq_s[d], Cp, lane);
        load_row<NCH>(puB[d], xTb, q_s[RING + d], Cp, lane); load_row<NCH>(pkB[d], xTb, kq_s[RING + d], Cp, lane);
    }

    const int nsteps = M - 1;                          // steps l = 1 .. M-1  <->  s = 0 .. nsteps-1
    const int nfull = nsteps / RING;                   // branch-free ring turns

#define IPSR_TURN(PU, PK, S0)                                                                                   \
    do {                                                                                                        \
        const int s0_ = (S0);                                                                                   \
        const int4 qn = *reinterpret_cast<const int4*>(&q_s[s0_ + 2 * RING]);                                   \
        const int4 kn = *reinterpret_cast<const int4*>(&kq_s[s0_ + 2 * RING]);                                  \
        const float4 iv4 = *reinterpret_cast<const float4*>(&iv_s[s0_]);                                        \
        const float4 vm4 = *reinterpret_cast<const float4*>(&vm_s[s0_]);                                        \
        const int qn_[4] = {qn.x, qn.y, qn.z, qn.w}, kn_[4] = {kn.x, kn.y, kn.z, kn.w};                         \
        const float iv_[4] = {iv4.x, iv4.y, iv4.z, iv4.w}, vm_[4] = {vm4.x, vm4.y, vm4.z, vm4.w};               \
        _Pragma("unroll") for (int d = 0; d < RING; ++d) {                                                      \
            float wn, wo;                                                                                       \
            rec_step<NCH, FULL>(o, PU[d], PK[d], iv_[d], vm_[d], Cp, lane, wn, wo);                                             \
            wn_s[s0_ + d + 1] = wn;                                                                             \
            wo_s[s0_ + d + 1] = wo;                                                                             \
            load_row<NCH>(PU[d], xTb, qn_[d], Cp, lane);                                                        \
            load_row<NCH>(PK[d], xTb, kn_[d], Cp, lane);                                                        \
        }                                                                                                       \
    } while (0)
#define IPSR_TAIL(PU, PK, S0)                                                                                   \
    do {                                                                                                        \
        const int s0_ = (S0);                                                                                   \
        _Pragma("unroll") for (int d = 0; d < RING - 1; ++d) {                                                  \
            if (s0_ + d < nsteps) {                                                                             \
                float wn, wo;                                                                                   \
                rec_step<NCH, FULL>(o, PU[d], PK[d], iv_s[s0_ + d], vm_s[s0_ + d], Cp, lane, wn, wo);                           \
                wn_s[s0_ + d + 1] = wn;                                                                         \
                wo_s[s0_ + d + 1] = wo;                                                                         \
            }                                                                                                   \
        }                                                                                                       \
    } while (0)

    int t = 0;
    for (; t + 2 <= nfull; t += 2) {
        IPSR_TURN(puA, pkA, t * RING);
        IPSR_TURN(puB, pkB, (t + 1) * RING);
    }
    if (t < nfull) {                                   // odd number of full turns: one more on A, partial turn on B
        IPSR_TURN(puA, pkA, t * RING);
        IPSR_TAIL(puB, pkB, (t + 1) * RING);
    } else {
        IPSR_TAIL(puA, pkA, t * RING);
    }
#undef IPSR_TURN
#undef IPSR_TAIL
    __syncthreads();
    for (int l = lane; l < M; l += 64) { wn_out[(size_t)b * M + l] = wn_s[l]; wo_out[(size_t)b * M + l] = wo_s[l]; }
}

// ---------------------------------------------------------------------------------------------------
// non-masked columns: out[c][q] = P[ind[q]][c]  (one-hot kbar column, IPSRFunction.py:129-133).
// (masked columns are written too and overwritten by recon_masked_kernel afterwards.)
__global__ void __launch_bounds__(256) recon_gather_kernel(const float* __restrict__ xT, const int32_t* __restrict__ ind,
                                                           int C, int Cp, int N, float* __restrict__ out)
{
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int q0 = blockIdx.x * 32, c0 = blockIdx.y * 32, b = blockIdx.z;
    const float* xTb = xT + (size_t)b * N * Cp;
    // two batches of independent loads (the 4 row indices, then the 4 row elements) instead of 4 dependent pairs
    int row[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = q0 + ty + 8 * i;
        row[i] = q < N ? ind[(size_t)b * N + q] : 0;
    }
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = q0 + ty + 8 * i;
        v[i] = (q < N && c0 + tx < Cp) ? xTb[(size_t)row[i] * Cp + c0 + tx] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[ty + 8 * i][tx] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cl = ty + 8 * i, c = c0 + cl;
        if (c < C && q0 + tx < N) out[((size_t)b * C + c) * N + q0 + tx] = tile[tx][cl];
    }
}

// ---------------------------------------------------------------------------------------------------
// Compressed attention.
//
// Row l of the reference's `in_attention` [M,N] (IPSRFunction.py:76,123-125) is non-zero only in the columns
// kq_0..kq_l it has touched, so all M rows live in the M' <= M "active" columns D = sorted{kq_l}.  Everything
// downstream works on the compressed matrix Ac[l][j] = a_l[D_j] (zero terms of the dense sums are exact no-ops):
//   attn_prepare_kernel     per sample: active-column list D (ascending k), rank of every column, jq_l = rank(kq_l),
//                           and the one-hot column counts of trunc(kbar) (non-masked q with ind[q] == k)
//   attn_compress_kernel    thread per active column j replays the scalar recurrence  a = a*wn_l (+ wo_l if jq_l == j)
//                           from LDS, writes Ac coalesced over j and counts the entries with |a| >= 1 that survive the
//                           reference's LongTensor truncation (:36,134)
//   index_scan_kernel       exclusive scan of the per-column entry counts -> col_off
//   csr_fill_kernel         block 0: one wave ranks the non-masked q by (ind[q], q) with ballot-based matching and
//                           writes the one-hot entries in ascending q; other blocks: the survivors, ascending l
//   attn_expand_kernel      only when the caller asks for the dense [M,N] rows (tests / inspection)
//   recon_masked_kernel     out[:,q_l] = sum_j Ac[l][j] * P[D_j,:]  on fp32 MFMA, j ascending == k ascending

__global__ void __launch_bounds__(1024) attn_prepare_kernel(const int32_t* __restrict__ ind, const int32_t* __restrict__ mpi,
                                                            const int32_t* __restrict__ kq, int N, int M, int Mc,
                                                            int32_t* __restrict__ dlist, int32_t* __restrict__ mprime,
                                                            int32_t* __restrict__ jq, int32_t* __restrict__ rankflag,
                                                            int32_t* __restrict__ onehot_cnt, int32_t* __restrict__ col_cnt)
{
    extern __shared__ __attribute__((aligned(16))) int lds[];
    int* flag = lds;            // [N] 1 = active column
    int* cnt = lds + N;         // [N] one-hot count
    int* ismask = lds + 2 * N;  // [N]
    __shared__ int wave_tot[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, b = blockIdx.x;
    const int32_t* indb = ind + (size_t)b * N;
    const int32_t* kqb = kq + (size_t)b * M;
    for (int k = tid; k < N; k += 1024) { flag[k] = 0; cnt[k] = 0; ismask[k] = 0; }
    __syncthreads();
    for (int l = tid; l < M; l += 1024) { flag[kqb[l]] = 1; ismask[mpi[l]] = 1; }
    __syncthreads();
    for (int q = tid; q < N; q += 1024)
        if (!ismask[q]) atomicAdd(&cnt[indb[q]], 1);
    // ordered compaction of the active columns: thread t owns k in [t*KPT, (t+1)*KPT)
    const int KPT = (N + 1023) / 1024;
    const int k_lo = tid * KPT, k_hi = min(N, k_lo + KPT);
    int total = 0;
    for (int k = k_lo; k < k_hi; ++k) total += flag[k];
    int incl = total;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const int t = __shfl_up(incl, s);
        if (lane >= s) incl += t;
    }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    int rank = incl - total;
    for (int j = 0; j < wv; ++j) rank += wave_tot[j];
    for (int k = k_lo; k < k_hi; ++k) {
        const int f = flag[k];
        rankflag[(size_t)b * N + k] = f ? rank : -1;
        if (f) { dlist[(size_t)b * Mc + rank] = k; flag[k] = rank + 1; }     // flag now holds rank+1 for the jq lookup
        rank += f;
        const int c = cnt[k];
        onehot_cnt[(size_t)b * N + k] = c;
        col_cnt[(size_t)b * N + k] = c;
    }
    if (tid == 1023) mprime[b] = rank;
    __syncthreads();
    int mp = 0;
    for (int j = 0; j < 16; ++j) mp += wave_tot[j];
    for (int j = mp + tid; j < Mc; j += 1024) dlist[(size_t)b * Mc + j] = 0;     // padding rows of the GEMM: any valid patch
    for (int l = tid; l < M; l += 1024) jq[(size_t)b * M + l] = flag[kqb[l]] - 1;
}

constexpr int AC_COLS = 128;

// (wn, wo, jq) of all steps into LDS as float4, 2 independent loads per thread per batch
__device__ __forceinline__ float4* load_steps_lds(int* lds, const float* __restrict__ wn, const float* __restrict__ wo,
                                                  const int32_t* __restrict__ jq, int M)
{
    float4* step = reinterpret_cast<float4*>(lds);
    for (int l0 = 0; l0 < M; l0 += 2 * AC_COLS) {
        float4 v[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int l = l0 + j * AC_COLS + threadIdx.x;
            v[j] = l < M ? make_float4(wn[l], wo[l], __int_as_float(jq[l]), 0.0f) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) { const int l = l0 + j * AC_COLS + threadIdx.x; if (l < M) step[l] = v[j]; }
    }
    __syncthreads();
    return step;
}

__global__ void __launch_bounds__(AC_COLS) attn_compress_kernel(const float* __restrict__ wn, const float* __restrict__ wo,
                                                                const int32_t* __restrict__ jq, const int32_t* __restrict__ dlist,
                                                                const int32_t* __restrict__ mprime, int N, int M, int Mc,
                                                                float* __restrict__ ac, int32_t* __restrict__ surv_cnt,
                                                                int32_t* __restrict__ col_cnt)
{
    extern __shared__ __attribute__((aligned(16))) int lds[];
    const int b = blockIdx.y;
    const float4* step = load_steps_lds(lds, wn + (size_t)b * M, wo + (size_t)b * M, jq + (size_t)b * M, M);
    const int j = blockIdx.x * AC_COLS + threadIdx.x;
    if (j >= Mc) return;
    float* acb = ac + (size_t)b * M * Mc;
    float a = 0.0f;
    int cnt = 0;
#pragma unroll 8
    for (int l = 0; l < M; ++l) {
        const float4 s = step[l];
        a = a * s.x;                                            // (:123)
        a = (__float_as_int(s.z) == j) ? a + s.y : a;           // (:124)
        acb[(size_t)l * Mc + j] = a;                            // (:125) compressed row l
        cnt += (truncf(a) != 0.0f) ? 1 : 0;
    }
    if (j < mprime[b]) {
        surv_cnt[(size_t)b * Mc + j] = cnt;
        if (cnt) col_cnt[(size_t)b * N + dlist[(size_t)b * Mc + j]] += cnt;     // single writer per column
    }
}

// exclusive scan of col_cnt over k -> col_off[0..N]; one workgroup per sample
__global__ void __launch_bounds__(1024) index_scan_kernel(const int32_t* __restrict__ col_cnt, int N,
                                                          int32_t* __restrict__ bwd_index, size_t ints_per_sample)
{
    __shared__ int wave_tot[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, b = blockIdx.x;
    int32_t* col_off = bwd_index + (size_t)b * ints_per_sample;
    const int KPT = (N + 1023) / 1024;
    const int k_lo = tid * KPT, k_hi = min(N, k_lo + KPT);
    int total = 0;
    for (int k = k_lo; k < k_hi; ++k) total += col_cnt[(size_t)b * N + k];
    int incl = total;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const int t = __shfl_up(incl, s);
        if (lane >= s) incl += t;
    }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    int off = incl - total;
    for (int j = 0; j < wv; ++j) off += wave_tot[j];
    for (int k = k_lo; k < k_hi; ++k) {
        col_off[k] = off;
        off += col_cnt[(size_t)b * N + k];
    }
    if (tid == 1023) col_off[N] = off;
}

__global__ void __launch_bounds__(AC_COLS) csr_fill_kernel(const int32_t* __restrict__ ind, const int32_t* __restrict__ mpi,
                                                           const float* __restrict__ wn, const float* __restrict__ wo,
                                                           const int32_t* __restrict__ jq, const int32_t* __restrict__ dlist,
                                                           const int32_t* __restrict__ mprime, const int32_t* __restrict__ onehot_cnt,
                                                           const int32_t* __restrict__ surv_cnt, int N, int M, int Mc, int nbits,
                                                           int32_t* __restrict__ bwd_index, size_t ints_per_sample, size_t cap)
{
    extern __shared__ __attribute__((aligned(16))) int lds[];
    const int b = blockIdx.y;
    int32_t* col_off = bwd_index + (size_t)b * ints_per_sample;
    int32_t* ent_q = col_off + N + 1;
    float* ent_w = reinterpret_cast<float*>(ent_q + cap);

    if (blockIdx.x == 0) {
        // ---- one-hot rows: non-masked q, grouped by k = ind[q], ascending q inside a group.  One wave walks the
        // positions 64 at a time; lanes with equal keys find each other with one ballot per key bit.  Keys and the
        // per-column write cursors (initialised to col_off) live in LDS, so the serial loop touches no global memory
        // except the entry stores.
        int* cursor = lds;           // [N] next entry slot of column k
        int* key = lds + N;          // [N] ind[q], or -1 for masked q
        const int32_t* indb = ind + (size_t)b * N;
        for (int k0 = 0; k0 < N; k0 += 4 * AC_COLS) {
            int c[4], v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + j * AC_COLS + threadIdx.x;
                c[j] = k < N ? col_off[k] : 0;
                v[j] = k < N ? indb[k] : -1;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + j * AC_COLS + threadIdx.x;
                if (k < N) { cursor[k] = c[j]; key[k] = v[j]; }
            }
        }
        __syncthreads();
        for (int l = threadIdx.x; l < M; l += AC_COLS) key[mpi[l]] = -1;
        __syncthreads();
        if (threadIdx.x >= 64) return;
        const int lane = threadIdx.x;
        const unsigned long long lt = (1ull << lane) - 1ull;
        for (int q0 = 0; q0 < N; q0 += 64) {
            const int q = q0 + lane;
            const int kv = q < N ? key[q] : -1;
            const bool valid = kv >= 0;
            unsigned long long m = __ballot(valid);
            for (int bit = 0; bit < nbits; ++bit) {
                const bool one = (kv >> bit) & 1;
                const unsigned long long bal = __ballot(valid && one);
                m &= one ? bal : ~bal;
            }
            if (valid) {
                const int base = cursor[kv];
                const int rank = __popcll(m & lt);
                ent_q[base + rank] = q;
                ent_w[base + rank] = 1.0f;
                if (rank == 0) cursor[kv] = base + __popcll(m);      // one leader per key; the reads above precede this write
            }
        }
        return;
    }
    // ---- masked rows that survive the truncation: thread per active column replays the recurrence
    const float4* step = load_steps_lds(lds, wn + (size_t)b * M, wo + (size_t)b * M, jq + (size_t)b * M, M);
    const int j = (blockIdx.x - 1) * AC_COLS + threadIdx.x;
    if (j >= mprime[b] || surv_cnt[(size_t)b * Mc + j] == 0) return;
    const int k = dlist[(size_t)b * Mc + j];
    int e = col_off[k] + onehot_cnt[(size_t)b * N + k];
    float a = 0.0f;
#pragma unroll 8
    for (int l = 0; l < M; ++l) {
        const float4 s = step[l];
        a = a * s.x;
        a = (__float_as_int(s.z) == j) ? a + s.y : a;
        const float t = truncf(a);
        if (t != 0.0f) { ent_q[e] = mpi[l]; ent_w[e] = t; ++e; }
    }
}

// dense rows of `in_attention` (optional output): attn[l][k] = active(k) ? Ac[l][rank(k)] : 0
__global__ void __launch_bounds__(256) attn_expand_kernel(const float* __restrict__ ac, const int32_t* __restrict__ rankflag,
                                                          int N, int M, int Mc, float* __restrict__ attn)
{
    const int b = blockIdx.z, l = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= N) return;
    const int r = rankflag[(size_t)b * N + k];
    attn[((size_t)b * M + l) * N + k] = r >= 0 ? ac[((size_t)b * M + l) * Mc + r] : 0.0f;
}

// ---------------------------------------------------------------------------------------------------
// masked columns on the matrix cores:  Dm[c][l] = sum_j xT[D_j][c] * Ac[l][j],  out[c][mpi[l]] = Dm[c][l].
// A[i=c][kk=j] = xT[D_j][c]: patch-major rows = MFMA operand order (plain LDS copy, rows picked through D);
// B[kk=j][jn=l] = Ac[l][j] is j-contiguous, so its LDS image is [l][j] with a padded row (33) for conflict-free
// column reads.  The K loop runs over the M' active columns only (vs N for the reference's dense GEMM).
constexpr int RM_BC = 64, RM_BL = 64, RM_BK = 32;

__global__ void __launch_bounds__(256) recon_masked_kernel(const float* __restrict__ xT, const float* __restrict__ ac,
                                                           const int32_t* __restrict__ dlist, const int32_t* __restrict__ mprime,
                                                           const int32_t* __restrict__ mpi, int C, int Cp, int N, int M, int Mc,
                                                           float* __restrict__ out)
{
    __shared__ __attribute__((aligned(16))) float As[2][RM_BK][RM_BC];
    __shared__ float Bs[2][RM_BL][RM_BK + 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    const int l0 = blockIdx.x * RM_BL, c0 = blockIdx.y * RM_BC, b = blockIdx.z;
    const float* xTb = xT + (size_t)b * N * Cp;
    const float* acb = ac + (size_t)b * M * Mc;
    const int32_t* db = dlist + (size_t)b * Mc;
    const int mp = mprime[b];

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;

    float4 ra[2], rb[2];
    auto gload = [&](int s) {
        const int j0 = s * RM_BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
            {   // A: 32 rows (j) x 16 float4 (c); dlist is padded to Mc with valid rows, Ac is zero there
                const int kk = idx >> 4, c4 = (idx & 15) * 4;
                const int j = j0 + kk, c = c0 + c4;
                if (j < Mc && c + 4 <= Cp) ra[i] = *reinterpret_cast<const float4*>(xTb + (size_t)db[j] * Cp + c);
                else ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            {   // B: 64 rows (l) x 8 float4 (j)   (Mc % 4 == 0)
                const int jl = idx >> 3, k4 = (idx & 7) * 4;
                const int l = l0 + jl, j = j0 + k4;
                if (l < M && j + 4 <= Mc) rb[i] = *reinterpret_cast<const float4*>(acb + (size_t)l * Mc + j);
                else rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
            *reinterpret_cast<float4*>(&As[buf][idx >> 4][(idx & 15) * 4]) = ra[i];
            const int jl = idx >> 3, k4 = (idx & 7) * 4;
            Bs[buf][jl][k4 + 0] = rb[i].x; Bs[buf][jl][k4 + 1] = rb[i].y;
            Bs[buf][jl][k4 + 2] = rb[i].z; Bs[buf][jl][k4 + 3] = rb[i].w;
        }
    };

    const int nstage = (mp + RM_BK - 1) / RM_BK;
    if (nstage > 0) {
        gload(0);
        sstore(0);
    }
    __syncthreads();
    for (int s = 0; s < nstage; ++s) {
        const int cur = s & 1;
        if (s + 1 < nstage) gload(s + 1);
#pragma unroll
        for (int kk = 0; kk < RM_BK / 2; ++kk) {
            const float a = As[cur][kk * 2 + h][wm * 32 + r];
            const float bb = Bs[cur][wn * 32 + r][kk * 2 + h];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc, 0, 0, 0);
        }
        if (s + 1 < nstage) sstore(cur ^ 1);
        __syncthreads();
    }

    const int l = l0 + wn * 32 + r;
    if (l < M) {
        const int q = mpi[l];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = c0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (c < C) out[((size_t)b * C + c) * N + q] = acc[e];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
int launch_attention(const AttnArgs& a, hipStream_t st)
{
    const int B = a.B, C = a.C, Cp = a.Cp, N = a.N, M = a.M, Mc = a.Mc;
    const size_t cap = (size_t)(N - M) + (size_t)M * (M + 1) / 2;
    const size_t ints = (size_t)N + 1 + 2 * cap;
    if (M > 0) {
        const int nch = cdiv(Cp, 512);
        const size_t lds = (size_t)6 * (((M + 3) & ~3) + 3 * RING) * sizeof(int);
        if (lds > 160 * 1024 - 1024) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_forward: M=%d too large for the recurrence's LDS index cache", M);
#define LAUNCH_REC2(NCH, FULL)                                                                                       \
    do {                                                                                                             \
        if (lds > 48 * 1024)                                                                                         \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&recurrence_kernel<NCH, FULL>),                  \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
        recurrence_kernel<NCH, FULL><<<B, 64, lds, st>>>(a.xT, a.inv, a.ind, a.vmax, a.mpi, Cp, N, M, a.wn, a.wo, a.kq); \
    } while (0)
#define LAUNCH_REC(NCH)                                                                                              \
    do {                                                                                                             \
        if (Cp == 512 * (NCH)) LAUNCH_REC2(NCH, true);                                                               \
        else LAUNCH_REC2(NCH, false);                                                                                \
    } while (0)
        switch (nch) {
            case 1: LAUNCH_REC(1); break;
            case 2: LAUNCH_REC(2); break;
            case 3: LAUNCH_REC(3); break;
            case 4: LAUNCH_REC(4); break;
            default: return fail(IPSR_ERR_UNSUPPORTED, "ipsr_forward: C=%d > 2048 channels not supported", C);
        }
#undef LAUNCH_REC
#undef LAUNCH_REC2
        if (int rc = check_launch("recurrence_kernel")) return rc;
    }
    // non-masked columns first (masked ones are overwritten below)
    recon_gather_kernel<<<dim3(cdiv(N, 32), cdiv(C, 32), B), 256, 0, st>>>(a.xT, a.ind, C, Cp, N, a.out);
    if (int rc = check_launch("recon_gather_kernel")) return rc;

    const bool need_index = a.bwd_index != nullptr;
    if (M > 0 || need_index) {
        const size_t lds_prep = (size_t)3 * N * sizeof(int);
        if (lds_prep > 150 * 1024) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_forward: N=%d too large for attn_prepare_kernel", N);
        if (lds_prep > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_prepare_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_prep);
        attn_prepare_kernel<<<B, 1024, lds_prep, st>>>(a.ind, a.mpi, a.kq, N, M, Mc, a.dlist, a.mprime, a.jq, a.rankflag, a.onehot_cnt, a.col_cnt);
        if (int rc = check_launch("attn_prepare_kernel")) return rc;
    }
    const size_t lds_steps = (size_t)4 * (M > 0 ? M : 1) * sizeof(int);
    if (M > 0) {
        if (lds_steps > 150 * 1024) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_forward: M=%d too large for attn_compress_kernel", M);
        if (lds_steps > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_compress_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_steps);
        attn_compress_kernel<<<dim3(cdiv(Mc, AC_COLS), B), AC_COLS, lds_steps, st>>>(a.wn, a.wo, a.jq, a.dlist, a.mprime, N, M, Mc, a.ac, a.surv_cnt, a.col_cnt);
        if (int rc = check_launch("attn_compress_kernel")) return rc;
        recon_masked_kernel<<<dim3(cdiv(M, RM_BL), cdiv(C, RM_BC), B), 256, 0, st>>>(a.xT, a.ac, a.dlist, a.mprime, a.mpi, C, Cp, N, M, Mc, a.out);
        if (int rc = check_launch("recon_masked_kernel")) return rc;
        if (a.attn) {
            attn_expand_kernel<<<dim3(cdiv(N, 256), M, B), 256, 0, st>>>(a.ac, a.rankflag, N, M, Mc, a.attn);
            if (int rc = check_launch("attn_expand_kernel")) return rc;
        }
    }
    if (need_index) {
        index_scan_kernel<<<B, 1024, 0, st>>>(a.col_cnt, N, a.bwd_index, ints);
        if (int rc = check_launch("index_scan_kernel")) return rc;
        const size_t lds_fill = (size_t)(2 * N > 4 * M ? 2 * N : 4 * M) * sizeof(int);
        if (lds_fill > 150 * 1024) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_forward: N=%d too large for csr_fill_kernel", N);
        if (lds_fill > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&csr_fill_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fill);
        int nbits = 1;
        while ((1 << nbits) < N) ++nbits;
        static const int dbg = getenv("IPSR_DEBUG_CSR") ? atoi(getenv("IPSR_DEBUG_CSR")) : 0;
        csr_fill_kernel<<<dim3(dbg == 1 ? 1 : 1 + (M > 0 ? cdiv(Mc, AC_COLS) : 0), B), AC_COLS, lds_fill, st>>>(
            a.ind, a.mpi, a.wn, a.wo, a.jq, a.dlist, a.mprime, a.onehot_cnt, a.surv_cnt, N, M, Mc, nbits, a.bwd_index, ints, cap);
        if (int rc = check_launch("csr_fill_kernel")) return rc;
    }
    return IPSR_OK;
}

}  // namespace ipsr
