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
//   * gather_body — non-masked q: kbar column is one-hot, so out[:,q] = P[ind[q]] is a row
//     gather (LDS-transposed so that both the read of xT rows and the write of out rows are coalesced).
//   * recon_masked_kernel — masked q: out[:,q_l] = sum_k a_l[k] * P[k,:], the dense part of the
//     reference's second GEMM, on fp32 MFMA over the active columns in ascending k (one fmaf chain per
//     output; the skipped terms are exact zeros, so the bits equal the oracle's dense chain).
//   * prepare_body / attn_compress_kernel — kbar is kept by the reference in a LongTensor (:36,134), i.e.
//     truncated toward zero.  What survives is stored as a CSR over the patch index k: the non-masked q
//     with ind[q] == k (weight 1, ascending q) followed by the masked rows whose |a_l[k]| >= 1 (weight
//     trunc(a_l[k]), ascending l).  That is all the backward needs.
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

// Step-indexed LDS arrays shared by both recurrence bodies: entry s describes step l = s + 1 (so a ring turn
// l = 1+4j.. is 16-byte aligned); padded by 3*RING entries that repeat a valid row index so that run-ahead prefetches
// stay in bounds.
struct RecLds {
    int Mp;
    int* q_s;
    int* kq_s;
    float* iv_s;
    float* vm_s;
    float* wn_s;      // (wn_l, wo_l) indexed by l, copied out in one coalesced pass at the end
    float* wo_s;
    int q0, kq0;      // step 0
};

// All four waves resolve the chain mask_point_idx -> arg-max (merged from the k-split partials) -> (inv, vmax) for all steps.
__device__ __forceinline__ RecLds recurrence_prologue(int* lds_raw, int b, const float* __restrict__ inv, const CorrPartials& part,
                                                      const int32_t* __restrict__ mpi, int N, int M)
{
    RecLds r;
    const int Mp = ((M + 3) & ~3) + 3 * RING;
    r.Mp = Mp;
    r.q_s = lds_raw;
    r.kq_s = r.q_s + Mp;
    r.iv_s = reinterpret_cast<float*>(r.kq_s + Mp);
    r.vm_s = r.iv_s + Mp;
    r.wn_s = r.vm_s + Mp;
    r.wo_s = r.wn_s + Mp;
    const int tid = threadIdx.x;
    for (int l = tid; l < M; l += 256) {
        const int q = mpi_at(mpi, l, N);
        float vq; int kq;
        merged_argmax(part, b, N, q, vq, kq);
        if (l >= 1) {
            r.q_s[l - 1] = q;
            r.kq_s[l - 1] = kq;
            r.iv_s[l - 1] = inv[(size_t)b * N + q];
            r.vm_s[l - 1] = vq;
        } else {
            r.q_s[Mp - 1] = q;        // step 0's (q, kq) parked in the last padding slot (rewritten below with the same values)
            r.kq_s[Mp - 1] = kq;
        }
    }
    __syncthreads();
    r.q0 = r.q_s[Mp - 1];
    r.kq0 = r.kq_s[Mp - 1];
    __syncthreads();
    for (int s = M - 1 + tid; s < Mp; s += 256) { r.q_s[s] = r.q0; r.kq_s[s] = r.kq0; r.iv_s[s] = 0.0f; r.vm_s[s] = 1.0f; }
    if (tid == 0) { r.wn_s[0] = 0.0f; r.wo_s[0] = 1.0f; }      // step 0: (wn, wo) = (0, 1) makes a_0 = onehot(kq_0)
    __syncthreads();
    return r;
}

// Runs in a 256-thread block of the stage kernel: all four waves fill / drain the LDS arrays, wave 0 alone walks the chain.
template <int NCH, bool FULL>
__device__ __forceinline__ void recurrence_body(int* lds_raw, int b, const float* __restrict__ xT, const float* __restrict__ inv,
                                                const CorrPartials& part, const int32_t* __restrict__ mpi, int Cp, int N, int M, int Ms,
                                                float* __restrict__ wn_out, float* __restrict__ wo_out)
{
    const RecLds rl = recurrence_prologue(lds_raw, b, inv, part, mpi, N, M);
    int* const q_s = rl.q_s; int* const kq_s = rl.kq_s;
    float* const iv_s = rl.iv_s; float* const vm_s = rl.vm_s; float* const wn_s = rl.wn_s; float* const wo_s = rl.wo_s;
    const int kq0 = rl.kq0;
    const int tid = threadIdx.x, lane = tid & 63;
    const float* xTb = xT + (size_t)b * N * Cp;

  if (tid < 64) {
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
  }
    __syncthreads();
    for (int l = tid; l < M; l += 256) { wn_out[(size_t)b * Ms + l] = wn_s[l]; wo_out[(size_t)b * Ms + l] = wo_s[l]; }
}

// Wide patches (Cp > 1536; shift_sz > 1 makes a patch C*p*p numbers): the RING-deep double register ring above
// would need 17 rows of NCH*8 registers.  Same arithmetic, same lane ownership, but ONE row pair in flight per set
// (A serves even steps, B odd ones): 5 rows of NCH*8 registers, each load has one full step of compute to land in.
template <int NCH>
__device__ __forceinline__ void recurrence_wide_body(int* lds_raw, int b, const float* __restrict__ xT, const float* __restrict__ inv,
                                                     const CorrPartials& part, const int32_t* __restrict__ mpi, int Cp, int N, int M, int Ms,
                                                     float* __restrict__ wn_out, float* __restrict__ wo_out)
{
    const RecLds rl = recurrence_prologue(lds_raw, b, inv, part, mpi, N, M);
    const int tid = threadIdx.x, lane = tid & 63;
    const float* xTb = xT + (size_t)b * N * Cp;
    if (tid < 64) {
        RowRegs<NCH> o, puA, pkA, puB, pkB;
        load_row<NCH>(o, xTb, rl.kq0, Cp, lane);
#pragma unroll
        for (int i = 0; i < NCH; ++i)
            if ((lane + 64 * i) * 8 >= Cp) {
#pragma unroll
                for (int e = 0; e < 8; ++e) o.v[i][e] = 0.0f;
            }
        load_row<NCH>(puA, xTb, rl.q_s[0], Cp, lane); load_row<NCH>(pkA, xTb, rl.kq_s[0], Cp, lane);
        load_row<NCH>(puB, xTb, rl.q_s[1], Cp, lane); load_row<NCH>(pkB, xTb, rl.kq_s[1], Cp, lane);
        const int nsteps = M - 1;
        int s = 0;
        for (; s + 2 <= nsteps; s += 2) {                 // entries up to s+3 <= Mp-1 exist (padding repeats a valid row)
            float wn, wo;
            rec_step<NCH, false>(o, puA, pkA, rl.iv_s[s], rl.vm_s[s], Cp, lane, wn, wo);
            rl.wn_s[s + 1] = wn; rl.wo_s[s + 1] = wo;
            load_row<NCH>(puA, xTb, rl.q_s[s + 2], Cp, lane); load_row<NCH>(pkA, xTb, rl.kq_s[s + 2], Cp, lane);
            rec_step<NCH, false>(o, puB, pkB, rl.iv_s[s + 1], rl.vm_s[s + 1], Cp, lane, wn, wo);
            rl.wn_s[s + 2] = wn; rl.wo_s[s + 2] = wo;
            load_row<NCH>(puB, xTb, rl.q_s[s + 3], Cp, lane); load_row<NCH>(pkB, xTb, rl.kq_s[s + 3], Cp, lane);
        }
        if (s < nsteps) {
            float wn, wo;
            rec_step<NCH, false>(o, puA, pkA, rl.iv_s[s], rl.vm_s[s], Cp, lane, wn, wo);
            rl.wn_s[s + 1] = wn; rl.wo_s[s + 1] = wo;
        }
    }
    __syncthreads();
    for (int l = tid; l < M; l += 256) { wn_out[(size_t)b * Ms + l] = rl.wn_s[l]; wo_out[(size_t)b * Ms + l] = rl.wo_s[l]; }
}

// Patches wider than 2048 numbers (shift_sz > 1: C*p*p, 4608 for the reference's 512 channels and 3x3 patches): one
// wave would hold 72+ registers per row and issue 144 row loads per step (4.4 us per step measured).  All FOUR waves of
// the block walk the chain together instead: lane J of 256 owns the 8-number chunks J, J+256, ..; every wave reduces its
// 64 partials with the canonical butterfly, the four wave sums meet in LDS (one raw s_barrier per step, double-buffered
// slot, no vmcnt drain) and are added as (s0+s1)+(s2+s3) — the oracle's lane_dot for C > 2048.  Every wave then
// computes the same (wn, wo) bits and updates its own slice of o.  Row prefetch: QD single-row sets, distance QD steps.
constexpr int QD = 4;

template <int NW>
__device__ __forceinline__ void load_row_quad(RowRegs<NW>& dst, const float* __restrict__ xTb, int row, int Cp, int J)
{
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int base = (J + 256 * i) * 8;
        const float* p = xTb + (size_t)row * Cp + (base < Cp ? base : 0);
        const float4 v0 = *reinterpret_cast<const float4*>(p);
        const float4 v1 = *reinterpret_cast<const float4*>(p + 4);
        dst.v[i][0] = v0.x; dst.v[i][1] = v0.y; dst.v[i][2] = v0.z; dst.v[i][3] = v0.w;
        dst.v[i][4] = v1.x; dst.v[i][5] = v1.y; dst.v[i][6] = v1.z; dst.v[i][7] = v1.w;
    }
}

template <int NW>
__device__ __forceinline__ void recurrence_quad_body(int* lds_raw, int b, const float* __restrict__ xT, const float* __restrict__ inv,
                                                     const CorrPartials& part, const int32_t* __restrict__ mpi, int Cp, int N, int M, int Ms,
                                                     float* __restrict__ wn_out, float* __restrict__ wo_out)
{
    const RecLds rl = recurrence_prologue(lds_raw, b, inv, part, mpi, N, M);
    __shared__ float wsum[2][4];
    const int J = threadIdx.x, lane = J & 63, wv = J >> 6;
    const float* xTb = xT + (size_t)b * N * Cp;

    RowRegs<NW> o, pu[QD], pk[QD];
    load_row_quad<NW>(o, xTb, rl.kq0, Cp, J);
#pragma unroll
    for (int i = 0; i < NW; ++i)
        if ((J + 256 * i) * 8 >= Cp) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o.v[i][e] = 0.0f;
        }
#pragma unroll
    for (int d = 0; d < QD; ++d) { load_row_quad<NW>(pu[d], xTb, rl.q_s[d], Cp, J); load_row_quad<NW>(pk[d], xTb, rl.kq_s[d], Cp, J); }

    const int nsteps = M - 1;
    // The chain is bound by VALU issue (4 cycles per wave instruction), so the lanes whose chunk lies past Cp are handled
    // with 2*NW selects per step instead of one per element: their row registers hold chunk 0 of the row (finite), which
    // is multiplied by a zeroed copy of the step's scalars; their slice of o stays (+-)0 and adds nothing to the dot.
    bool live[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) live[i] = (J + 256 * i) * 8 < Cp;
    auto step = [&](const RowRegs<NW>& ru, const RowRegs<NW>& rk, int s, float iq, float v) {
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const float iqi = live[i] ? iq : 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float u = ru.v[i][e] * iqi;
                acc = __builtin_fmaf(u, o.v[i][e], acc);
            }
        }
        const float sw = wave_sum_canonical(acc);
        float* slot = wsum[s & 1];
        if (lane == 0) slot[wv] = sw;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const float at = (slot[0] + slot[1]) + (slot[2] + slot[3]);
        const float sden = at + v;
        const float wn = at / sden, wo = v / sden;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const float woi = live[i] ? wo : 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t0 = wn * o.v[i][e];
                const float t1 = woi * rk.v[i][e];
                o.v[i][e] = t0 + t1;
            }
        }
        if (J == 0) { rl.wn_s[s + 1] = wn; rl.wo_s[s + 1] = wo; }
    };

    int t = 0;
    for (; t + QD <= nsteps; t += QD) {
        // the scalars and row indices of the whole turn in one go (the barrier's memory clobber would otherwise pin one
        // LDS round trip into every step)
        float ivs[QD], vms[QD];
        int qn[QD], kn[QD];
#pragma unroll
        for (int d = 0; d < QD; ++d) { ivs[d] = rl.iv_s[t + d]; vms[d] = rl.vm_s[t + d]; qn[d] = rl.q_s[t + d + QD]; kn[d] = rl.kq_s[t + d + QD]; }
#pragma unroll
        for (int d = 0; d < QD; ++d) {
            step(pu[d], pk[d], t + d, ivs[d], vms[d]);
            load_row_quad<NW>(pu[d], xTb, qn[d], Cp, J);        // padding entries repeat a valid row
            load_row_quad<NW>(pk[d], xTb, kn[d], Cp, J);
        }
    }
#pragma unroll
    for (int d = 0; d < QD - 1; ++d)
        if (t + d < nsteps) step(pu[d], pk[d], t + d, rl.iv_s[t + d], rl.vm_s[t + d]);   // uniform across the block: every wave reaches the barrier
    __syncthreads();
    for (int l = J; l < M; l += 256) { wn_out[(size_t)b * Ms + l] = rl.wn_s[l]; wo_out[(size_t)b * Ms + l] = rl.wo_s[l]; }
}

// ---------------------------------------------------------------------------------------------------
// non-masked columns: out[c][q] = P[ind[q]][c]  (one-hot kbar column, IPSRFunction.py:129-133).
// (masked columns are written too and overwritten by recon_masked_kernel afterwards.)
__device__ __forceinline__ void gather_body(float (*tile)[33], int tileid, const float* __restrict__ xT, const CorrPartials& part,
                                            int B, int C, int Cp, int N, float* __restrict__ out)
{
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int nq = (N + 31) / 32, nc = (C + 31) / 32;
    const int q0 = (tileid % nq) * 32, c0 = ((tileid / nq) % nc) * 32, b = tileid / (nq * nc);
    (void)B;
    const float* xTb = xT + (size_t)b * N * Cp;
    // two batches of independent loads (the 4 merged row indices, then the 4 row elements) instead of 4 dependent pairs
    int row[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = q0 + ty + 8 * i;
        float v; int k = 0;
        if (q < N) merged_argmax(part, b, N, q, v, k);
        row[i] = k;
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
//   prepare_body            per sample: active-column list D (ascending k), rank of every column, jq_l = rank(kq_l),
//                           and the one-hot column counts of trunc(kbar) (non-masked q with ind[q] == k)
//                           and the one-hot CSR of trunc(kbar): one wave ranks the non-masked q by (ind[q], q) with
//                           ballot-based matching and writes the entries in ascending q — all inside the stage kernel,
//                           i.e. while the serial recurrence runs
//   attn_compress_kernel    thread per active column j replays the scalar recurrence  a = a*wn_l (+ wo_l if jq_l == j)
//                           from LDS, writes Ac coalesced over j, and builds the CSR of the entries with |a| >= 1 that
//                           survive the reference's LongTensor truncation (:36,134)
//   attn_expand_kernel      only when the caller asks for the dense [M,N] rows (tests / inspection)
//   recon_masked_kernel     out[:,q_l] = sum_j Ac[l][j] * P[D_j,:]  on fp32 MFMA, j ascending == k ascending

__device__ __forceinline__ void prepare_body(int* lds, int b, const CorrPartials& part, const int32_t* __restrict__ mpi,
                                             int N, int M, int Ms, int Mc, int nbits, int32_t* __restrict__ ind, float* __restrict__ vmax,
                                             int32_t* __restrict__ dlist, int32_t* __restrict__ mprime,
                                             int32_t* __restrict__ jq, int32_t* __restrict__ rankflag,
                                             int32_t* __restrict__ bwd_index, size_t ints_per_sample)
{
    int* flag = lds;            // [N] 1 = active column (later rank+1)
    int* cnt = lds + N;         // [N] one-hot count, later the write cursor of column k
    int* key = lds + 2 * N;     // [N] merged arg-max of q, or -1 for masked q
    int* indm = lds + 3 * N;    // [N] merged arg-max of this sample
    __shared__ int wave_tot[4], wave_cnt[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // this role is the one that materialises the merged (ind, vmax) outputs of the layer
    for (int q = tid; q < N; q += 256) {
        float v; int k;
        merged_argmax(part, b, N, q, v, k);
        indm[q] = k;
        key[q] = k;
        ind[(size_t)b * N + q] = k;
        vmax[(size_t)b * N + q] = v;
        flag[q] = 0; cnt[q] = 0;
    }
    __syncthreads();
    for (int l = tid; l < M; l += 256) { const int q = mpi_at(mpi, l, N); flag[indm[q]] = 1; key[q] = -1; }
    __syncthreads();
    for (int q = tid; q < N; q += 256)
        if (key[q] >= 0) atomicAdd(&cnt[key[q]], 1);
    __syncthreads();
    // two exclusive scans over k at once (thread t owns k in [t*KPT, (t+1)*KPT)):
    //   active-column ranks (ordered compaction -> dlist) and one-hot entry offsets (offA of the backward's CSR)
    const int KPT = (N + 255) / 256;
    const int k_lo = tid * KPT, k_hi = min(N, k_lo + KPT);
    int total = 0, ctotal = 0;
    for (int k = k_lo; k < k_hi; ++k) { total += flag[k]; ctotal += cnt[k]; }
    int incl = total, cincl = ctotal;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const int t = __shfl_up(incl, s), ct = __shfl_up(cincl, s);
        if (lane >= s) { incl += t; cincl += ct; }
    }
    if (lane == 63) { wave_tot[wv] = incl; wave_cnt[wv] = cincl; }
    __syncthreads();
    int rank = incl - total, off = cincl - ctotal;
    for (int j = 0; j < wv; ++j) { rank += wave_tot[j]; off += wave_cnt[j]; }
    int32_t* offA = bwd_index ? bwd_index + (size_t)b * ints_per_sample : nullptr;
    for (int k = k_lo; k < k_hi; ++k) {
        const int f = flag[k];
        // active column: its rank; inactive: -(number of active columns below k) - 1
        rankflag[(size_t)b * N + k] = f ? rank : -rank - 1;
        if (f) { dlist[(size_t)b * Mc + rank] = k; flag[k] = rank + 1; }     // flag now holds rank+1 for the jq lookup
        rank += f;
        const int c = cnt[k];
        if (offA) offA[k] = off;
        cnt[k] = off;                                                          // cursor of column k
        off += c;
    }
    if (tid == 255) { mprime[b] = rank; if (offA) offA[N] = off; }
    __syncthreads();
    const int mp = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    for (int j = mp + tid; j < Mc; j += 256) dlist[(size_t)b * Mc + j] = 0;     // padding rows of the GEMM: any valid patch
    for (int l = tid; l < M; l += 256) jq[(size_t)b * Ms + l] = flag[indm[mpi_at(mpi, l, N)]] - 1;

    // ---- one-hot rows of trunc(kbar): non-masked q grouped by k = ind[q], ascending q inside a group.  One wave walks
    // the positions 64 at a time; lanes with equal keys find each other with one ballot per key bit.  Keys and cursors
    // are in LDS, so the serial loop touches no global memory except the entry stores.
    if (offA && tid < 64) {
        int32_t* entA = offA + N + 1;
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
                const int base = cnt[kv];
                const int rnk = __popcll(m & lt);
                entA[base + rnk] = q;
                if (rnk == 0) cnt[kv] = base + __popcll(m);      // one leader per key; the reads above precede this write
            }
        }
    }
}

// masked positions of sample b: the shared count a.M, or its own (device array, clamped to the capacity a.M)
__device__ __forceinline__ int sample_count(const AttnArgs& a, int b)
{
    if (!a.mcount) return a.M;
    const int m = a.mcount[b];
    return m < 0 ? 0 : (m > a.M ? a.M : m);
}

// ---------------------------------------------------------------------------------------------------
// Stage kernel: ONE launch behind the correlation kernel runs three independent jobs side by side, because the
// coherent-attention recurrence is a serial chain on one wave per sample and would otherwise leave 248 CUs idle:
//   blocks [0, nrec)            recurrence_body  (nrec = B, or 0 when nothing is masked)
//   blocks [nrec, nrec + B)     prepare_body     (merged ind/vmax outputs, active columns, one-hot counts)
//   remaining blocks            gather_body      (non-masked reconstruction, one 32x32 tile each)
// All of them fold the correlation kernel's k-split partials themselves, so no merge launch is needed either.
template <int NCH, bool FULL>
__global__ void __launch_bounds__(256) attention_stage_kernel(AttnArgs a, int nrec, int nbits, size_t ints_per_sample)
{
    extern __shared__ __attribute__((aligned(16))) int lds_dyn[];
    __shared__ float gtile[32][33];
    const int bid = blockIdx.x;
    if (bid < nrec) {
        // per-sample masks: every sample has its own index row and its own number of masked positions (a.M = the capacity)
        const int M = sample_count(a, bid);
        const int32_t* mpi = a.mpi + (size_t)bid * a.mpi_stride;
        if (M <= 0) return;                                  // nothing masked in this sample: no recurrence
        if constexpr (NCH > 4) recurrence_quad_body<(NCH + 3) / 4>(lds_dyn, bid, a.xT, a.inv, a.part, mpi, a.Cp, a.N, M, a.M, a.wn, a.wo);
        else if constexpr (NCH > 3) recurrence_wide_body<NCH>(lds_dyn, bid, a.xT, a.inv, a.part, mpi, a.Cp, a.N, M, a.M, a.wn, a.wo);
        else recurrence_body<NCH, FULL>(lds_dyn, bid, a.xT, a.inv, a.part, mpi, a.Cp, a.N, M, a.M, a.wn, a.wo);
    } else if (bid < nrec + a.B) {
        const int b = bid - nrec;
        prepare_body(lds_dyn, b, a.part, a.mpi + (size_t)b * a.mpi_stride, a.N, sample_count(a, b), a.M, a.Mc, nbits, a.ind, a.vmax, a.dlist,
                     a.mprime, a.jq, a.rankflag, a.bwd_index, ints_per_sample);
    } else {
        gather_body(gtile, bid - nrec - a.B, a.xT, a.part, a.B, a.C, a.Cp, a.N, a.out);
    }
}

constexpr int AC_MAXT = 1024;   // threads per workgroup: one active column each (columns beyond 1024 loop)

// Compressed attention rows + the survivor CSR, one workgroup per sample.
//   replay: thread per active column j runs  a = a*wn_l (+ wo_l if jq_l == j)  from an LDS copy of the steps, writes
//           Ac[l][j] (coalesced over j) and counts the entries with |a| >= 1 that survive the LongTensor truncation;
//   scan  : survivor counts over the active columns (ascending j == ascending k) -> offB for EVERY k;
//   fill  : a column with survivors replays once more and writes them in ascending l.
template <bool WITH_INDEX>
__global__ void __launch_bounds__(AC_MAXT) attn_compress_kernel(const float* __restrict__ wn, const float* __restrict__ wo,
                                                                const int32_t* __restrict__ jq, const int32_t* __restrict__ mprime,
                                                                const int32_t* __restrict__ rankflag, const int32_t* __restrict__ mpi_all,
                                                                int mpi_stride, const int32_t* __restrict__ mcount,
                                                                int N, int Ms, int Mc, float* __restrict__ ac,
                                                                int32_t* __restrict__ bwd_index, size_t ints_per_sample)
{
    extern __shared__ __attribute__((aligned(16))) int lds[];
    float4* step = reinterpret_cast<float4*>(lds);          // [M] {wn, wo, jq bits, 0}
    int* offj = lds + 4 * Ms;                                // [Mc + 1] survivor count, then offset, of active column j
    __shared__ int wave_tot[AC_MAXT / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nthr = blockDim.x, nwave = nthr >> 6;
    // Ms = capacity (row strides, LDS carve, CSR layout); M = this sample's masked positions
    int M = Ms;
    if (mcount) { M = mcount[b]; M = M < 0 ? 0 : (M > Ms ? Ms : M); }
    const int32_t* mpi = mpi_all + (size_t)b * mpi_stride;
    const float* wnb = wn + (size_t)b * Ms;
    const float* wob = wo + (size_t)b * Ms;
    const int32_t* jqb = jq + (size_t)b * Ms;
    for (int l = tid; l < M; l += nthr) step[l] = make_float4(wnb[l], wob[l], __int_as_float(jqb[l]), 0.0f);
    __syncthreads();
    float* acb = ac + (size_t)b * Ms * Mc;
    const size_t capB = (size_t)Ms * (Ms + 1) / 2;
    // columns that can be non-zero: the active ones, rounded up to the GEMM's 32-column stages (the rest of a row is never
    // read by recon_masked_kernel) — with per-sample masks Mc is sized for the capacity, not for this sample
    const int Mce = min(Mc, (mprime[b] + 31) & ~31);
    int32_t* offB = WITH_INDEX ? bwd_index + (size_t)b * ints_per_sample + (N + 1) + N : nullptr;
    int32_t* entB_q = WITH_INDEX ? offB + N + 1 : nullptr;
    float* entB_w = reinterpret_cast<float*>(entB_q + capB);

    int* mq = offj + Mc + 1;                                  // [M] position q of masked step l (survivor entries name it)
    if (WITH_INDEX) {
        for (int l = tid; l < M; l += nthr) mq[l] = mpi_at(mpi, l, N);
        __syncthreads();
    }
    for (int j0 = 0; j0 < Mce; j0 += nthr) {                 // one pass for Mc <= 1024
        const int j = j0 + tid;
        // pass 1: the chain itself (mul, conditional add), the compressed row, and the NUMBER of survivors — nothing else rides
        // on the 256 dependent steps (remembering survivors in registers tripled the per-step cost, and on training features a
        // dozen columns hold 130-220 of them each, far beyond any register budget)
        float a = 0.0f;
        int cnt = 0;
        if (j < Mce) {
#pragma unroll 8
            for (int l = 0; l < M; ++l) {
                const float4 s = step[l];
                a = a * s.x;                                            // (:123)
                a = (__float_as_int(s.z) == j) ? a + s.y : a;           // (:124)
                acb[(size_t)l * Mc + j] = a;                            // (:125) compressed row l
                if (WITH_INDEX) cnt += !(fabsf(a) < 1.0f) ? 1 : 0;      // trunc(a) != 0 (NaN counts, as in the LongTensor cast's input)
            }
        }
        if (!WITH_INDEX) continue;
        // exclusive scan of the survivor counts of this batch of columns (one column per thread), on top of the
        // running total carried in offj[Mc] from the previous batch
        __syncthreads();
        int incl = cnt;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const int t = __shfl_up(incl, s);
            if (lane >= s) incl += t;
        }
        if (lane == 63) wave_tot[wv] = incl;
        __syncthreads();
        int off = (j0 == 0 ? 0 : offj[Mc]) + incl - cnt;
        for (int w = 0; w < wv; ++w) off += wave_tot[w];
        if (j < Mce) offj[j] = off;
        __syncthreads();
        if (tid == nthr - 1) {
            int tot = j0 == 0 ? 0 : offj[Mc];
            for (int w = 0; w < nwave; ++w) tot += wave_tot[w];
            offj[Mc] = tot;
        }
        // pass 2, columns with survivors only: the same chain again (same bits), entries written in ascending l
        if (j < Mce && cnt > 0) {
            float a2 = 0.0f;
            int e = off;
#pragma unroll 8
            for (int l = 0; l < M; ++l) {
                const float4 s = step[l];
                a2 = a2 * s.x;
                a2 = (__float_as_int(s.z) == j) ? a2 + s.y : a2;
                if (!(fabsf(a2) < 1.0f)) { entB_q[e] = mq[l]; entB_w[e] = truncf(a2); ++e; }
            }
        }
        __syncthreads();
    }
    if (!WITH_INDEX) return;
    // the grand total also answers "all active columns are below k" (rank == mprime): when the replay stopped short of Mc that
    // slot is offj[Mce] (mprime == Mce), and with no active column at all the total is 0
    __syncthreads();
    if (tid == 0 && Mce < Mc) offj[Mce] = Mce == 0 ? 0 : offj[Mc];
    __syncthreads();
    // offB[k] = survivors in columns < k: active column -> offj[rank], inactive -> offj[#active columns below]
    for (int k = tid; k <= N; k += nthr) {
        int r;
        if (k < N) { const int rf = rankflag[(size_t)b * N + k]; r = rf >= 0 ? rf : -rf - 1; }
        else r = mprime[b];
        offB[k] = offj[r];
    }
}

// dense rows of `in_attention` (optional output): attn[l][k] = active(k) ? Ac[l][rank(k)] : 0
__global__ void __launch_bounds__(256) attn_expand_kernel(const float* __restrict__ ac, const int32_t* __restrict__ rankflag,
                                                          const int32_t* __restrict__ mcount, int N, int Ms, int Mc, float* __restrict__ attn)
{
    const int b = blockIdx.z, l = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= N) return;
    const int M = mcount ? min(max(mcount[b], 0), Ms) : Ms;           // rows past this sample's count are zero
    const int r = rankflag[(size_t)b * N + k];
    attn[((size_t)b * Ms + l) * N + k] = (l < M && r >= 0) ? ac[((size_t)b * Ms + l) * Mc + r] : 0.0f;
}

// ---------------------------------------------------------------------------------------------------
// masked columns on the matrix cores:  Dm[c][l] = sum_j xT[D_j][c] * Ac[l][j],  out[c][mpi[l]] = Dm[c][l].
// A[i=c][kk=j] = xT[D_j][c]: patch-major rows = MFMA operand order (plain LDS copy, rows picked through D);
// B[kk=j][jn=l] = Ac[l][j] is j-contiguous, so its LDS image is [l][j] with a padded row (33) for conflict-free
// column reads.  The K loop runs over the M' active columns only (vs N for the reference's dense GEMM).
constexpr int RM_BC = 64, RM_BL = 64, RM_BK = 32;

__global__ void __launch_bounds__(256) recon_masked_kernel(const float* __restrict__ xT, const float* __restrict__ ac,
                                                           const int32_t* __restrict__ dlist, const int32_t* __restrict__ mprime,
                                                           const int32_t* __restrict__ mpi_all, int mpi_stride,
                                                           const int32_t* __restrict__ mcount, int C, int Cp, int N, int Ms, int Mc,
                                                           float* __restrict__ out)
{
    __shared__ __attribute__((aligned(16))) float As[2][RM_BK][RM_BC];
    __shared__ float Bs[2][RM_BL][RM_BK + 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    const int l0 = blockIdx.x * RM_BL, c0 = blockIdx.y * RM_BC, b = blockIdx.z;
    const int M = mcount ? min(max(mcount[b], 0), Ms) : Ms;           // this sample's masked positions (Ms = capacity / row stride)
    if (l0 >= M) return;                                              // uniform for the block
    const int32_t* mpi = mpi_all + (size_t)b * mpi_stride;
    const float* xTb = xT + (size_t)b * N * Cp;
    const float* acb = ac + (size_t)b * Ms * Mc;
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
        const int q = mpi_at(mpi, l, N);
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
    const size_t ints = 2 * ((size_t)N + 1) + (size_t)N + (size_t)M * (M + 1);     // ipsr_bwd_index_ints(N, M)
    int nbits = 1;
    while ((1 << nbits) < N) ++nbits;
    {
        const int nch = cdiv(Cp, 512);
        const size_t lds_rec = M > 0 ? (size_t)6 * (((M + 3) & ~3) + 3 * RING) * sizeof(int) : 0;
        const size_t lds_prep = (size_t)4 * N * sizeof(int);
        const size_t lds = lds_rec > lds_prep ? lds_rec : lds_prep;
        if (lds > 150 * 1024) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_forward: N=%d / M=%d too large for the stage kernel's LDS", N, M);
        const int nrec = M > 0 ? B : 0;
        const int grid = nrec + B + cdiv(N, 32) * cdiv(C, 32) * B;
#define LAUNCH_STAGE2(NCH, FULL)                                                                                     \
    do {                                                                                                             \
        if (lds > 48 * 1024)                                                                                         \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_stage_kernel<NCH, FULL>),             \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
        attention_stage_kernel<NCH, FULL><<<grid, 256, lds, st>>>(a, nrec, nbits, ints);                                          \
    } while (0)
#define LAUNCH_STAGE(NCH)                                                                                            \
    do {                                                                                                             \
        if (Cp == 512 * (NCH)) LAUNCH_STAGE2(NCH, true);                                                             \
        else LAUNCH_STAGE2(NCH, false);                                                                              \
    } while (0)
        switch (nch) {
            case 1: LAUNCH_STAGE(1); break;
            case 2: LAUNCH_STAGE(2); break;
            case 3: LAUNCH_STAGE(3); break;
            case 4: LAUNCH_STAGE(4); break;
            // wide patches (shift_sz > 1: C*p*p numbers per patch): four-wave recurrence, one instantiation per 2048 numbers
            case 5: case 6: case 7: case 8: LAUNCH_STAGE2(8, false); break;              // 2 chunks per lane of 256
            case 9: case 10: case 11: case 12: LAUNCH_STAGE2(12, false); break;          // 3 (C=512, p=3: 4608 numbers)
            case 13: case 14: case 15: case 16: LAUNCH_STAGE2(16, false); break;         // 4
            default: return fail(IPSR_ERR_UNSUPPORTED, "ipsr_forward: patch length C*p*p=%d > 8192 not supported", C);
        }
#undef LAUNCH_STAGE
#undef LAUNCH_STAGE2
        if (int rc = check_launch("attention_stage_kernel")) return rc;
    }
    const bool need_index = a.bwd_index != nullptr;
    if (M > 0) {
        const size_t lds_c = ((size_t)4 * M + Mc + 1 + M) * sizeof(int);
        if (lds_c > 150 * 1024) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_forward: M=%d too large for attn_compress_kernel", M);
        if (lds_c > 48 * 1024) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_compress_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_compress_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c);
        }
        const int nthr = Mc < AC_MAXT ? ((Mc + 63) & ~63) : AC_MAXT;
        if (need_index) attn_compress_kernel<true><<<B, nthr, lds_c, st>>>(a.wn, a.wo, a.jq, a.mprime, a.rankflag, a.mpi, a.mpi_stride, a.mcount, N, M, Mc, a.ac, a.bwd_index, ints);
        else attn_compress_kernel<false><<<B, nthr, lds_c, st>>>(a.wn, a.wo, a.jq, a.mprime, a.rankflag, a.mpi, a.mpi_stride, a.mcount, N, M, Mc, a.ac, nullptr, ints);
        if (int rc = check_launch("attn_compress_kernel")) return rc;
        recon_masked_kernel<<<dim3(cdiv(M, RM_BL), cdiv(C, RM_BC), B), 256, 0, st>>>(a.xT, a.ac, a.dlist, a.mprime, a.mpi, a.mpi_stride, a.mcount, C, Cp, N, M, Mc, a.out);
        if (int rc = check_launch("recon_masked_kernel")) return rc;
        if (a.attn) {
            attn_expand_kernel<<<dim3(cdiv(N, 256), M, B), 256, 0, st>>>(a.ac, a.rankflag, a.mcount, N, M, Mc, a.attn);
            if (int rc = check_launch("attn_expand_kernel")) return rc;
        }
    } else if (need_index) {
        // nothing masked: the survivor CSR is empty (offB = 0)
        for (int b = 0; b < B; ++b)
            if (hipMemsetAsync(a.bwd_index + (size_t)b * ints + (N + 1) + N, 0, sizeof(int32_t) * (N + 1), st) != hipSuccess)
                return fail(IPSR_ERR_LAUNCH, "ipsr_forward: hipMemsetAsync failed");
    }
    return IPSR_OK;
}

}  // namespace ipsr
