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
//   * column_fill_kernel — a_l[k] is a scalar recurrence per k, independent across k: one thread per k
//     replays (wn_l, wo_l) from LDS and writes the reference's `in_attention` rows [M,N], coalesced.
//   * recon_gather_kernel — non-masked q: kbar column is one-hot, so out[:,q] = P[ind[q]] is a row
//     gather (LDS-transposed so that both the read of xT rows and the write of out rows are coalesced).
//   * recon_masked_kernel — masked q: out[:,q_l] = sum_k a_l[k] * P[k,:], the dense part of the
//     reference's second GEMM, on fp32 MFMA with k walked in ascending order (one fmaf chain per
//     output, same bits as the oracle).
//   * column_count / index_scan / column_fill — kbar is kept by the reference in a LongTensor (:36,134), i.e.
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
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ql = ty + 8 * i, q = q0 + ql;
        float v = 0.0f;
        if (q < N && c0 + tx < Cp) v = xTb[(size_t)ind[(size_t)b * N + q] * Cp + c0 + tx];
        tile[ql][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cl = ty + 8 * i, c = c0 + cl;
        if (c < C && q0 + tx < N) out[((size_t)b * C + c) * N + q0 + tx] = tile[tx][cl];
    }
}

// ---------------------------------------------------------------------------------------------------
// masked columns on the matrix cores:  D[c][l] = sum_k xT[k][c] * attn[l][k],  out[c][mpi[l]] = D[c][l].
// A[i=c][kk=k] = xT[k][c] is patch-major = MFMA operand order (plain LDS copy); B[kk=k][j=l] = attn[l][k]
// is k-contiguous, so its LDS image is [l][k] with a padded row (33) for conflict-free column reads.
constexpr int RM_BC = 64, RM_BL = 64, RM_BK = 32;

__global__ void __launch_bounds__(256) recon_masked_kernel(const float* __restrict__ xT, const float* __restrict__ attn,
                                                           const int32_t* __restrict__ mpi, int C, int Cp, int N, int M,
                                                           float* __restrict__ out)
{
    __shared__ __attribute__((aligned(16))) float As[2][RM_BK][RM_BC];
    __shared__ float Bs[2][RM_BL][RM_BK + 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    const int l0 = blockIdx.x * RM_BL, c0 = blockIdx.y * RM_BC, b = blockIdx.z;
    const float* xTb = xT + (size_t)b * N * Cp;
    const float* ab = attn + (size_t)b * M * N;
    const bool vec_ok = (N % 4 == 0);

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;

    float4 ra[2], rb[2];
    auto gload = [&](int s) {
        const int k0 = s * RM_BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
            {   // A: 32 rows (k) x 16 float4 (c)
                const int kk = idx >> 4, c4 = (idx & 15) * 4;
                const int k = k0 + kk, c = c0 + c4;
                if (k < N && c + 4 <= Cp) ra[i] = *reinterpret_cast<const float4*>(xTb + (size_t)k * Cp + c);
                else ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            {   // B: 64 rows (l) x 8 float4 (k)
                const int j = idx >> 3, k4 = (idx & 7) * 4;
                const int l = l0 + j, k = k0 + k4;
                if (l < M && vec_ok && k + 4 <= N) rb[i] = *reinterpret_cast<const float4*>(ab + (size_t)l * N + k);
                else {
                    float t[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = (l < M && k + e < N) ? ab[(size_t)l * N + k + e] : 0.0f;
                    rb[i] = make_float4(t[0], t[1], t[2], t[3]);
                }
            }
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
            *reinterpret_cast<float4*>(&As[buf][idx >> 4][(idx & 15) * 4]) = ra[i];
            const int j = idx >> 3, k4 = (idx & 7) * 4;
            Bs[buf][j][k4 + 0] = rb[i].x; Bs[buf][j][k4 + 1] = rb[i].y;
            Bs[buf][j][k4 + 2] = rb[i].z; Bs[buf][j][k4 + 3] = rb[i].w;
        }
    };

    const int nstage = (N + RM_BK - 1) / RM_BK;
    gload(0);
    sstore(0);
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
// Attention rows + the sparse form of trunc(kbar).
//
// a_l[k] (the reference's `in_attention`, IPSRFunction.py:100,123-125) is a scalar recurrence per column k,
// independent across k:  a = a*wn_l;  if (kq_l == k) a += wo_l.  One thread owns one column and replays the
// M steps from an LDS copy of (wn, wo, kq) — M LDS broadcasts, no global latency in the loop.
//
// trunc(kbar) per sample (int32 words; layout shared with the oracle):
//   col_off[N+1] | ent_q[cap] | ent_w[cap] (fp32 bits),   cap = (N-M) + M(M+1)/2
// Column k = the non-masked q with ind[q] == k (weight 1, ascending q: found by scanning an LDS copy of the
// keys ind[q], -1 for masked q) followed by the masked rows with |a_l[k]| >= 1 (weight trunc(a_l[k]),
// ascending l: found while replaying the recurrence).  Three launches:
//   column_count_kernel  per column: #one-hot + #survivors
//   index_scan_kernel    exclusive scan over k -> col_off
//   column_fill_kernel   writes the attention rows [M,N] (coalesced over k) and the CSR entries in order
constexpr int IX_COLS = 128;

struct ColumnLds {
    int* key;        // [Npad]
    float4* step;    // [M]  {wn, wo, kq bits, 0}
};

__device__ __forceinline__ ColumnLds load_column_lds(int* lds, const int32_t* __restrict__ ind, const int32_t* __restrict__ mpi,
                                                     const float* __restrict__ wn, const float* __restrict__ wo,
                                                     const int32_t* __restrict__ kq, int N, int Npad, int M)
{
    ColumnLds L;
    L.step = reinterpret_cast<float4*>(lds);
    L.key = lds + 4 * M;
    // batches of 4 independent global loads per thread (a plain strided loop would serialise the round trips)
    for (int q0 = 0; q0 < Npad; q0 += 4 * IX_COLS) {
        int v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int q = q0 + j * IX_COLS + threadIdx.x; v[j] = q < N ? ind[q] : -1; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int q = q0 + j * IX_COLS + threadIdx.x; if (q < Npad) L.key[q] = v[j]; }
    }
    for (int l0 = 0; l0 < M; l0 += 2 * IX_COLS) {
        float4 v[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int l = l0 + j * IX_COLS + threadIdx.x;
            v[j] = l < M ? make_float4(wn[l], wo[l], __int_as_float(kq[l]), 0.0f) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) { const int l = l0 + j * IX_COLS + threadIdx.x; if (l < M) L.step[l] = v[j]; }
    }
    __syncthreads();
    for (int l0 = 0; l0 < M; l0 += 4 * IX_COLS) {
        int v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int l = l0 + j * IX_COLS + threadIdx.x; v[j] = l < M ? mpi[l] : -1; }
#pragma unroll
        for (int j = 0; j < 4; ++j) if (v[j] >= 0) L.key[v[j]] = -1;
    }
    __syncthreads();
    return L;
}

__global__ void __launch_bounds__(IX_COLS) column_count_kernel(const int32_t* __restrict__ ind, const int32_t* __restrict__ mpi,
                                                               const float* __restrict__ wn, const float* __restrict__ wo,
                                                               const int32_t* __restrict__ kq, int N, int M,
                                                               int32_t* __restrict__ col_cnt)
{
    extern __shared__ __attribute__((aligned(16))) int lds[];
    const int b = blockIdx.y, Npad = (N + 3) & ~3;
    const ColumnLds L = load_column_lds(lds, ind + (size_t)b * N, mpi, wn + (size_t)b * M, wo + (size_t)b * M, kq + (size_t)b * M, N, Npad, M);
    const int k = blockIdx.x * IX_COLS + threadIdx.x;
    if (k >= N) return;
    int cnt = 0;
#pragma unroll 8
    for (int q = 0; q < Npad; q += 4) {
        const int4 v = *reinterpret_cast<const int4*>(&L.key[q]);
        cnt += (v.x == k) + (v.y == k) + (v.z == k) + (v.w == k);
    }
    float a = 0.0f;
#pragma unroll 8
    for (int l = 0; l < M; ++l) {
        const float4 s = L.step[l];
        a = a * s.x;
        a = (__float_as_int(s.z) == k) ? a + s.y : a;
        cnt += (truncf(a) != 0.0f) ? 1 : 0;
    }
    col_cnt[(size_t)b * N + k] = cnt;
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

template <bool WITH_INDEX>
__global__ void __launch_bounds__(IX_COLS) column_fill_kernel(const int32_t* __restrict__ ind, const int32_t* __restrict__ mpi,
                                                              const float* __restrict__ wn, const float* __restrict__ wo,
                                                              const int32_t* __restrict__ kq, int N, int M,
                                                              float* __restrict__ attn, int32_t* __restrict__ bwd_index,
                                                              size_t ints_per_sample, size_t cap)
{
    extern __shared__ __attribute__((aligned(16))) int lds[];
    const int b = blockIdx.y, Npad = (N + 3) & ~3;
    const ColumnLds L = load_column_lds(lds, ind + (size_t)b * N, mpi, wn + (size_t)b * M, wo + (size_t)b * M, kq + (size_t)b * M, N, Npad, M);
    const int k = blockIdx.x * IX_COLS + threadIdx.x;
    if (k >= N) return;
    int32_t* ent_q = nullptr;
    float* ent_w = nullptr;
    int e = 0;
    if (WITH_INDEX) {
        int32_t* col_off = bwd_index + (size_t)b * ints_per_sample;
        ent_q = col_off + N + 1;
        ent_w = reinterpret_cast<float*>(ent_q + cap);
        e = col_off[k];
#pragma unroll 8
        for (int q = 0; q < Npad; q += 4) {
            const int4 v = *reinterpret_cast<const int4*>(&L.key[q]);
            if (v.x == k) { ent_q[e] = q;     ent_w[e] = 1.0f; ++e; }
            if (v.y == k) { ent_q[e] = q + 1; ent_w[e] = 1.0f; ++e; }
            if (v.z == k) { ent_q[e] = q + 2; ent_w[e] = 1.0f; ++e; }
            if (v.w == k) { ent_q[e] = q + 3; ent_w[e] = 1.0f; ++e; }
        }
    }
    float* ab = attn + (size_t)b * M * N;
    float a = 0.0f;
#pragma unroll 8
    for (int l = 0; l < M; ++l) {
        const float4 s = L.step[l];
        a = a * s.x;                                            // (:123)
        a = (__float_as_int(s.z) == k) ? a + s.y : a;           // (:124)
        ab[(size_t)l * N + k] = a;                              // (:125) row l of in_attention
        if (WITH_INDEX) {
            const float t = truncf(a);
            if (t != 0.0f) { ent_q[e] = mpi[l]; ent_w[e] = t; ++e; }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
int launch_attention(const AttnArgs& a, hipStream_t st)
{
    const int B = a.B, C = a.C, Cp = a.Cp, N = a.N, M = a.M;
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
    {
        const size_t lds = ((size_t)((N + 3) & ~3) + 4 * (size_t)M) * sizeof(int);
        if (lds > 150 * 1024) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_forward: N=%d, M=%d too large for the column kernels' LDS", N, M);
        if (lds > 48 * 1024) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&column_count_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&column_fill_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&column_fill_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }
        const dim3 grid(cdiv(N, IX_COLS), B);
        if (a.bwd_index) {
            column_count_kernel<<<grid, IX_COLS, lds, st>>>(a.ind, a.mpi, a.wn, a.wo, a.kq, N, M, a.col_cnt);
            if (int rc = check_launch("column_count_kernel")) return rc;
            index_scan_kernel<<<B, 1024, 0, st>>>(a.col_cnt, N, a.bwd_index, ints);
            if (int rc = check_launch("index_scan_kernel")) return rc;
            column_fill_kernel<true><<<grid, IX_COLS, lds, st>>>(a.ind, a.mpi, a.wn, a.wo, a.kq, N, M, a.attn, a.bwd_index, ints, cap);
            if (int rc = check_launch("column_fill_kernel")) return rc;
        } else if (M > 0) {
            column_fill_kernel<false><<<grid, IX_COLS, lds, st>>>(a.ind, a.mpi, a.wn, a.wo, a.kq, N, M, a.attn, nullptr, ints, cap);
            if (int rc = check_launch("column_fill_kernel")) return rc;
        }
    }
    recon_gather_kernel<<<dim3(cdiv(N, 32), cdiv(C, 32), B), 256, 0, st>>>(a.xT, a.ind, C, Cp, N, a.out);
    if (int rc = check_launch("recon_gather_kernel")) return rc;
    if (M > 0) {
        recon_masked_kernel<<<dim3(cdiv(M, RM_BL), cdiv(C, RM_BC), B), 256, 0, st>>>(a.xT, a.attn, a.mpi, C, Cp, N, M, a.out);
        if (int rc = check_launch("recon_masked_kernel")) return rc;
    }
    return IPSR_OK;
}

}  // namespace ipsr
