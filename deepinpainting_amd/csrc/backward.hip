// backward.hip — K8: IPSRFunction.backward (models/IPSRFunction.py:144-178).
//
//   grad_in[b,c,k] = g[b,c,k] + triple_w * sum_q W[q][k] * g[b,c,q],     W = trunc(kbar^T)
// The reference rebuilds the dense N x N matrix W with a Python loop (:160-163) and runs a dense mm
// (:169).  W is one-hot on every non-masked row (W[q][ind[q]] = 1) and, because kbar was stored in a
// LongTensor (:36,134), all-zero on masked rows except where |a_l[k]| >= 1.  So the product is a
// scatter-add; written as a GATHER over the CSR built by bwd_index_kernel it needs no atomics and is
// deterministic:   acc = sum_{q in col(k), ascending} g[c][q]  (+ the few surviving masked rows).
//
// HBM-bound: reads g once, writes grad_in once (2*C*N*4 bytes per sample); the gathered g[c][q] are
// re-reads of the 4 KB row the workgroup is streaming anyway (L1/L2 hits).
#include "ipsr_common.h"

namespace ipsr {

constexpr int BW_KT = 256;   // k columns per workgroup (one per thread)
constexpr int BW_CT = 16;    // channel rows per workgroup (accumulators in registers)

__global__ void __launch_bounds__(BW_KT) ipsr_backward_kernel(const float* __restrict__ g, const int32_t* __restrict__ bwd_index,
                                                              size_t ints_per_sample, size_t cap, float triple_w, int C, int N,
                                                              float* __restrict__ gin)
{
    const int k = blockIdx.x * BW_KT + threadIdx.x;
    const int c0 = blockIdx.y * BW_CT, b = blockIdx.z;
    if (k >= N) return;
    const int32_t* col_off = bwd_index + (size_t)b * ints_per_sample;
    const int32_t* ent_q = col_off + N + 1;
    const float* ent_w = reinterpret_cast<const float*>(ent_q + cap);
    const int e0 = col_off[k], e1 = col_off[k + 1];
    const int c_hi = min(C, c0 + BW_CT);
    const float* gb = g + ((size_t)b * C + c0) * N;
    float acc[BW_CT];
#pragma unroll
    for (int i = 0; i < BW_CT; ++i) acc[i] = 0.0f;
    // column k of trunc(kbar)^T: one-hot rows first (weight 1, ascending q), then the masked rows that survive
    // the truncation (ascending l) — one fmaf chain per output, the same order as the oracle
#pragma unroll 4
    for (int e = e0; e < e1; ++e) {
        const int q = ent_q[e];
        const float wgt = ent_w[e];
#pragma unroll
        for (int i = 0; i < BW_CT; ++i)
            if (c0 + i < c_hi) acc[i] = __builtin_fmaf(wgt, gb[(size_t)i * N + q], acc[i]);
    }
#pragma unroll
    for (int i = 0; i < BW_CT; ++i)
        if (c0 + i < c_hi) {
            const float t = acc[i] * triple_w;                 // (:173) mul then add, separately rounded
            gin[((size_t)b * C + c0 + i) * N + k] = gb[(size_t)i * N + k] + t;
        }
}

int launch_backward(const float* g, const int32_t* mpi, int M, const float* attn, const int32_t* bwd_index,
                    float triple_w, int B, int C, int N, float* gin, hipStream_t st)
{
    (void)mpi; (void)attn;      // everything the backward needs is in bwd_index
    const size_t cap = (size_t)(N - M) + (size_t)M * (M + 1) / 2;
    const size_t ints = (size_t)N + 1 + 2 * cap;
    ipsr_backward_kernel<<<dim3(cdiv(N, BW_KT), cdiv(C, BW_CT), B), BW_KT, 0, st>>>(g, bwd_index, ints, cap, triple_w, C, N, gin);
    return check_launch("ipsr_backward_kernel");
}

}  // namespace ipsr
