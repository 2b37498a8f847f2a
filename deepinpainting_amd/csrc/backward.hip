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
constexpr int BW_CT = 16;    // channel rows per workgroup
constexpr int BW_MAXE = 4;   // CSR entries cached in registers per column

__global__ void __launch_bounds__(BW_KT) ipsr_backward_kernel(const float* __restrict__ g, const int32_t* __restrict__ mpi,
                                                              int M, const float* __restrict__ attn,
                                                              const int32_t* __restrict__ bwd_index, size_t ints_per_sample,
                                                              float triple_w, int C, int N, float* __restrict__ gin)
{
    const int k = blockIdx.x * BW_KT + threadIdx.x;
    const int c0 = blockIdx.y * BW_CT, b = blockIdx.z;
    if (k >= N) return;
    const int32_t* col_off = bwd_index + (size_t)b * ints_per_sample;
    const int32_t* col_q = col_off + N + 1;
    const int nz = col_q[N];
    const int32_t* nz_rows = col_q + N + 1;
    const float* ab = attn + (size_t)b * M * N;

    const int e0 = col_off[k], e1 = col_off[k + 1];
    int qe[BW_MAXE];
#pragma unroll
    for (int i = 0; i < BW_MAXE; ++i) qe[i] = (e0 + i < e1) ? col_q[e0 + i] : -1;

    const int c_hi = min(C, c0 + BW_CT);
    for (int c = c0; c < c_hi; ++c) {
        const float* gr = g + ((size_t)b * C + c) * N;
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < BW_MAXE; ++i)
            if (qe[i] >= 0) acc = acc + gr[qe[i]];
        for (int e = e0 + BW_MAXE; e < e1; ++e) acc = acc + gr[col_q[e]];
        for (int i = 0; i < nz; ++i) {
            const int l = nz_rows[i];
            acc = __builtin_fmaf(truncf(ab[(size_t)l * N + k]), gr[mpi[l]], acc);
        }
        const float t = acc * triple_w;                    // (:173) mul then add, separately rounded
        gin[((size_t)b * C + c) * N + k] = gr[k] + t;
    }
}

int launch_backward(const float* g, const int32_t* mpi, int M, const float* attn, const int32_t* bwd_index,
                    float triple_w, int B, int C, int N, float* gin, hipStream_t st)
{
    const size_t ints = (size_t)2 * N + 2 + M;
    ipsr_backward_kernel<<<dim3(cdiv(N, BW_KT), cdiv(C, BW_CT), B), BW_KT, 0, st>>>(g, mpi, M, attn, bwd_index, ints,
                                                                                  triple_w, C, N, gin);
    return check_launch("ipsr_backward_kernel");
}

}  // namespace ipsr
