// backward.hip — K8: IPSRFunction.backward (models/IPSRFunction.py:144-178).
//
//   grad_in[b,c,k] = g[b,c,k] + triple_w * sum_q W[q][k] * g[b,c,q],     W = trunc(kbar^T)
// The reference rebuilds the dense N x N matrix W with a Python loop (:160-163) and runs a dense mm
// (:169).  W is one-hot on every non-masked row (W[q][ind[q]] = 1) and, because kbar was stored in a
// LongTensor (:36,134), all-zero on masked rows except where |a_l[k]| >= 1.  So the product is a
// scatter-add; written as a GATHER over the CSR built by bwd_index_kernel it needs no atomics and is
// deterministic:   acc = sum_{q in col(k), ascending} g[c][q]  (+ the few surviving masked rows).
//
// HBM-bound: reads g once, writes grad_in once (2*C*N*4 bytes per sample); the gathered g[c][q] come from the
// LDS copy of the rows the workgroup owns.
//
// identity = 0 (shift_sz > 1, see ipsr_backward_patch): the kernel runs on the UNFOLDED gradient and returns only
// triple_w * sum(...); the caller folds that back and adds g.
#include "ipsr_common.h"
#include <algorithm>

namespace ipsr {

constexpr int BW_THREADS = 256;
constexpr int BW_ROWS = 8;               // channel rows per workgroup (LDS-resident), fewer when N is large
constexpr int BW_LDS_BYTES = 66 * 1024;
constexpr int BW_PAD = 4;                  // floats of padding per LDS row: the R rows of one column then fall on distinct banks
constexpr int BW_BATCH = 32;
constexpr size_t BW_LDS_LIMIT = 150 * 1024; // dynamic LDS a workgroup may ask for (160 KB per CU, minus the static lists)
constexpr int BW_BCAP = 2048;              // survivor entries staged in LDS (8 bytes each)               // entries of a long chain fetched per round trip (phase 2)
constexpr int BW_INLINE = 8;               // entry lists up to this length are finished by the column's own thread
constexpr int BW_MAXLONG = 512;            // deferred (long) columns per workgroup

// The column walk of one workgroup.  Force-inlined once per index residency (LDS / global), so that every pointer keeps its
// address space: the chains below are bound by the latency of fetching the NEXT entry, which is ~100 cycles from LDS and
// ~1 us from L2 behind a busy step.
template <int R>
__device__ __forceinline__ void backward_columns(const float* rows, int NP, const int32_t* offA, const int32_t* entA, const int32_t* offB,
                                                 const int32_t* entB_q, const float* entB_w, int* long_k, int* n_long_p, int nrow, int N,
                                                 float triple_w, int identity, float* ob)
{
    const int tid = threadIdx.x;
    // Columns whose entry list is long (real training features are signed: attention weights leave [0,1] and a few
    // columns collect hundreds of truncation survivors — measured up to ~220 of 256 rows) are deferred to a second
    // phase where every (column, channel row) chain gets its own lane; short columns are finished inline.
    for (int k = tid; k < N; k += BW_THREADS) {
        const int a0 = offA[k], a1 = offA[k + 1], b0 = offB[k], b1 = offB[k + 1];
        if ((a1 - a0) + (b1 - b0) > BW_INLINE) {
            const int slot = atomicAdd(n_long_p, 1);
            if (slot < BW_MAXLONG) { long_k[slot] = k; continue; }       // else: list full, fall through and do it inline
        }
        float acc[R];
#pragma unroll
        for (int i = 0; i < R; ++i) acc[i] = 0.0f;
        // column k of trunc(kbar)^T: one-hot rows first (weight 1, ascending q), then the masked rows that survive
        // the truncation (ascending l) — one chain per output, the same order as the oracle
        for (int e = a0; e < a1; ++e) {
            const int q = entA[e];
#pragma unroll
            for (int i = 0; i < R; ++i) acc[i] = acc[i] + rows[(size_t)i * NP + q];                       // rows past nrow: stale LDS, never stored
        }
        for (int e = b0; e < b1; ++e) {
            const int q = entB_q[e];
            const float wgt = entB_w[e];
#pragma unroll
            for (int i = 0; i < R; ++i) acc[i] = __builtin_fmaf(wgt, rows[(size_t)i * NP + q], acc[i]);
        }
#pragma unroll
        for (int i = 0; i < R; ++i)
            if (i < nrow) {
                const float t = acc[i] * triple_w;             // (:173) mul then add, separately rounded
                ob[(size_t)i * N + k] = identity ? rows[(size_t)i * NP + k] + t : t;
            }
    }
    __syncthreads();
    // phase 2: one lane per (long column, channel row); the chain itself stays sequential (same bits as phase 1).
    // In training the conv features are signed and the matches collapse: measured (tools/instep_layer.py, profiles/
    // r02_instep_layer_before.txt) ~10 columns per sample hold ALL ~750 one-hot entries (190-600 each) and up to 6 columns
    // 130-220 survivors, so this phase is the kernel.  Entries are fetched BW_BATCH at a time (independent loads) and only
    // the dependent adds / fmas stay serial.
    const int nl = min(*n_long_p, BW_MAXLONG);
    // neighbouring lanes take the R channel rows of ONE column: the index reads of a column are one address per R lanes and the
    // padded rows put their LDS reads on distinct banks
    for (int idx = tid; idx < nl * R; idx += BW_THREADS) {
        const int k = long_k[idx / R], i = idx % R;
        if (i >= nrow) continue;
        const float* row = rows + (size_t)i * NP;
        float acc = 0.0f;
        {
            // full batches carry no bounds arithmetic at all (the chain costs one add per entry plus the two LDS reads that
            // feed it); the last, partial batch is clamped and predicated
            const int e0 = offA[k], e1 = offA[k + 1];
            int e = e0;
            for (; e + BW_BATCH <= e1; e += BW_BATCH) {
                int qn[BW_BATCH];
                float v[BW_BATCH];
#pragma unroll
                for (int j = 0; j < BW_BATCH; ++j) qn[j] = entA[e + j];
#pragma unroll
                for (int j = 0; j < BW_BATCH; ++j) v[j] = row[qn[j]];
#pragma unroll
                for (int j = 0; j < BW_BATCH; ++j) acc = acc + v[j];
            }
            if (e < e1) {
                int qn[BW_BATCH];
                float v[BW_BATCH];
#pragma unroll
                for (int j = 0; j < BW_BATCH; ++j) qn[j] = entA[min(e + j, e1 - 1)];
#pragma unroll
                for (int j = 0; j < BW_BATCH; ++j) v[j] = row[qn[j]];
#pragma unroll
                for (int j = 0; j < BW_BATCH; ++j)
                    if (e + j < e1) acc = acc + v[j];
            }
        }
        {
            constexpr int HB = BW_BATCH / 2;
            const int e0 = offB[k], e1 = offB[k + 1];
            int e = e0;
            for (; e + HB <= e1; e += HB) {
                int qn[HB];
                float v[HB], wc[HB];
#pragma unroll
                for (int j = 0; j < HB; ++j) { qn[j] = entB_q[e + j]; wc[j] = entB_w[e + j]; }
#pragma unroll
                for (int j = 0; j < HB; ++j) v[j] = row[qn[j]];
#pragma unroll
                for (int j = 0; j < HB; ++j) acc = __builtin_fmaf(wc[j], v[j], acc);
            }
            if (e < e1) {
                int qn[HB];
                float v[HB], wc[HB];
#pragma unroll
                for (int j = 0; j < HB; ++j) { const int ee = min(e + j, e1 - 1); qn[j] = entB_q[ee]; wc[j] = entB_w[ee]; }
#pragma unroll
                for (int j = 0; j < HB; ++j) v[j] = row[qn[j]];
#pragma unroll
                for (int j = 0; j < HB; ++j)
                    if (e + j < e1) acc = __builtin_fmaf(wc[j], v[j], acc);
            }
        }
        const float t = acc * triple_w;
        ob[(size_t)i * N + k] = identity ? row[k] + t : t;
    }
}

// One workgroup = R channel rows of one sample.  The rows (R x N fp32) are staged once into LDS with coalesced
// 16-byte loads — that is the ONLY read of grad_out from HBM/L2 — and every gathered g[c][q] of the scatter-add
// then comes from LDS.  Threads walk the patch index k; the CSR column of k is read once and applied to all R
// rows, so the output is written with coalesced stores.  HBM traffic = the algorithmic 2*C*N*4 bytes (+ index).
// STAGED: the sample's CSR (offsets, one-hot entries, and the survivor entries when there are at most `bcap` of them) is
// copied into LDS beside the rows, so that no step of any chain waits on L2.
template <int R, bool STAGED>
__global__ void __launch_bounds__(BW_THREADS) ipsr_backward_kernel(const float* __restrict__ g, const int32_t* __restrict__ bwd_index,
                                                                   size_t ints_per_sample, size_t capB, float triple_w, int C, int N,
                                                                   int identity, int bcap, float* __restrict__ gin)
{
    extern __shared__ __attribute__((aligned(16))) float rows[];      // [R][N + BW_PAD], then the staged index
    const int NP = N + BW_PAD;
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * R, b = blockIdx.y;
    const int nrow = min(R, C - c0);
    const float* gb = g + ((size_t)b * C + c0) * N;
    float* ob = gin + ((size_t)b * C + c0) * N;

    const int32_t* offA = bwd_index + (size_t)b * ints_per_sample;
    const int32_t* entA = offA + N + 1;
    const int32_t* offB = entA + N;
    const int32_t* entB_q = offB + N + 1;
    const float* entB_w = reinterpret_cast<const float*>(entB_q + capB);

    __shared__ int long_k[BW_MAXLONG];
    __shared__ int n_long;
    if (tid == 0) n_long = 0;

    int32_t* s_offA = reinterpret_cast<int32_t*>(rows + (size_t)R * NP);
    int32_t* s_entA = s_offA + N + 1;
    int32_t* s_offB = s_entA + N;
    int32_t* s_entBq = s_offB + N + 1;
    float* s_entBw = reinterpret_cast<float*>(s_entBq + bcap);
    int totB = 0;
    if (STAGED) {
        totB = offB[N];
        for (int i = tid; i < 3 * N + 2; i += BW_THREADS) s_offA[i] = offA[i];          // offA | entA | offB are contiguous
        if (totB <= bcap)
            for (int i = tid; i < totB; i += BW_THREADS) { s_entBq[i] = entB_q[i]; s_entBw[i] = entB_w[i]; }
    }

    const size_t total = (size_t)nrow * N;
    if ((N & 3) == 0) {
        // 8 independent 16-byte loads in flight per lane before the first LDS store (a load-store loop would pay one HBM
        // round trip per iteration: with one workgroup per CU nothing else hides it)
        const int n4 = N >> 2;
        const int tot4 = (int)(total >> 2);
        for (int i0 = tid; i0 < tot4; i0 += BW_THREADS * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * BW_THREADS;
                v[u] = *reinterpret_cast<const float4*>(gb + 4 * (size_t)min(i, tot4 - 1));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * BW_THREADS;
                if (i < tot4) {
                    const int r = i / n4, c4 = i - r * n4;
                    *reinterpret_cast<float4*>(&rows[(size_t)r * NP + 4 * c4]) = v[u];
                }
            }
        }
    } else {
        for (size_t i = tid; i < total; i += BW_THREADS) { const int r = (int)(i / N); rows[(size_t)r * NP + (i - (size_t)r * N)] = gb[i]; }
    }
    __syncthreads();

    if (STAGED && totB <= bcap)
        backward_columns<R>(rows, NP, s_offA, s_entA, s_offB, s_entBq, s_entBw, long_k, &n_long, nrow, N, triple_w, identity, ob);
    else if (STAGED)
        backward_columns<R>(rows, NP, s_offA, s_entA, s_offB, entB_q, entB_w, long_k, &n_long, nrow, N, triple_w, identity, ob);
    else
        backward_columns<R>(rows, NP, offA, entA, offB, entB_q, entB_w, long_k, &n_long, nrow, N, triple_w, identity, ob);
}

int launch_backward(const float* g, const int32_t* mpi, int M, const float* attn, const int32_t* bwd_index,
                    float triple_w, int B, int C, int N, float* gin, hipStream_t st, int identity)
{
    (void)mpi; (void)attn;      // everything the backward needs is in bwd_index
    const size_t capB = (size_t)M * (M + 1) / 2;
    const size_t ints = 2 * ((size_t)N + 1) + (size_t)N + 2 * capB;
    int R = BW_ROWS;
    while (R > 1 && (size_t)R * (N + BW_PAD) * sizeof(float) > BW_LDS_BYTES) R >>= 1;
    const size_t rows_bytes = (size_t)R * (N + BW_PAD) * sizeof(float);
    if (rows_bytes > BW_LDS_LIMIT) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_backward: N=%d too large for one LDS-resident row", N);
    // the sample's CSR rides in LDS beside the rows when it fits: offsets + one-hot entries always (3N+2 ints), the
    // survivor entries up to bcap of them (more than that: read from L2, as everything is when even the offsets do not fit)
    const size_t idx_bytes = (3 * (size_t)N + 2) * sizeof(int32_t);
    const bool staged = rows_bytes + idx_bytes <= BW_LDS_LIMIT;
    int bcap = 0;
    if (staged) bcap = (int)std::min<size_t>({(size_t)BW_BCAP, capB, (BW_LDS_LIMIT - rows_bytes - idx_bytes) / 8});
    const size_t lds = rows_bytes + (staged ? idx_bytes + (size_t)bcap * 8 : 0);
#define LAUNCH_BW2(RR, ST)                                                                                         \
    do {                                                                                                           \
        if (lds > 48 * 1024)                                                                                       \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ipsr_backward_kernel<RR, ST>),                \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
        ipsr_backward_kernel<RR, ST><<<dim3(cdiv(C, RR), B), BW_THREADS, lds, st>>>(g, bwd_index, ints, capB, triple_w, C, N, identity, bcap, gin); \
    } while (0)
#define LAUNCH_BW(RR)                                                                                              \
    do {                                                                                                           \
        if (staged) LAUNCH_BW2(RR, true); else LAUNCH_BW2(RR, false);                                              \
    } while (0)
    switch (R) {
        case 16: LAUNCH_BW(16); break;
        case 8: LAUNCH_BW(8); break;
        case 4: LAUNCH_BW(4); break;
        case 2: LAUNCH_BW(2); break;
        default: LAUNCH_BW(1); break;
    }
#undef LAUNCH_BW
#undef LAUNCH_BW2
    return check_launch("ipsr_backward_kernel");
}

}  // namespace ipsr
