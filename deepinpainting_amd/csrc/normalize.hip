// normalize.hip — K3: patch unfold + per-patch L2 normalisation (patch size 1).
//
// Reference: NonparametricShift._extract_patches/_build (util/NonparametricShift.py:36-73):
//   P[k,:] = x[:,k];   Pn[k] = P[k] * (1 / (||P[k]||_2 + 1e-8))
//
// Outputs (all per sample):
//   inv [N]      1/(||x[:,k]|| + 1e-8)
//   xn  [C,N]    x * inv, channel-major — the A operand of the correlation GEMM (its LDS image is a plain
//                copy of the global tile, so it can be staged with wide coalesced loads)
//   xT  [N,Cp]   raw patches, patch-major (the reference's `patches_all` [N,C,1,1]), zero padded to
//                Cp = roundup(C,8): rows of it feed the recurrence and the reconstruction.
//
// Canonical summation order of the squared norm (oracle ipsr_patch_normalize_cpu): the C
// channels are cut into 8 contiguous segments of L = ceil(C/8); each segment is one fmaf chain in
// ascending c, and the 8 partials are added in ascending order.
//
// HBM-bound: reads x once from HBM (the second pass hits L2), writes xn + xT: 3*C*N*4 bytes per sample.
#include "ipsr_common.h"

namespace ipsr {

constexpr int NCOL = 32;   // patches (columns) per workgroup
constexpr int NSEG = 8;    // channel segments

// ldx / ldn: row strides of x and xn (>= N).  With ldn > N the columns [N, ldn) of xn are written as zeros: the
// correlation kernel's fast path wants whole 128-column tiles (shift_sz > 1 window grids are ragged).
__global__ void __launch_bounds__(NCOL * NSEG) patch_normalize_kernel(const float* __restrict__ x, int C, int N, int Cp,
                                                                      int ldx, int ldn,
                                                                      float* __restrict__ xn, float* __restrict__ xT,
                                                                      float* __restrict__ inv)
{
    __shared__ float part[NSEG][NCOL];
    __shared__ float inv_s[NCOL];
    __shared__ float tile[32][NCOL + 1];

    const int tid = threadIdx.x;
    const int col = tid & (NCOL - 1), seg = tid / NCOL;
    const int ntile = (ldn + NCOL - 1) / NCOL;
    const int b = blockIdx.x / ntile, k0 = (blockIdx.x % ntile) * NCOL;
    const int k = k0 + col;
    const float* xb = x + (size_t)b * C * ldx;

    // phase 1: segment partial of sum(x^2), one fmaf chain per (segment, column)
    const int L = (C + NSEG - 1) / NSEG;
    const int c_lo = seg * L, c_hi = min(C, c_lo + L);
    float acc = 0.0f;
    if (k < N) {
        int c = c_lo;
        for (; c + 4 <= c_hi; c += 4) {
            const float v0 = xb[(size_t)(c + 0) * ldx + k], v1 = xb[(size_t)(c + 1) * ldx + k];
            const float v2 = xb[(size_t)(c + 2) * ldx + k], v3 = xb[(size_t)(c + 3) * ldx + k];
            acc = __builtin_fmaf(v0, v0, acc);
            acc = __builtin_fmaf(v1, v1, acc);
            acc = __builtin_fmaf(v2, v2, acc);
            acc = __builtin_fmaf(v3, v3, acc);
        }
        for (; c < c_hi; ++c) { const float v = xb[(size_t)c * ldx + k]; acc = __builtin_fmaf(v, v, acc); }
    }
    part[seg][col] = acc;
    __syncthreads();
    if (seg == 0) {
        float tot = part[0][col];
#pragma unroll
        for (int s = 1; s < NSEG; ++s) tot = tot + part[s][col];
        const float iv = 1.0f / (sqrtf(tot) + 1e-8f);
        inv_s[col] = iv;
        if (k < N) inv[(size_t)b * N + k] = iv;
    }
    __syncthreads();

    // phase 2: scale (channel-major) and transpose (patch-major), 32 channels at a time
    const float iv = inv_s[col];
    float* xnb = xn + (size_t)b * C * ldn;
    float* xTb = xT ? xT + (size_t)b * N * Cp : nullptr;
    const int r = seg;   // 8 channel rows per pass
    for (int c0 = 0; c0 < Cp; c0 += 32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cl = r + 8 * i, c = c0 + cl;
            float v = 0.0f;
            if (c < C && k < N) {
                v = xb[(size_t)c * ldx + k];
                xnb[(size_t)c * ldn + k] = v * iv;
            } else if (c < C && k < ldn) {
                xnb[(size_t)c * ldn + k] = 0.0f;
            }
            tile[cl][col] = v;
        }
        if (xTb) {
            __syncthreads();
            const int cc = col, c = c0 + cc;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kl = r + 8 * i;
                if (c < Cp && k0 + kl < N) xTb[(size_t)(k0 + kl) * Cp + c] = tile[cc][kl];
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Register-resident variant for C <= 512 (segment length L <= 64): a thread keeps its (segment, column) slice
// of x in registers between the norm and the scaling, so x is read from HBM exactly once and nothing goes
// through LDS on the way to xn.  The patch-major copy xT is transposed through an LDS tile so that it leaves the
// CU as full 1-KiB wave stores (see the comment at the store).
constexpr int LMAX = 64;

template <int L>
__global__ void __launch_bounds__(NCOL * NSEG) patch_normalize_reg_kernel(const float* __restrict__ x, int C, int N, int Cp,
                                                                          float* __restrict__ xn, float* __restrict__ xT,
                                                                          float* __restrict__ inv)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];     // [NCOL][Cp + 4] staging of the patch-major rows
    __shared__ float part[NSEG][NCOL];
    __shared__ float inv_s[NCOL];
    const int tid = threadIdx.x;
    const int col = tid & (NCOL - 1), seg = tid / NCOL;
    const int ntile = (N + NCOL - 1) / NCOL;
    const int b = blockIdx.x / ntile, k0 = (blockIdx.x % ntile) * NCOL;
    const int k = k0 + col;
    const bool kin = k < N;
    const float* xb = x + (size_t)b * C * N;
    const int c_lo = seg * L;                      // L = ceil(C/8) is a template constant: no per-load guards when C == 8*L
    const bool full = (C == NSEG * L);

    float v[L];
    if (full && kin) {
#pragma unroll
        for (int i = 0; i < L; ++i) v[i] = xb[(size_t)(c_lo + i) * N + k];
    } else {
#pragma unroll
        for (int i = 0; i < L; ++i) v[i] = (kin && c_lo + i < C) ? xb[(size_t)(c_lo + i) * N + k] : 0.0f;
    }
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < L; ++i) acc = __builtin_fmaf(v[i], v[i], acc);        // trailing zeros leave the chain unchanged
    part[seg][col] = acc;
    __syncthreads();
    if (seg == 0) {
        float tot = part[0][col];
#pragma unroll
        for (int s = 1; s < NSEG; ++s) tot = tot + part[s][col];
        const float iv = 1.0f / (sqrtf(tot) + 1e-8f);
        inv_s[col] = iv;
        if (kin) inv[(size_t)b * N + k] = iv;
    }
    __syncthreads();
    const float iv = inv_s[col];
    float* xnb = xn + (size_t)b * C * N;
    if (kin) {
#pragma unroll
        for (int i = 0; i < L; ++i)
            if (full || c_lo + i < C) xnb[(size_t)(c_lo + i) * N + k] = v[i] * iv;
    }
    if (xT) {
        // Patch-major copy through LDS.  Writing the 256-byte register slice straight to xT[k][seg*64..] made every
        // lane store 16 B into a different cache line: PMC WRITE_SIZE showed 99 MB for 32 MB of payload.  Instead the
        // workgroup builds its [32 patches][Cp] tile in LDS (row stride Cp+4 floats: ds_write_b128 lanes land in 16
        // distinct 16-byte bank groups) and streams it out with full 1-KiB wave stores.
        const int stride = Cp + 4;
        float* trow = tile + (size_t)col * stride + c_lo;
#pragma unroll
        for (int i = 0; i < L; i += 4)
            if (full || c_lo + i < Cp) *reinterpret_cast<float4*>(trow + i) = make_float4(v[i], v[i + 1], v[i + 2], v[i + 3]);
        __syncthreads();
        const int nvec = Cp >> 2;                                  // float4 per patch row
        float* xTb = xT + ((size_t)b * N + k0) * Cp;
        for (int idx = tid; idx < NCOL * nvec; idx += NCOL * NSEG) {
            const int row = idx / nvec, c4 = (idx - row * nvec) * 4;
            if (k0 + row < N)
                *reinterpret_cast<float4*>(xTb + (size_t)row * Cp + c4) = *reinterpret_cast<const float4*>(tile + (size_t)row * stride + c4);
        }
    }
}

int launch_patch_normalize(const float* x, int B, int C, int N, float* xn, float* xT, int Cp, float* inv,
                           hipStream_t st, int ldx, int ldn)
{
    if (ldx <= 0) ldx = N;
    if (ldn <= 0) ldn = N;
    const int ntile = cdiv(N, NCOL);
    const int L = cdiv(C, NSEG);
    // register path: segment fits 64 registers; float4 rows of xT need L % 4 == 0 and 8*L >= Cp (all of the padded row written)
    if (ldx == N && ldn == N && L <= LMAX && L % 4 == 0 && NSEG * L >= Cp) {
        const int grid = B * ntile;
        const size_t lds = xT ? (size_t)NCOL * (Cp + 4) * sizeof(float) : 0;
        switch (L) {
#define NORM_CASE(LL)                                                                                             \
    case LL:                                                                                                      \
        if (lds > 48 * 1024)                                                                                      \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_normalize_reg_kernel<LL>),             \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                      \
        patch_normalize_reg_kernel<LL><<<grid, NCOL * NSEG, lds, st>>>(x, C, N, Cp, xn, xT, inv);                 \
        return check_launch("patch_normalize_reg_kernel")
            NORM_CASE(64); NORM_CASE(32); NORM_CASE(16); NORM_CASE(8); NORM_CASE(4);
#undef NORM_CASE
            default: break;      // other lengths: generic kernel below
        }
    }
    patch_normalize_kernel<<<B * cdiv(ldn, NCOL), NCOL * NSEG, 0, st>>>(x, C, N, Cp, ldx, ldn, xn, xT, inv);
    return check_launch("patch_normalize_kernel");
}

// ---------------------------------------------------------------------------------------------------
// shift_sz > 1, the shifted-sum form (see window_corr_argmax in oracle/ipsr_oracle.c and corr_argmax.hip):
//   chan_sumsq_kernel      n1[b][a] = squared channel norm of position a, in the canonical order of the p = 1 normalisation
//   window_inv_kernel      inv[b][k'] = 1 / (sqrt(sum_{dy,dx} n1[(ky+dy)*w + kx+dx]) + 1e-8)
//   unfold_patchmajor_kernel   xT[b][k'][(c*p+dy)*p+dx] = x[b][c][ky+dy][kx+dx]   (raw windows, patch-major: the recurrence /
//                          gather / reconstruction operand).  The normalised unfolded matrix xn is no longer needed.
__global__ void __launch_bounds__(NCOL * NSEG) chan_sumsq_kernel(const float* __restrict__ x, int C, int N, float* __restrict__ n1)
{
    __shared__ float part[NSEG][NCOL];
    const int tid = threadIdx.x;
    const int col = tid & (NCOL - 1), seg = tid / NCOL;
    const int ntile = (N + NCOL - 1) / NCOL;
    const int b = blockIdx.x / ntile, k = (blockIdx.x % ntile) * NCOL + col;
    const float* xb = x + (size_t)b * C * N;
    const int L = (C + NSEG - 1) / NSEG;
    const int c_lo = seg * L, c_hi = min(C, c_lo + L);
    float acc = 0.0f;
    if (k < N)
        for (int c = c_lo; c < c_hi; ++c) { const float v = xb[(size_t)c * N + k]; acc = __builtin_fmaf(v, v, acc); }
    part[seg][col] = acc;
    __syncthreads();
    if (seg == 0 && k < N) {
        float tot = part[0][col];
#pragma unroll
        for (int s2 = 1; s2 < NSEG; ++s2) tot = tot + part[s2][col];
        n1[(size_t)b * N + k] = tot;
    }
}

__global__ void __launch_bounds__(256) window_inv_kernel(const float* __restrict__ n1, int h, int w, int patch, int nW, int Np,
                                                         float* __restrict__ inv)
{
    const int k = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (k >= Np) return;
    const float* nb = n1 + (size_t)b * h * w + (size_t)(k / nW) * w + k % nW;
    float tot = 0.0f;
    bool first = true;
    for (int dy = 0; dy < patch; ++dy)
        for (int dx = 0; dx < patch; ++dx) { const float v = nb[dy * w + dx]; tot = first ? v : tot + v; first = false; }
    inv[(size_t)b * Np + k] = 1.0f / (sqrtf(tot) + 1e-8f);
}

__global__ void __launch_bounds__(NCOL * NSEG) unfold_patchmajor_kernel(const float* __restrict__ x, int C, int h, int w, int patch, int nW,
                                                                        int N, int Cp, float* __restrict__ xT)
{
    __shared__ float tile[32][NCOL + 1];
    const int tid = threadIdx.x;
    const int col = tid & (NCOL - 1), seg = tid / NCOL;
    const int ntile = (N + NCOL - 1) / NCOL;
    const int b = blockIdx.x / ntile, k0 = (blockIdx.x % ntile) * NCOL;
    const int k = k0 + col;
    const int pp = patch * patch, K = C * pp;
    const float* xb = x + (size_t)b * C * h * w;
    const int wi = k < N ? k / nW : 0, wj = k < N ? k - wi * nW : 0;
    const float* win = xb + (size_t)wi * w + wj;
    float* xTb = xT + (size_t)b * N * Cp;
    for (int r0 = 0; r0 < Cp; r0 += 32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rl = seg + 8 * i, r = r0 + rl;
            float v = 0.0f;
            if (r < K && k < N) {
                const int c = r / pp, d = r - c * pp, dy = d / patch, dx = d - dy * patch;
                v = win[((size_t)c * h + dy) * w + dx];
            }
            tile[rl][col] = v;
        }
        __syncthreads();
        const int rr = r0 + col;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kl = seg + 8 * i;
            if (rr < Cp && k0 + kl < N) xTb[(size_t)(k0 + kl) * Cp + rr] = tile[col][kl];
        }
        __syncthreads();
    }
}

int launch_window_prepare(const float* x, int B, int C, int h, int w, int patch, float* n1, float* inv, float* xT, int Cp, hipStream_t st)
{
    const int nW = w - patch + 1, Np = (h - patch + 1) * nW, N = h * w;
    chan_sumsq_kernel<<<B * cdiv(N, NCOL), NCOL * NSEG, 0, st>>>(x, C, N, n1);
    window_inv_kernel<<<dim3(cdiv(Np, 256), B), 256, 0, st>>>(n1, h, w, patch, nW, Np, inv);
    unfold_patchmajor_kernel<<<B * cdiv(Np, NCOL), NCOL * NSEG, 0, st>>>(x, C, h, w, patch, nW, Np, Cp, xT);
    return check_launch("unfold_patchmajor_kernel");
}


}  // namespace ipsr
