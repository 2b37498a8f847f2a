// unfold.hip — shift_sz > 1: patch unfold in front of, and overlap-add fold behind, the patch-size-1 pipeline.
//
// Reference: NonparametricShift._extract_patches (util/NonparametricShift.py:59-73) cuts p x p windows (stride 1, no
// padding) out of the [C,h,w] feature: N' = (h-p+1)(w-p+1) patches of K = C*p*p numbers.  Everything between that and
// the final ConvTranspose2d (models/IPSRFunction.py:130) treats a patch as one flat K-vector, so the layer for p > 1
// is   unfold -> (normalise, correlate, arg-max, recurrence, reconstruct on a [K, N'] matrix) -> fold.
// Both kernels are pure data movement (HBM-bound; the unfolded matrices are p*p times the feature and stay far below
// the cost of the 2*N'^2*K correlation they feed).
//
//   unfold:  xu[b][(c*p+dy)*p+dx][i*nW+j] = x[b][c][i+dy][j+dx];  row stride ld >= N', pad columns written as zeros
//            (the correlation kernel's fast path reads whole 128-column tiles)
//   fold:    out[b][c][y][x] = sum over (dy,dx) ascending, window (y-dy, x-dx) inside the grid, of yu[b][(c,dy,dx)][window]
//            — each add rounded, fixed order (oracle: ipsr_fold_cpu); with `addend` the result is addend + that sum
//            (backward: grad_in = g + fold(...)).
#include "ipsr_common.h"

namespace ipsr {

// one thread -> 4 consecutive columns of one row: four gathered loads, one 16-byte store (ld % 4 == 0 or scalar tail)
__global__ void __launch_bounds__(256) unfold_kernel(const float* __restrict__ x, int C, int h, int w, int patch, int nW,
                                                     int Np, int ld, float* __restrict__ xu)
{
    const int col0 = (blockIdx.x * 256 + threadIdx.x) * 4;   // first window index (or pad column) of this thread
    const int k = blockIdx.y;                                // row of the unfolded matrix
    const int b = blockIdx.z;
    if (col0 >= ld) return;
    const int pp = patch * patch;
    const int c = k / pp, d = k - c * pp, dy = d / patch, dx = d - dy * patch;
    const float* src = x + (((size_t)b * C + c) * h + dy) * w + dx;
    float v[4];
    int i = col0 / nW, j = col0 - i * nW;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[e] = (col0 + e < Np) ? src[(size_t)i * w + j] : 0.0f;
        if (++j == nW) { j = 0; ++i; }
    }
    float* dst = xu + ((size_t)b * C * pp + k) * ld + col0;
    if ((ld & 3) == 0) {
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (col0 + e < ld) dst[e] = v[e];
    }
}

__global__ void __launch_bounds__(256) fold_kernel(const float* __restrict__ yu, int C, int h, int w, int patch, int nH, int nW,
                                                   const float* __restrict__ addend, float* __restrict__ out)
{
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (pix >= h * w) return;
    const int y = pix / w, xx = pix - y * w;
    const int pp = patch * patch, Np = nH * nW;
    const float* yb = yu + ((size_t)b * C + c) * pp * Np;
    float acc = 0.0f;
    for (int dy = 0; dy < patch; ++dy) {
        const int i = y - dy;
        if (i < 0 || i >= nH) continue;
        for (int dx = 0; dx < patch; ++dx) {
            const int j = xx - dx;
            if (j < 0 || j >= nW) continue;
            acc = acc + yb[(size_t)(dy * patch + dx) * Np + (size_t)i * nW + j];
        }
    }
    const size_t o = ((size_t)b * C + c) * h * w + pix;
    out[o] = addend ? addend[o] + acc : acc;
}

int launch_unfold(const float* x, int B, int C, int h, int w, int patch, int ld, float* xu, hipStream_t st)
{
    const int nH = h - patch + 1, nW = w - patch + 1, Np = nH * nW, K = C * patch * patch;
    if (K > 65535 || B > 65535) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_forward: C*p*p=%d exceeds the unfold grid", K);
    unfold_kernel<<<dim3(cdiv(ld, 1024), K, B), 256, 0, st>>>(x, C, h, w, patch, nW, Np, ld, xu);
    return check_launch("unfold_kernel");
}

int launch_fold(const float* yu, int B, int C, int h, int w, int patch, float* out, hipStream_t st, const float* addend)
{
    const int nH = h - patch + 1, nW = w - patch + 1;
    if (C > 65535 || B > 65535) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_forward: C=%d exceeds the fold grid", C);
    fold_kernel<<<dim3(cdiv(h * w, 256), C, B), 256, 0, st>>>(yu, C, h, w, patch, nH, nW, addend, out);
    return check_launch("fold_kernel");
}

}  // namespace ipsr
