// innercos.hip — K9: InnerCos / InnerCos2 feature-consistency loss.
//
// Reference: InnerCos.forward (models/InnerCos.py:30-41) and InnerCos2.forward (models/InnerCos2.py:34-46):
//     loss = MSELoss( (x * mask) * strength , target )      mean over B*Cuse*N elements
// with x [B,Cx,N] (InnerCos2 reads the first Cuse=512 of Cx=1024 channels, :38), mask [N] fp32 0/1 and
// target [B,Cuse,N] = VGG relu4_3 of the ground truth.  The reference makes three element-wise passes
// plus a reduction; here it is ONE pass: 2 streamed reads (x, target), per-thread fp32 terms, fp64
// accumulation (deterministic: fixed grid, fixed tree, a second tiny kernel folds the block partials).
// HBM-bound: 2*B*Cuse*N*4 bytes.
#include "ipsr_common.h"

namespace ipsr {

constexpr int IC_THREADS = 256;
constexpr int IC_MAX_BLOCKS = 1024;

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s);
    return v;
}

__global__ void __launch_bounds__(IC_THREADS) innercos_partial_kernel(const float* __restrict__ x, int Cx, int Cuse, int N,
                                                                      const float* __restrict__ mask,
                                                                      const float* __restrict__ target, float strength,
                                                                      size_t total, double* __restrict__ partial)
{
    __shared__ double wsum[IC_THREADS / 64];
    double acc = 0.0;
    const size_t stride = (size_t)gridDim.x * IC_THREADS;
    const size_t per_sample = (size_t)Cuse * N;
    if ((N & 3) == 0) {
        // 16 B per lane: N % 4 == 0 keeps a float4 inside one (b,c) row and mask-aligned
        const size_t total4 = total >> 2;
        for (size_t i4 = (size_t)blockIdx.x * IC_THREADS + threadIdx.x; i4 < total4; i4 += stride) {
            const size_t i = i4 << 2;
            const size_t b = i / per_sample, rem = i - b * per_sample;
            const int n = (int)(rem % N);
            const float4 xv = *reinterpret_cast<const float4*>(x + b * (size_t)Cx * N + rem);
            const float4 tv = *reinterpret_cast<const float4*>(target + i);
            const float4 mv = *reinterpret_cast<const float4*>(mask + n);
            const float d0 = (xv.x * mv.x) * strength - tv.x;
            const float d1 = (xv.y * mv.y) * strength - tv.y;
            const float d2 = (xv.z * mv.z) * strength - tv.z;
            const float d3 = (xv.w * mv.w) * strength - tv.w;
            acc += (double)(d0 * d0);
            acc += (double)(d1 * d1);
            acc += (double)(d2 * d2);
            acc += (double)(d3 * d3);
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * IC_THREADS + threadIdx.x; i < total; i += stride) {
            const size_t b = i / per_sample, rem = i - b * per_sample;
            const int n = (int)(rem % N);
            const float d = (x[b * (size_t)Cx * N + rem] * mask[n]) * strength - target[i];
            acc += (double)(d * d);
        }
    }
    acc = wave_sum_f64(acc);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < IC_THREADS / 64; ++i) t += wsum[i];
        partial[blockIdx.x] = t;
    }
}

__global__ void __launch_bounds__(64) innercos_final_kernel(const double* __restrict__ partial, int nblocks, double inv_count,
                                                            float* __restrict__ loss)
{
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 64) acc += partial[i];
    acc = wave_sum_f64(acc);
    if (threadIdx.x == 0) *loss = (float)(acc * inv_count);
}

// d loss / d x = grad_loss * 2/(B*Cuse*N) * ((x*m)*s - t) * (m*s) on the first Cuse channels, 0 elsewhere.
__global__ void __launch_bounds__(256) innercos_backward_kernel(const float* __restrict__ x, int Cx, int Cuse, int N,
                                                                const float* __restrict__ mask, const float* __restrict__ target,
                                                                float strength, const float* __restrict__ grad_loss,
                                                                float two_over_count, size_t total_x, float* __restrict__ grad_x)
{
    const float scale = (*grad_loss) * two_over_count;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t per_x = (size_t)Cx * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_x; i += stride) {
        const size_t b = i / per_x, rem = i - b * per_x;
        const int c = (int)(rem / N), n = (int)(rem - (size_t)c * N);
        float gv = 0.0f;
        if (c < Cuse) {
            const float m = mask[n];
            const float d = (x[i] * m) * strength - target[(b * Cuse + c) * (size_t)N + n];
            gv = (d * scale) * (m * strength);
        }
        grad_x[i] = gv;
    }
}

static int ic_blocks(size_t total)
{
    size_t need = (total / 4 + IC_THREADS - 1) / IC_THREADS;
    if (need < 1) need = 1;
    return (int)(need > IC_MAX_BLOCKS ? IC_MAX_BLOCKS : need);
}

size_t innercos_ws_bytes(int B, int Cuse, int N)
{
    (void)B; (void)Cuse; (void)N;
    return (size_t)IC_MAX_BLOCKS * sizeof(double) + 256;
}

int launch_innercos_loss(const float* x, int B, int Cx, int Cuse, int N, const float* mask, const float* target,
                         float strength, float* loss, void* ws, size_t ws_bytes, hipStream_t st)
{
    if (ws_bytes < innercos_ws_bytes(B, Cuse, N)) return fail(IPSR_ERR_WORKSPACE, "innercos_loss: workspace %zu < %zu", ws_bytes, innercos_ws_bytes(B, Cuse, N));
    const size_t total = (size_t)B * Cuse * N;
    const int nb = ic_blocks(total);
    double* partial = reinterpret_cast<double*>(ws);
    innercos_partial_kernel<<<nb, IC_THREADS, 0, st>>>(x, Cx, Cuse, N, mask, target, strength, total, partial);
    if (int rc = check_launch("innercos_partial_kernel")) return rc;
    innercos_final_kernel<<<1, 64, 0, st>>>(partial, nb, 1.0 / (double)total, loss);
    return check_launch("innercos_final_kernel");
}

int launch_innercos_backward(const float* x, int B, int Cx, int Cuse, int N, const float* mask,
                             const float* target, float strength, const float* grad_loss, float* grad_x,
                             hipStream_t st)
{
    const size_t total_x = (size_t)B * Cx * N;
    size_t nb = (total_x + 255) / 256;
    if (nb > 2048) nb = 2048;
    const float two_over_count = 2.0f / (float)((double)B * Cuse * N);
    innercos_backward_kernel<<<(int)nb, 256, 0, st>>>(x, Cx, Cuse, N, mask, target, strength, grad_loss, two_over_count, total_x, grad_x);
    return check_launch("innercos_backward_kernel");
}

}  // namespace ipsr
