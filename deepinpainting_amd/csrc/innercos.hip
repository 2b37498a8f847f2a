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
constexpr int IC_MAX_BLOCKS = 4096;

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s);
    return v;
}

__global__ void __launch_bounds__(256) innercos_final_kernel(const double* __restrict__ partial, int nblocks, double inv_count,
                                                             float* __restrict__ loss)
{
    __shared__ double wsum[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) acc += partial[i];
    acc = wave_sum_f64(acc);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *loss = (float)(((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])) * inv_count);
}

// Round 4: a workgroup owns whole (sample, channel) rows — no 64-bit divisions per element, the mask float4 of a thread is loaded once.
// ticket == nullptr (the shipped path): block partials, folded by innercos_final_kernel.  ticket != nullptr (innercos_loss_fused): the
// LAST workgroup to finish (a word of caller memory, zero on entry, zero again on exit) folds them in the same launch.  Measured in the
// training step at [8,512,32,32] (HIP events around the launch): two launches 8.9 + 5.8 us; ONE launch 19.1 us with 512 workgroups and
// 70 us with 4096 — thousands of same-address atomics arrive at ~17 ns each, so the "last block" pattern that works for the norm
// kernels' 8 arrivals per counter does not scale to one counter per launch.  The fused entry point stays (tested), the default is two.
__global__ void __launch_bounds__(IC_THREADS) innercos_fused_kernel(const float* __restrict__ x, int Cx, int Cuse, int N, int rows, int rows_per_block,
                                                                    const float* __restrict__ mask, const float* __restrict__ target, float strength,
                                                                    double inv_count, double* __restrict__ partial, unsigned* __restrict__ ticket,
                                                                    float* __restrict__ loss)
{
    __shared__ double wsum[IC_THREADS / 64];
    __shared__ int last_s;
    double acc = 0.0;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    if ((N & 3) == 0) {
        const int n4 = N >> 2;
        for (int i4 = threadIdx.x; i4 < n4; i4 += IC_THREADS) {
            const float4 mv = *reinterpret_cast<const float4*>(mask + 4 * i4);
            for (int row = r0; row < r1; ++row) {
                const int b = row / Cuse, c = row - b * Cuse;                // uniform: scalar unit
                const float4 xv = *reinterpret_cast<const float4*>(x + ((size_t)b * Cx + c) * N + 4 * i4);
                const float4 tv = *reinterpret_cast<const float4*>(target + (size_t)row * N + 4 * i4);
                const float d0 = (xv.x * mv.x) * strength - tv.x;
                const float d1 = (xv.y * mv.y) * strength - tv.y;
                const float d2 = (xv.z * mv.z) * strength - tv.z;
                const float d3 = (xv.w * mv.w) * strength - tv.w;
                acc += (double)(d0 * d0);
                acc += (double)(d1 * d1);
                acc += (double)(d2 * d2);
                acc += (double)(d3 * d3);
            }
        }
    } else {
        for (int i = threadIdx.x; i < N; i += IC_THREADS) {
            const float mv = mask[i];
            for (int row = r0; row < r1; ++row) {
                const int b = row / Cuse, c = row - b * Cuse;
                const float d = (x[((size_t)b * Cx + c) * N + i] * mv) * strength - target[(size_t)row * N + i];
                acc += (double)(d * d);
            }
        }
    }
    acc = wave_sum_f64(acc);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < IC_THREADS / 64; ++i) t += wsum[i];
        last_s = 0;
        if (ticket) {
            __hip_atomic_store(partial + blockIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the write-through store has been acknowledged
            const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_s = old == gridDim.x - 1;
            if (old == gridDim.x - 1) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            partial[blockIdx.x] = t;
        }
    }
    __syncthreads();
    if (last_s && threadIdx.x < 64) {
        double t = 0.0;
        for (int i = threadIdx.x; i < (int)gridDim.x; i += 64) t += __hip_atomic_load(partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = wave_sum_f64(t);
        if (threadIdx.x == 0) *loss = (float)(t * inv_count);
    }
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

size_t innercos_ws_bytes(int B, int Cuse, int N)
{
    (void)B; (void)Cuse; (void)N;
    return (size_t)IC_MAX_BLOCKS * sizeof(double) + 256;
}

int launch_innercos_loss(const float* x, int B, int Cx, int Cuse, int N, const float* mask, const float* target,
                         float strength, float* loss, void* ws, size_t ws_bytes, hipStream_t st)
{
    if (ws_bytes < innercos_ws_bytes(B, Cuse, N)) return fail(IPSR_ERR_WORKSPACE, "innercos_loss: workspace %zu < %zu", ws_bytes, innercos_ws_bytes(B, Cuse, N));
    double* partial = reinterpret_cast<double*>(ws);
    const int rows = B * Cuse;
    const int rpb = (rows + IC_MAX_BLOCKS - 1) / IC_MAX_BLOCKS;           // a streaming kernel lives on waves in flight: up to 4096 workgroups
    const int nblk = (rows + rpb - 1) / rpb;
    profile_mark_start(st, 5);
    innercos_fused_kernel<<<nblk, IC_THREADS, 0, st>>>(x, Cx, Cuse, N, rows, rpb, mask, target, strength, 0.0, partial, nullptr, nullptr);
    profile_mark_stop(st, 5, 2.0 * rows * (double)N * 4.0, 2.0 * rows * (double)N * 4.0);
    if (int rc = check_launch("innercos_fused_kernel")) return rc;
    innercos_final_kernel<<<1, 256, 0, st>>>(partial, nblk, 1.0 / ((double)rows * N), loss);
    return check_launch("innercos_final_kernel");
}

// `ticket`: one 32-bit word of caller memory, zero on entry (left zero)
int launch_innercos_loss_fused(const float* x, int B, int Cx, int Cuse, int N, const float* mask, const float* target, float strength, float* loss,
                               void* ws, size_t ws_bytes, unsigned* ticket, hipStream_t st)
{
    if (ws_bytes < innercos_ws_bytes(B, Cuse, N)) return fail(IPSR_ERR_WORKSPACE, "innercos_loss: workspace %zu < %zu", ws_bytes, innercos_ws_bytes(B, Cuse, N));
    const int rows = B * Cuse;
    const int rpb = (rows + 511) / 512;                                   // few workgroups: every one of them is a same-address atomic
    const int nb = (rows + rpb - 1) / rpb;
    profile_mark_start(st, 5);
    innercos_fused_kernel<<<nb, IC_THREADS, 0, st>>>(x, Cx, Cuse, N, rows, rpb, mask, target, strength, 1.0 / ((double)rows * N),
                                                     reinterpret_cast<double*>(ws), ticket, loss);
    profile_mark_stop(st, 5, 2.0 * rows * (double)N * 4.0, 2.0 * rows * (double)N * 4.0);
    return check_launch("innercos_fused_kernel");
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
