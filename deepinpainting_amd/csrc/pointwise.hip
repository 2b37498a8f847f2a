// pointwise.hip — the bias / activation / pooling passes between the convolutions of the frozen VGG16 feature net.
//
// Reference: models/vgg16.py:6-37 wraps torchvision's vgg16.features: Conv2d(bias) -> ReLU(inplace) [-> MaxPool2d(2,2)].
// The training step runs that net three times per step without gradients (models/IPSR.py:163,187,212-213).  MIOpen
// convolutions carry no bias, so PyTorch issues a broadcast add, then the ReLU, then the pool: three full passes over
// activations of up to 134 MB.  Here the convolution is called without bias and ONE pass does the rest:
//     bias_act_kernel           x[b,c,:] = act(x[b,c,:] + bias[c])                 in place   (R + W)
//     bias_relu_pool2_kernel    y[b,c,i,j] = max_{2x2} relu(x[b,c,2i+di,2j+dj] + bias[c])    (R + W/4)
// Same arithmetic as the three torch kernels (one add, one max with 0, max over the window): bit-identical outputs.
// HBM-bound; 16-byte accesses, one (b,c) plane per blockIdx.y so the bias is a scalar.
#include "ipsr_common.h"

namespace ipsr {

template <int ACT>   // 0 none, 1 relu, 2 leaky relu
__device__ __forceinline__ float act_apply(float v, float slope)
{
    if (ACT == 1) return v > 0.0f ? v : 0.0f;
    if (ACT == 2) return v > 0.0f ? v : v * slope;
    return v;
}

template <int ACT>
__global__ void __launch_bounds__(256) bias_act_kernel(float* __restrict__ x, const float* __restrict__ bias, int C, int HW, float slope)
{
    const int plane = blockIdx.y;                       // b*C + c
    const float bv = bias ? bias[plane % C] : 0.0f;
    float* p = x + (size_t)plane * HW;
    if ((HW & 3) == 0) {
        const int n4 = HW >> 2;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
            float4 v = reinterpret_cast<float4*>(p)[i];
            v.x = act_apply<ACT>(v.x + bv, slope); v.y = act_apply<ACT>(v.y + bv, slope);
            v.z = act_apply<ACT>(v.z + bv, slope); v.w = act_apply<ACT>(v.w + bv, slope);
            reinterpret_cast<float4*>(p)[i] = v;
        }
    } else {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) p[i] = act_apply<ACT>(p[i] + bv, slope);
    }
}

// one thread -> two horizontally adjacent pooled outputs (reads two float4-aligned row segments when W % 4 == 0)
__global__ void __launch_bounds__(256) bias_relu_pool2_kernel(const float* __restrict__ x, const float* __restrict__ bias, int C, int H, int W,
                                                              float* __restrict__ y)
{
    const int plane = blockIdx.y;
    const float bv = bias ? bias[plane % C] : 0.0f;
    const int Ho = H >> 1, Wo = W >> 1;
    const float* p = x + (size_t)plane * H * W;
    float* q = y + (size_t)plane * Ho * Wo;
    if ((W & 3) == 0) {
        const int Wp = Wo >> 1;                          // output pairs per row
        const int n = Ho * Wp;
        for (int t = blockIdx.x * 256 + threadIdx.x; t < n; t += gridDim.x * 256) {
            const int i = t / Wp, jp = t - i * Wp;
            const float4 a = *reinterpret_cast<const float4*>(p + (size_t)(2 * i) * W + 4 * jp);
            const float4 b = *reinterpret_cast<const float4*>(p + (size_t)(2 * i + 1) * W + 4 * jp);
            // relu(v + bias) per element, then the window max in torch's order (row-major over the window)
            const float a0 = fmaxf(a.x + bv, 0.0f), a1 = fmaxf(a.y + bv, 0.0f), a2 = fmaxf(a.z + bv, 0.0f), a3 = fmaxf(a.w + bv, 0.0f);
            const float b0 = fmaxf(b.x + bv, 0.0f), b1 = fmaxf(b.y + bv, 0.0f), b2 = fmaxf(b.z + bv, 0.0f), b3 = fmaxf(b.w + bv, 0.0f);
            float2 o;
            o.x = fmaxf(fmaxf(a0, a1), fmaxf(b0, b1));
            o.y = fmaxf(fmaxf(a2, a3), fmaxf(b2, b3));
            *reinterpret_cast<float2*>(q + (size_t)i * Wo + 2 * jp) = o;
        }
    } else {
        const int n = Ho * Wo;
        for (int t = blockIdx.x * 256 + threadIdx.x; t < n; t += gridDim.x * 256) {
            const int i = t / Wo, j = t - i * Wo;
            const float* r0 = p + (size_t)(2 * i) * W + 2 * j;
            const float* r1 = r0 + W;
            const float v0 = fmaxf(r0[0] + bv, 0.0f), v1 = fmaxf(r0[1] + bv, 0.0f);
            const float v2 = fmaxf(r1[0] + bv, 0.0f), v3 = fmaxf(r1[1] + bv, 0.0f);
            q[t] = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
        }
    }
}

int launch_bias_act(float* x, const float* bias, int B, int C, int HW, int act, float slope, hipStream_t st)
{
    const int planes = B * C;
    if (planes > 65535) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_bias_act: B*C=%d > 65535 planes", planes);
    const int per = (HW & 3) == 0 ? HW >> 2 : HW;
    int gx = cdiv(per, 256 * 4);                        // ~4 vectors per thread
    if (gx < 1) gx = 1;
    const dim3 grid(gx, planes);
    switch (act) {
        case 0: bias_act_kernel<0><<<grid, 256, 0, st>>>(x, bias, C, HW, slope); break;
        case 1: bias_act_kernel<1><<<grid, 256, 0, st>>>(x, bias, C, HW, slope); break;
        case 2: bias_act_kernel<2><<<grid, 256, 0, st>>>(x, bias, C, HW, slope); break;
        default: return fail(IPSR_ERR_INVALID, "ipsr_bias_act: unknown activation %d", act);
    }
    return check_launch("bias_act_kernel");
}

int launch_bias_relu_pool2(const float* x, const float* bias, int B, int C, int H, int W, float* y, hipStream_t st)
{
    const int planes = B * C;
    if (planes > 65535) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_bias_relu_pool2: B*C=%d > 65535 planes", planes);
    const int Ho = H >> 1, Wo = W >> 1;
    const int per = (W & 3) == 0 ? Ho * (Wo >> 1) : Ho * Wo;
    int gx = cdiv(per, 256 * 2);
    if (gx < 1) gx = 1;
    bias_relu_pool2_kernel<<<dim3(gx, planes), 256, 0, st>>>(x, bias, C, H, W, y);
    return check_launch("bias_relu_pool2_kernel");
}

}  // namespace ipsr
