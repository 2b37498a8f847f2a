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

// torch semantics: relu and max_pool2d propagate NaN (fmaxf would swallow it)
__device__ __forceinline__ float relu_nan(float v) { return v < 0.0f ? 0.0f : v; }
__device__ __forceinline__ float max_nan(float a, float b) { return (a > b || a != a) ? a : b; }

template <int ACT>   // 0 none, 1 relu, 2 leaky relu
__device__ __forceinline__ float act_apply(float v, float slope)
{
    if (ACT == 1) return v < 0.0f ? 0.0f : v;             // NaN stays NaN, as in torch
    if (ACT == 2) return v < 0.0f ? v * slope : v;
    return v;
}

// y2 (optional): relu(x + bias) as a second output — the skip half of the child level's concatenated tensor (see instnorm.hip's y2),
// channel slice given by its batch stride.
template <typename IO, int ACT>
__global__ void __launch_bounds__(256) bias_act_kernel(IO* __restrict__ x, const float* __restrict__ bias, int C, int HW, float slope,
                                                       IO* __restrict__ y2, size_t y2_bstride, unsigned* __restrict__ tickets)
{
    const int plane = blockIdx.y;                       // b*C + c
    // the per-channel counters of the matching backward's in-launch batch sum (instnorm.hip) start from zero
    if (tickets && blockIdx.x == 0 && threadIdx.x == 0 && plane < C) tickets[plane] = 0u;
    const float bv = bias ? bias[plane % C] : 0.0f;
    IO* p = x + (size_t)plane * HW;
    IO* q = y2 ? y2 + (size_t)(plane / C) * y2_bstride + (size_t)(plane % C) * HW : nullptr;
    if ((HW & 3) == 0) {
        const int n4 = HW >> 2;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
            float4 v = ld4(p, i);
            v.x += bv; v.y += bv; v.z += bv; v.w += bv;
            if (q) st4(q, i, make_float4(relu_nan(v.x), relu_nan(v.y), relu_nan(v.z), relu_nan(v.w)));
            v.x = act_apply<ACT>(v.x, slope); v.y = act_apply<ACT>(v.y, slope);
            v.z = act_apply<ACT>(v.z, slope); v.w = act_apply<ACT>(v.w, slope);
            st4(p, i, v);
        }
    } else {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
            const float v = ld1(p, i) + bv;
            if (q) st1(q, i, relu_nan(v));
            st1(p, i, act_apply<ACT>(v, slope));
        }
    }
}

// one thread -> two horizontally adjacent pooled outputs (reads two float4-aligned row segments when W % 4 == 0)
template <typename IO>
__global__ void __launch_bounds__(256) bias_relu_pool2_kernel(const IO* __restrict__ x, const float* __restrict__ bias, int C, int H, int W,
                                                              IO* __restrict__ y)
{
    const int plane = blockIdx.y;
    const float bv = bias ? bias[plane % C] : 0.0f;
    const int Ho = H >> 1, Wo = W >> 1;
    const IO* p = x + (size_t)plane * H * W;
    IO* q = y + (size_t)plane * Ho * Wo;
    if ((W & 3) == 0) {
        const int Wp = Wo >> 1;                          // output pairs per row
        const int n = Ho * Wp;
        for (int t = blockIdx.x * 256 + threadIdx.x; t < n; t += gridDim.x * 256) {
            const int i = t / Wp, jp = t - i * Wp;
            const float4 a = ld4(p + (size_t)(2 * i) * W, jp);
            const float4 b = ld4(p + (size_t)(2 * i + 1) * W, jp);
            // relu(v + bias) per element, then the window max in torch's order (row-major over the window)
            const float a0 = relu_nan(a.x + bv), a1 = relu_nan(a.y + bv), a2 = relu_nan(a.z + bv), a3 = relu_nan(a.w + bv);
            const float b0 = relu_nan(b.x + bv), b1 = relu_nan(b.y + bv), b2 = relu_nan(b.z + bv), b3 = relu_nan(b.w + bv);
            st1(q, (size_t)i * Wo + 2 * jp, max_nan(max_nan(a0, a1), max_nan(b0, b1)));
            st1(q, (size_t)i * Wo + 2 * jp + 1, max_nan(max_nan(a2, a3), max_nan(b2, b3)));
        }
    } else {
        const int n = Ho * Wo;
        for (int t = blockIdx.x * 256 + threadIdx.x; t < n; t += gridDim.x * 256) {
            const int i = t / Wo, j = t - i * Wo;
            const size_t r0 = (size_t)(2 * i) * W + 2 * j, r1 = r0 + W;
            const float v0 = relu_nan(ld1(p, r0) + bv), v1 = relu_nan(ld1(p, r0 + 1) + bv);
            const float v2 = relu_nan(ld1(p, r1) + bv), v3 = relu_nan(ld1(p, r1 + 1) + bv);
            st1(q, t, max_nan(max_nan(v0, v1), max_nan(v2, v3)));
        }
    }
}

// torch.cat([y, x], 1) followed by the parent level's in-place ReLU (models/networks.py:270-278 + :229 `uprelu`): one pass
//     out[b, :C1] = relu(y[b]),  out[b, C1:] = relu(x[b])
// and its backward (threshold_backward + the two channel slices, written as two contiguous tensors):
//     dy = g[b, :C1] * (out[b, :C1] > 0),  dx = g[b, C1:] * (out[b, C1:] > 0)
template <typename IO>
__global__ void __launch_bounds__(256) cat_relu_fwd_kernel(const IO* __restrict__ y, const IO* __restrict__ x, int C1, int C2, int HW,
                                                           IO* __restrict__ out)
{
    const int b = blockIdx.y;
    const size_t n1 = (size_t)C1 * HW, n2 = (size_t)C2 * HW, n = n1 + n2;
    const IO* yb = y + (size_t)b * n1;
    const IO* xb = x + (size_t)b * n2;
    IO* ob = out + (size_t)b * n;
    if ((HW & 3) == 0) {
        const size_t n4 = n >> 2, n14 = n1 >> 2;
        for (size_t i = (y ? 0 : n14) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {      // y == NULL: its half is already there
            float4 v = i < n14 ? ld4(yb, i) : ld4(xb, i - n14);
            v.x = relu_nan(v.x); v.y = relu_nan(v.y); v.z = relu_nan(v.z); v.w = relu_nan(v.w);
            st4(ob, i, v);
        }
    } else {
        for (size_t i = (y ? 0 : n1) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
            st1(ob, i, relu_nan(i < n1 ? ld1(yb, i) : ld1(xb, i - n1)));
    }
}

template <typename IO>
__global__ void __launch_bounds__(256) cat_relu_bwd_kernel(const IO* __restrict__ g, const IO* __restrict__ out, int C1, int C2, int HW,
                                                           IO* __restrict__ dy, IO* __restrict__ dx)
{
    const int b = blockIdx.y;
    const size_t n1 = (size_t)C1 * HW, n2 = (size_t)C2 * HW, n = n1 + n2;
    const IO* gb = g + (size_t)b * n;
    const IO* ob = out + (size_t)b * n;
    IO* dyb = dy + (size_t)b * n1;
    IO* dxb = dx + (size_t)b * n2;
    if ((HW & 3) == 0) {
        const size_t n4 = n >> 2, n14 = n1 >> 2;
        for (size_t i = (dy ? 0 : n14) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {     // dy == NULL: only the skip half is wanted
            const float4 gv = ld4(gb, i), ov = ld4(ob, i);
            float4 r;
            r.x = ov.x > 0.0f ? gv.x : 0.0f; r.y = ov.y > 0.0f ? gv.y : 0.0f;
            r.z = ov.z > 0.0f ? gv.z : 0.0f; r.w = ov.w > 0.0f ? gv.w : 0.0f;
            if (i < n14) st4(dyb, i, r); else st4(dxb, i - n14, r);
        }
    } else {
        for (size_t i = (dy ? 0 : n1) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
            const float r = ld1(ob, i) > 0.0f ? ld1(gb, i) : 0.0f;
            if (i < n1) st1(dyb, i, r); else st1(dxb, i - n1, r);
        }
    }
}

int launch_cat_relu_fwd(const void* y, const void* x, int B, int C1, int C2, int HW, int io_bf16, void* out, hipStream_t st)
{
    if (B > 65535) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_cat_relu: batch %d > 65535", B);
    const size_t n = (size_t)(C1 + C2) * HW;
    int gx = (int)((((HW & 3) == 0 ? n >> 2 : n) + 256 * 4 - 1) / (256 * 4));
    if (gx < 1) gx = 1;
    if (io_bf16) cat_relu_fwd_kernel<bf16_t><<<dim3(gx, B), 256, 0, st>>>(static_cast<const bf16_t*>(y), static_cast<const bf16_t*>(x), C1, C2, HW, static_cast<bf16_t*>(out));
    else cat_relu_fwd_kernel<float><<<dim3(gx, B), 256, 0, st>>>(static_cast<const float*>(y), static_cast<const float*>(x), C1, C2, HW, static_cast<float*>(out));
    return check_launch("cat_relu_fwd_kernel");
}

int launch_cat_relu_bwd(const void* g, const void* out, int B, int C1, int C2, int HW, int io_bf16, void* dy, void* dx, hipStream_t st)
{
    if (B > 65535) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_cat_relu: batch %d > 65535", B);
    const size_t n = (size_t)(C1 + C2) * HW;
    int gx = (int)((((HW & 3) == 0 ? n >> 2 : n) + 256 * 4 - 1) / (256 * 4));
    if (gx < 1) gx = 1;
    if (io_bf16) cat_relu_bwd_kernel<bf16_t><<<dim3(gx, B), 256, 0, st>>>(static_cast<const bf16_t*>(g), static_cast<const bf16_t*>(out), C1, C2, HW, static_cast<bf16_t*>(dy), static_cast<bf16_t*>(dx));
    else cat_relu_bwd_kernel<float><<<dim3(gx, B), 256, 0, st>>>(static_cast<const float*>(g), static_cast<const float*>(out), C1, C2, HW, static_cast<float*>(dy), static_cast<float*>(dx));
    return check_launch("cat_relu_bwd_kernel");
}

int launch_bias_act(void* x, const float* bias, int B, int C, int HW, int act, float slope, int io_bf16, void* y2, size_t y2bs, unsigned* tickets,
                    hipStream_t st)
{
    const int planes = B * C;
    if (planes > 65535) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_bias_act: B*C=%d > 65535 planes", planes);
    const int per = (HW & 3) == 0 ? HW >> 2 : HW;
    int gx = cdiv(per, 256 * 4);                        // ~4 vectors per thread
    if (gx < 1) gx = 1;
    const dim3 grid(gx, planes);
    if (act < 0 || act > 2) return fail(IPSR_ERR_INVALID, "ipsr_bias_act: unknown activation %d", act);
#define BA_LAUNCH(IO)                                                                                         \
    switch (act) {                                                                                            \
        case 0: bias_act_kernel<IO, 0><<<grid, 256, 0, st>>>(static_cast<IO*>(x), bias, C, HW, slope, static_cast<IO*>(y2), y2bs, tickets); break; \
        case 1: bias_act_kernel<IO, 1><<<grid, 256, 0, st>>>(static_cast<IO*>(x), bias, C, HW, slope, static_cast<IO*>(y2), y2bs, tickets); break; \
        default: bias_act_kernel<IO, 2><<<grid, 256, 0, st>>>(static_cast<IO*>(x), bias, C, HW, slope, static_cast<IO*>(y2), y2bs, tickets); break; \
    }
    if (io_bf16) { BA_LAUNCH(bf16_t) } else { BA_LAUNCH(float) }
#undef BA_LAUNCH
    return check_launch("bias_act_kernel");
}

int launch_bias_relu_pool2(const void* x, const float* bias, int B, int C, int H, int W, int io_bf16, void* y, hipStream_t st)
{
    const int planes = B * C;
    if (planes > 65535) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_bias_relu_pool2: B*C=%d > 65535 planes", planes);
    const int Ho = H >> 1, Wo = W >> 1;
    const int per = (W & 3) == 0 ? Ho * (Wo >> 1) : Ho * Wo;
    int gx = cdiv(per, 256 * 2);
    if (gx < 1) gx = 1;
    if (io_bf16)
        bias_relu_pool2_kernel<bf16_t><<<dim3(gx, planes), 256, 0, st>>>(static_cast<const bf16_t*>(x), bias, C, H, W, static_cast<bf16_t*>(y));
    else
        bias_relu_pool2_kernel<float><<<dim3(gx, planes), 256, 0, st>>>(static_cast<const float*>(x), bias, C, H, W, static_cast<float*>(y));
    return check_launch("bias_relu_pool2_kernel");
}

}  // namespace ipsr
