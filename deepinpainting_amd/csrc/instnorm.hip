// instnorm.hip — conv-bias + InstanceNorm2d(affine) + LeakyReLU/ReLU in one pass, forward and backward.
//
// Reference: every U-Net level and discriminator stage is  Conv2d/ConvTranspose2d(bias) -> InstanceNorm2d -> (the next
// level's) LeakyReLU(0.2, inplace) or ReLU(inplace)  (models/networks.py:220-259, 404-432, 470-497, 507-514).  MIOpen
// convolutions carry no bias, so PyTorch runs a broadcast add, the norm (2-3 passes) and the activation as separate
// kernels — 4 reads + 3 writes of the activation forward, 6 reads + 2 writes backward (+ a bias-gradient reduction).
// Here one workgroup owns one (sample, channel) plane, keeps it in REGISTERS between the statistics and the output:
//     forward   z = x + bias[c];  mean, var over the plane;  y = act((z - mean) * rstd * gamma[c] + beta[c])      1 R + 1 W
//     backward  dz = dy * act'(y);  xh = (x + bias - mean) * rstd;  s1 = sum dz, s2 = sum dz*xh;
//               dx = gamma * rstd * (dz - s1/HW - xh * s2/HW);                                                    3 R + 1 W
//               per-plane partials: dgamma += s2, dbeta += s1, dbias += sum dx
//               batch sums: the LAST of a channel's B planes to finish (a per-channel ticket) adds the B partials in
//               fixed order b = 0..B-1 -> deterministic, and no separate reduction launch per layer
// HBM-bound.  Planes up to 128x128 (16384 elements) live in the registers of 256 threads; the one larger normalised map of
// the reference's nets at 256x256 (netG's outermost level, 64 x 256 x 256) takes 512 threads x 128 elements per plane.
// fp32 tolerance vs torch (different summation order): ~1e-6 relative, asserted in tests/test_gpu_model.py.
#include "ipsr_common.h"

namespace ipsr {

template <int T>
__device__ __forceinline__ float block_sum(float v, float* red)
{
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s);
    if (T == 64) return v;
    const int wv = threadIdx.x >> 6;
    __syncthreads();                       // protect `red` from the previous use
    if ((threadIdx.x & 63) == 0) red[wv] = v;
    __syncthreads();
    float t = red[0];
#pragma unroll
    for (int i = 1; i < T / 64; ++i) t += red[i];
    return t;
}

// Per-channel tickets for the batch reduction: C words of CALLER memory (the library keeps no device state).  They must be zero when
// a backward launch starts — the matching FORWARD entry point zeroes them (plane b = 0 of every channel; the forward always precedes its
// backward on the stream) — and the plane that draws ticket B-1 puts the zero back, so a second backward over the same node
// (retain_graph) finds them ready.  Every node of the graph owns its words: launches that overlap on other streams cannot alias, and
// a launch that died leaves nothing behind for anybody else.

// Store a plane's partial so that another XCD can read it: a device-scope (write-through) store — NOT a device-scope release
// fence, which on this chip writes back the whole L2 (measured: +100 us per launch when every plane's workgroup did one).
__device__ __forceinline__ void st_partial(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Called by every thread of a plane's workgroup after thread 0 stored the plane's partials part[k][b][c] with st_partial (k < nk
// arrays at stride B*C).  The workgroup that arrives last for channel c writes sums[k][c] = sum_b part[k][b][c].
// Ordering: thread 0 waits for its write-through stores to be acknowledged (vmcnt(0)) before it draws the ticket, so whoever
// draws B-1 finds every partial at the device-coherent level; it reads them with device-scope loads (no stale L2 / L1 line).
__device__ __forceinline__ void batch_sum_by_last_plane(float* const* part, int nk, float* __restrict__ sums, unsigned* ticket, int B,
                                                        int C, int c, int* last_s)
{
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the write-through partial stores have been acknowledged
        const unsigned old = __hip_atomic_fetch_add(ticket + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *last_s = old == (unsigned)(B - 1);
        if (old == (unsigned)(B - 1)) __hip_atomic_store(ticket + c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (*last_s && (int)threadIdx.x < nk) {
        const float* p = part[threadIdx.x];
        if (p) {
            float t = 0.0f;
            for (int b = 0; b < B; ++b) t += __hip_atomic_load(p + (size_t)b * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sums[(size_t)threadIdx.x * C + c] = t;
        }
    }
}

__device__ __forceinline__ float act_fwd(float v, int act, float slope)
{
    // NaN-propagating like torch's relu / leaky_relu (a diverged net must show, not read as 0)
    if (act == 1) return v < 0.0f ? 0.0f : v;
    if (act == 2) return v < 0.0f ? v * slope : v;
    return v;
}
// derivative from the OUTPUT (slope > 0 keeps the sign, ReLU maps negatives to 0)
__device__ __forceinline__ float act_bwd(float y, int act, float slope)
{
    if (act == 1) return y > 0.0f ? 1.0f : 0.0f;
    if (act == 2) return y > 0.0f ? 1.0f : slope;
    return 1.0f;
}

// NE = plane elements a thread may hold; VEC: 16-byte accesses (HW % 4 == 0)
template <typename IO, int T, int NE, bool VEC>
__global__ void __launch_bounds__(T) instnorm_act_fwd_kernel(const IO* __restrict__ x, const float* __restrict__ bias,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float eps, int act, float slope, int C, int HW,
                                                             IO* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                             size_t y_bstride, IO* __restrict__ y2, size_t y2_bstride, unsigned* __restrict__ tickets)
{
    __shared__ float red[16];
    const int plane = blockIdx.x, c = plane % C, tid = threadIdx.x;
    if (tickets && tid == 0 && plane < C) tickets[plane] = 0u;          // the backward's per-channel counters start from zero
    const float bv = bias ? bias[c] : 0.0f;
    const IO* xp = x + (size_t)plane * HW;
    IO* yp = y + (size_t)(plane / C) * y_bstride + (size_t)c * HW;      // y may be a channel slice of a wider tensor (skip concatenation)
    // y2 (optional): relu of the same normalised value, written into the skip half of the child level's concatenated tensor
    IO* y2p = y2 ? y2 + (size_t)(plane / C) * y2_bstride + (size_t)c * HW : nullptr;
    float v[NE];
    float sum = 0.0f;
    if (VEC) {
        const int n4 = HW >> 2;
#pragma unroll
        for (int k = 0; k < NE / 4; ++k) {
            const int i = k * T + tid;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < n4) {
                t = ld4(xp, i);
                t.x += bv; t.y += bv; t.z += bv; t.w += bv;
                sum += (t.x + t.y) + (t.z + t.w);
            }
            v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int i = k * T + tid;
            v[k] = i < HW ? ld1(xp, i) + bv : 0.0f;
            if (i < HW) sum += v[k];
        }
    }
    const float inv_n = 1.0f / (float)HW;
    const float mean = block_sum<T>(sum, red) * inv_n;
    float ss = 0.0f;
    if (VEC) {
        const int n4 = HW >> 2;
#pragma unroll
        for (int k = 0; k < NE / 4; ++k)
            if (k * T + tid < n4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[4 * k + e] - mean; ss = __builtin_fmaf(d, d, ss); }
            }
    } else {
#pragma unroll
        for (int k = 0; k < NE; ++k)
            if (k * T + tid < HW) { const float d = v[k] - mean; ss = __builtin_fmaf(d, d, ss); }
    }
    const float var = block_sum<T>(ss, red) * inv_n;
    const float rstd = 1.0f / sqrtf(var + eps);
    const float g = (gamma ? gamma[c] : 1.0f) * rstd, bt = beta ? beta[c] : 0.0f;
    if (VEC) {
        const int n4 = HW >> 2;
#pragma unroll
        for (int k = 0; k < NE / 4; ++k) {
            const int i = k * T + tid;
            if (i < n4) {
                float4 n, o;
                n.x = (v[4 * k] - mean) * g + bt; n.y = (v[4 * k + 1] - mean) * g + bt;
                n.z = (v[4 * k + 2] - mean) * g + bt; n.w = (v[4 * k + 3] - mean) * g + bt;
                o.x = act_fwd(n.x, act, slope); o.y = act_fwd(n.y, act, slope);
                o.z = act_fwd(n.z, act, slope); o.w = act_fwd(n.w, act, slope);
                st4(yp, i, o);
                if (y2p) st4(y2p, i, make_float4(act_fwd(n.x, 1, 0.f), act_fwd(n.y, 1, 0.f), act_fwd(n.z, 1, 0.f), act_fwd(n.w, 1, 0.f)));
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int i = k * T + tid;
            if (i < HW) {
                const float nv = (v[k] - mean) * g + bt;
                st1(yp, i, act_fwd(nv, act, slope));
                if (y2p) st1(y2p, i, act_fwd(nv, 1, 0.f));
            }
        }
    }
    if (tid == 0) { mean_out[plane] = mean; rstd_out[plane] = rstd; }
}

// REX (planes of more than 16384 elements: 512 threads x 128 elements): dz alone stays in registers and the second pass reads
// x again (the plane is 256 KB and comes back from L2 / Infinity Cache) — holding xh as well would take 256 + registers per
// lane.  Same expressions, same bits.
template <typename IO, int T, int NE, bool VEC, bool REX = false>
__global__ void __launch_bounds__(T) instnorm_act_bwd_kernel(const IO* __restrict__ dy, const IO* __restrict__ y,
                                                             const IO* __restrict__ x, const float* __restrict__ bias,
                                                             const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                             const float* __restrict__ rstd_in, int act, float slope, int C, int HW,
                                                             IO* __restrict__ dx, float* __restrict__ dgamma_p,
                                                             float* __restrict__ dbeta_p, float* __restrict__ dbias_p,
                                                             float* __restrict__ sums, unsigned* __restrict__ ticket,
                                                             size_t dy_bstride, size_t y_bstride, const IO* __restrict__ dy2, size_t dy2_bstride)
{
    static_assert(!REX || VEC, "the re-reading variant exists for vector-aligned planes only");
    __shared__ float red[16];
    __shared__ int last_s;
    const int plane = blockIdx.x, c = plane % C, tid = threadIdx.x;
    const float bv = bias ? bias[c] : 0.0f;
    const float mean = mean_in[plane], rstd = rstd_in[plane];
    const size_t off = (size_t)plane * HW;
    // dy and y may be channel slices of wider tensors (the gradient / the output of a skip concatenation); x and dx are dense
    dy += (size_t)(plane / C) * dy_bstride + (size_t)c * HW - off;
    y += (size_t)(plane / C) * y_bstride + (size_t)c * HW - off;
    // dy2 (optional): the gradient of the relu'd copy that went into the child level's concatenated tensor (forward's y2): the two
    // consumers of this norm's value meet here instead of in an add kernel; relu'(y2) has the sign of y whatever `act` is
    if (dy2) dy2 += (size_t)(plane / C) * dy2_bstride + (size_t)c * HW - off;
    float dz[NE], xh[REX ? 4 : NE];
    float s1 = 0.0f, s2 = 0.0f;
    if (VEC) {
        const int n4 = HW >> 2;
#pragma unroll
        for (int k = 0; k < NE / 4; ++k) {
            const int i = k * T + tid;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), h = a;
            if (i < n4) {
                const float4 d = ld4(dy + off, i);
                const float4 o = ld4(y + off, i);
                const float4 xv = ld4(x + off, i);
                a.x = d.x * act_bwd(o.x, act, slope); a.y = d.y * act_bwd(o.y, act, slope);
                a.z = d.z * act_bwd(o.z, act, slope); a.w = d.w * act_bwd(o.w, act, slope);
                if (dy2) {
                    const float4 e2 = ld4(dy2 + off, i);
                    a.x += e2.x * act_bwd(o.x, 1, 0.f); a.y += e2.y * act_bwd(o.y, 1, 0.f);
                    a.z += e2.z * act_bwd(o.z, 1, 0.f); a.w += e2.w * act_bwd(o.w, 1, 0.f);
                }
                h.x = ((xv.x + bv) - mean) * rstd; h.y = ((xv.y + bv) - mean) * rstd;
                h.z = ((xv.z + bv) - mean) * rstd; h.w = ((xv.w + bv) - mean) * rstd;
                s1 += (a.x + a.y) + (a.z + a.w);
                s2 = __builtin_fmaf(a.x, h.x, s2); s2 = __builtin_fmaf(a.y, h.y, s2);
                s2 = __builtin_fmaf(a.z, h.z, s2); s2 = __builtin_fmaf(a.w, h.w, s2);
            }
            dz[4 * k] = a.x; dz[4 * k + 1] = a.y; dz[4 * k + 2] = a.z; dz[4 * k + 3] = a.w;
            if (!REX) { xh[4 * k] = h.x; xh[4 * k + 1] = h.y; xh[4 * k + 2] = h.z; xh[4 * k + 3] = h.w; }
        }
    } else {
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int i = k * T + tid;
            dz[k] = 0.0f; xh[k] = 0.0f;
            if (i < HW) {
                dz[k] = ld1(dy, off + i) * act_bwd(ld1(y, off + i), act, slope);
                if (dy2) dz[k] += ld1(dy2, off + i) * act_bwd(ld1(y, off + i), 1, 0.f);
                xh[k] = ((ld1(x, off + i) + bv) - mean) * rstd;
                s1 += dz[k];
                s2 = __builtin_fmaf(dz[k], xh[k], s2);
            }
        }
    }
    s1 = block_sum<T>(s1, red);
    s2 = block_sum<T>(s2, red);
    const float inv_n = 1.0f / (float)HW;
    const float m1 = s1 * inv_n, m2 = s2 * inv_n;
    const float g = (gamma ? gamma[c] : 1.0f) * rstd;
    float sdx = 0.0f;
    if (VEC) {
        const int n4 = HW >> 2;
#pragma unroll
        for (int k = 0; k < NE / 4; ++k) {
            const int i = k * T + tid;
            if (i < n4) {
                float4 h;
                if (REX) {
                    const float4 xv = ld4(x + off, i);
                    h.x = ((xv.x + bv) - mean) * rstd; h.y = ((xv.y + bv) - mean) * rstd;
                    h.z = ((xv.z + bv) - mean) * rstd; h.w = ((xv.w + bv) - mean) * rstd;
                } else {
                    h = make_float4(xh[4 * k], xh[4 * k + 1], xh[4 * k + 2], xh[4 * k + 3]);
                }
                float4 o;
                o.x = g * ((dz[4 * k] - m1) - h.x * m2);
                o.y = g * ((dz[4 * k + 1] - m1) - h.y * m2);
                o.z = g * ((dz[4 * k + 2] - m1) - h.z * m2);
                o.w = g * ((dz[4 * k + 3] - m1) - h.w * m2);
                sdx += (o.x + o.y) + (o.z + o.w);
                st4(dx + off, i, o);
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int i = k * T + tid;
            if (i < HW) {
                const float o = g * ((dz[k] - m1) - xh[k] * m2);
                sdx += o;
                st1(dx, off + i, o);
            }
        }
    }
    sdx = block_sum<T>(sdx, red);
    if (tid == 0) {
        if (dgamma_p) st_partial(dgamma_p + plane, s2);
        if (dbeta_p) st_partial(dbeta_p + plane, s1);
        if (dbias_p) st_partial(dbias_p + plane, sdx);
    }
    if (sums) {
        float* const part[3] = {dgamma_p, dbeta_p, dbias_p};
        batch_sum_by_last_plane(part, 3, sums, ticket, gridDim.x / C, C, c, &last_s);
    }
}

// act(x + bias) backward: dx = dy * act'(y), per-plane partial of the bias gradient
template <typename IO>
__global__ void __launch_bounds__(256) bias_act_bwd_kernel(const IO* __restrict__ dy, const IO* __restrict__ y, int act, float slope,
                                                           int HW, IO* __restrict__ dx, float* __restrict__ dbias_p, int C,
                                                           float* __restrict__ sums, unsigned* __restrict__ ticket,
                                                           const IO* __restrict__ dy2, size_t dy2_bstride)
{
    __shared__ float red[4];
    __shared__ int last_s;
    const int plane = blockIdx.x, tid = threadIdx.x;
    const size_t off = (size_t)plane * HW;
    // dy2 (optional): gradient of the relu'd second output (forward's y2), a channel slice of the concatenated tensor's gradient
    if (dy2) dy2 += (size_t)(plane / C) * dy2_bstride + (size_t)(plane % C) * HW - off;
    float s = 0.0f;
    if ((HW & 3) == 0) {
        const int n4 = HW >> 2;
        for (int i = tid; i < n4; i += 256) {
            const float4 d = ld4(dy + off, i);
            const float4 o = ld4(y + off, i);
            float4 r;
            r.x = d.x * act_bwd(o.x, act, slope); r.y = d.y * act_bwd(o.y, act, slope);
            r.z = d.z * act_bwd(o.z, act, slope); r.w = d.w * act_bwd(o.w, act, slope);
            if (dy2) {
                const float4 e2 = ld4(dy2 + off, i);
                r.x += e2.x * act_bwd(o.x, 1, 0.f); r.y += e2.y * act_bwd(o.y, 1, 0.f);
                r.z += e2.z * act_bwd(o.z, 1, 0.f); r.w += e2.w * act_bwd(o.w, 1, 0.f);
            }
            s += (r.x + r.y) + (r.z + r.w);
            st4(dx + off, i, r);
        }
    } else {
        for (int i = tid; i < HW; i += 256) {
            float r = ld1(dy, off + i) * act_bwd(ld1(y, off + i), act, slope);
            if (dy2) r += ld1(dy2, off + i) * act_bwd(ld1(y, off + i), 1, 0.f);
            s += r;
            st1(dx, off + i, r);
        }
    }
    s = block_sum<256>(s, red);
    if (tid == 0 && dbias_p) st_partial(dbias_p + plane, s);
    if (sums) {
        float* const part[1] = {dbias_p};
        batch_sum_by_last_plane(part, 1, sums, ticket, gridDim.x / C, C, plane % C, &last_s);
    }
}

constexpr int IN_MAX_HW = 65536;       // 256 x 256: the outermost norm of netG (planes above 16384 must be vector aligned)
constexpr int IN_MAX_HW_REG = 16384;   // largest plane a 256-thread workgroup holds in registers

// Launch shape: wave-sized workgroups for small planes, 256 threads above; a thread holds at most 64 elements.
#define IN_DISPATCH(KERNEL, IO, ...)                                                                                    \
    do {                                                                                                                \
        const bool vec = (HW & 3) == 0;                                                                                 \
        if (HW <= 256) { if (vec) KERNEL<IO, 64, 4, true><<<planes, 64, 0, st>>>(__VA_ARGS__); else KERNEL<IO, 64, 4, false><<<planes, 64, 0, st>>>(__VA_ARGS__); }          \
        else if (HW <= 1024) { if (vec) KERNEL<IO, 64, 16, true><<<planes, 64, 0, st>>>(__VA_ARGS__); else KERNEL<IO, 64, 16, false><<<planes, 64, 0, st>>>(__VA_ARGS__); }  \
        else if (HW <= 4096) { if (vec) KERNEL<IO, 256, 16, true><<<planes, 256, 0, st>>>(__VA_ARGS__); else KERNEL<IO, 256, 16, false><<<planes, 256, 0, st>>>(__VA_ARGS__); } \
        else { if (vec) KERNEL<IO, 256, 64, true><<<planes, 256, 0, st>>>(__VA_ARGS__); else KERNEL<IO, 256, 64, false><<<planes, 256, 0, st>>>(__VA_ARGS__); }              \
    } while (0)

int launch_instnorm_act_fwd(const void* x, const float* bias, const float* gamma, const float* beta, float eps, int act, float slope,
                            int B, int C, int HW, int io_bf16, void* y, float* mean, float* rstd, size_t ybs, void* y2, size_t y2bs, unsigned* tickets,
                            hipStream_t st)
{
    if (ybs == 0) ybs = (size_t)C * HW;
    if (HW > IN_MAX_HW) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_instnorm_act_forward: plane of %d elements > %d", HW, IN_MAX_HW);
    const int planes = B * C;
    if (HW > IN_MAX_HW_REG) {
        if (HW & 3) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_instnorm_act_forward: a plane of %d elements (> %d) must be a multiple of 4", HW, IN_MAX_HW_REG);
        if (io_bf16)
            instnorm_act_fwd_kernel<bf16_t, 512, 128, true><<<planes, 512, 0, st>>>(static_cast<const bf16_t*>(x), bias, gamma, beta, eps, act, slope,
                                                                                   C, HW, static_cast<bf16_t*>(y), mean, rstd, ybs, static_cast<bf16_t*>(y2), y2bs, tickets);
        else
            instnorm_act_fwd_kernel<float, 512, 128, true><<<planes, 512, 0, st>>>(static_cast<const float*>(x), bias, gamma, beta, eps, act, slope,
                                                                                  C, HW, static_cast<float*>(y), mean, rstd, ybs, static_cast<float*>(y2), y2bs, tickets);
        return check_launch("instnorm_act_fwd_kernel");
    }
    if (io_bf16)
        IN_DISPATCH(instnorm_act_fwd_kernel, bf16_t, static_cast<const bf16_t*>(x), bias, gamma, beta, eps, act, slope, C, HW,
                    static_cast<bf16_t*>(y), mean, rstd, ybs, static_cast<bf16_t*>(y2), y2bs, tickets);
    else
        IN_DISPATCH(instnorm_act_fwd_kernel, float, static_cast<const float*>(x), bias, gamma, beta, eps, act, slope, C, HW,
                    static_cast<float*>(y), mean, rstd, ybs, static_cast<float*>(y2), y2bs, tickets);
    return check_launch("instnorm_act_fwd_kernel");
}

int launch_instnorm_act_bwd(const void* dy, const void* y, const void* x, const float* bias, const float* gamma, const float* mean,
                            const float* rstd, int act, float slope, int B, int C, int HW, int io_bf16, void* dx, float* dgamma_p,
                            float* dbeta_p, float* dbias_p, float* sums, unsigned* ticket, size_t dybs, size_t ybs, const void* dy2, size_t dy2bs,
                            hipStream_t st)
{
    if (dybs == 0) dybs = (size_t)C * HW;
    if (ybs == 0) ybs = (size_t)C * HW;
    if (HW > IN_MAX_HW) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_instnorm_act_backward: plane of %d elements > %d", HW, IN_MAX_HW);
    if (sums && !ticket) return fail(IPSR_ERR_INVALID, "ipsr_instnorm_act_backward: batch sums need the C ticket words the forward entry point zeroed");
    if (!sums) ticket = nullptr;
    const int planes = B * C;
    if (HW > IN_MAX_HW_REG) {
        if (HW & 3) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_instnorm_act_backward: a plane of %d elements (> %d) must be a multiple of 4", HW, IN_MAX_HW_REG);
        if (io_bf16)
            instnorm_act_bwd_kernel<bf16_t, 512, 128, true, true><<<planes, 512, 0, st>>>(
                static_cast<const bf16_t*>(dy), static_cast<const bf16_t*>(y), static_cast<const bf16_t*>(x), bias, gamma, mean, rstd, act, slope, C, HW,
                static_cast<bf16_t*>(dx), dgamma_p, dbeta_p, dbias_p, sums, ticket, dybs, ybs, static_cast<const bf16_t*>(dy2), dy2bs);
        else
            instnorm_act_bwd_kernel<float, 512, 128, true, true><<<planes, 512, 0, st>>>(
                static_cast<const float*>(dy), static_cast<const float*>(y), static_cast<const float*>(x), bias, gamma, mean, rstd, act, slope, C, HW,
                static_cast<float*>(dx), dgamma_p, dbeta_p, dbias_p, sums, ticket, dybs, ybs, static_cast<const float*>(dy2), dy2bs);
        return check_launch("instnorm_act_bwd_kernel");
    }
    if (io_bf16)
        IN_DISPATCH(instnorm_act_bwd_kernel, bf16_t, static_cast<const bf16_t*>(dy), static_cast<const bf16_t*>(y),
                    static_cast<const bf16_t*>(x), bias, gamma, mean, rstd, act, slope, C, HW, static_cast<bf16_t*>(dx), dgamma_p, dbeta_p, dbias_p, sums, ticket, dybs, ybs, static_cast<const bf16_t*>(dy2), dy2bs);
    else
        IN_DISPATCH(instnorm_act_bwd_kernel, float, static_cast<const float*>(dy), static_cast<const float*>(y),
                    static_cast<const float*>(x), bias, gamma, mean, rstd, act, slope, C, HW, static_cast<float*>(dx), dgamma_p, dbeta_p, dbias_p, sums, ticket, dybs, ybs, static_cast<const float*>(dy2), dy2bs);
    return check_launch("instnorm_act_bwd_kernel");
}

int launch_bias_act_bwd(const void* dy, const void* y, int act, float slope, int B, int C, int HW, int io_bf16, void* dx, float* dbias_p,
                        float* sums, unsigned* ticket, const void* dy2, size_t dy2bs, hipStream_t st)
{
    if (sums && (!dbias_p || !ticket))
        return fail(IPSR_ERR_INVALID, "ipsr_bias_act_backward: batch sums need the partials and the C ticket words ipsr_bias_act zeroed");
    if (!sums) ticket = nullptr;
    if (io_bf16)
        bias_act_bwd_kernel<bf16_t><<<B * C, 256, 0, st>>>(static_cast<const bf16_t*>(dy), static_cast<const bf16_t*>(y), act, slope, HW,
                                                           static_cast<bf16_t*>(dx), dbias_p, C, sums, ticket, static_cast<const bf16_t*>(dy2), dy2bs);
    else
        bias_act_bwd_kernel<float><<<B * C, 256, 0, st>>>(static_cast<const float*>(dy), static_cast<const float*>(y), act, slope, HW,
                                                          static_cast<float*>(dx), dbias_p, C, sums, ticket, static_cast<const float*>(dy2), dy2bs);
    return check_launch("bias_act_bwd_kernel");
}

}  // namespace ipsr
