// ipsr_common.h — shared helpers of libipsr_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/ipsr_hip.h"

namespace ipsr {

// ---- error reporting (thread-local message behind ipsr_last_error()) ----
void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);
// Checks the launch that was just issued on `what`.
int check_launch(const char* what);

// opt-in event bracketing of the dominant kernel (see ipsr_profile_enable in ipsr_hip.h)
void profile_mark_start(hipStream_t st, int region = 0);
void profile_mark_stop(hipStream_t st, int region = 0, double work = 0.0, double work2 = 0.0);

// tuning switches (ipsr_debug_set_option in ipsr_hip.h): A/B aids, every key defaults to the shipped behaviour
int debug_option(int key);

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Bump allocator over the caller's workspace (256-B aligned slices).
struct Carver {
    char* base;
    size_t off;
    size_t cap;
    explicit Carver(void* p, size_t bytes) : base(static_cast<char*>(p)), off(0), cap(bytes) {}
    template <typename T>
    T* take(size_t count) {
        size_t start = align_up(off, 256);
        off = start + count * sizeof(T);
        return reinterpret_cast<T*>(base + start);
    }
    bool ok() const { return off <= cap; }
};

// XCD-aware remap (cdna_hip_programming.md T1): the dispatcher deals workgroups round-robin over the 8
// XCDs, so blocks b and b+8 share an L2.  This bijection hands each XCD a CONTIGUOUS range of logical
// ids, so logically adjacent work (same sample / same q-tile) shares an L2.  Speed only, never
// correctness.
__device__ __forceinline__ unsigned xcd_remap(unsigned wg, unsigned nwg)
{
    const unsigned q = nwg >> 3, r = nwg & 7u;
    const unsigned xcd = wg & 7u, slot = wg >> 3;
    const unsigned start = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return start + slot;
}

// 2-D grids of the transform kernels, blockIdx = (tile block, channel): the dispatcher deals consecutive linear block ids to
// different XCDs, so with few tile blocks per channel every XCD ends up writing a fixed 1-KB slot of every row of the operand
// planes.  remap2d hands each XCD a CONTIGUOUS range of (channel, tile block) instead: whole rows per XCD.
__device__ __forceinline__ void remap2d(bool on, unsigned& bx, unsigned& by)
{
    bx = blockIdx.x; by = blockIdx.y;
    if (on) {
        const unsigned L = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
        by = L / gridDim.x; bx = L - by * gridDim.x;
    }
}

// k-split partial (max, argmax) of the correlation kernel, merged on the fly by their consumers.
struct CorrPartials {
    const float* pval;     // [B, ksplit, N]
    const int32_t* pidx;   // [B, ksplit, N]
    int ksplit;
};

// Total order of (value, patch index) candidates, torch.max semantics (util/MaxCoord.py:23): the larger value wins, a NaN
// counts as larger than every number, equal values (or two NaNs) resolve to the lower index.  Total = the arg-max is an
// in-range index for EVERY input (all -inf, all NaN ...), never the 0x7fffffff start value.
__device__ __forceinline__ bool better(float v1, int i1, float v0, int i0)
{
    const bool n1 = v1 != v1, n0 = v0 != v0;
    return (v1 > v0) || (n1 && !n0) || ((v1 == v0 || (n1 && n0)) && i1 < i0);
}
// running form for candidates visited in ASCENDING index: only a strictly better value replaces (first max / first NaN wins)
__device__ __forceinline__ bool takes_over(float v, float best) { return (v > best) || (v != v && best == best); }

// arg-max over k of column q of sample b: fold the k-split partials in ascending k (lowest k on ties)
__device__ __forceinline__ void merged_argmax(const CorrPartials& cp, int b, int N, int q, float& v, int& i)
{
    const size_t base = (size_t)b * cp.ksplit * N + q;
    v = cp.pval[base];
    i = cp.pidx[base];
    for (int s = 1; s < cp.ksplit; ++s) {
        const float vs = cp.pval[base + (size_t)s * N];
        const int is = cp.pidx[base + (size_t)s * N];
        if (better(vs, is, v, i)) { v = vs; i = is; }
    }
    i = min(max(i, 0), N - 1);      // consumers index patch rows and LDS with it: in range whatever the correlation produced
}

// I/O element types: fp32, or bf16 (BASELINE config 5: activations under bf16 autocast) with all arithmetic in fp32.
// A "vector" is 4 elements: 16 bytes of fp32 or 8 bytes of bf16.
typedef unsigned short bf16_t;
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ bf16_t f2bf(float f)
{
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);                   // round to nearest even (inputs are finite)
    return (bf16_t)(u >> 16);
}
// two values into one dword.  (gfx950's v_cvt_pk_bf16_f32 — what __builtin_convertvector to a __bf16 vector compiles to — was measured in
// the thin kernels' gathers and epilogues and is no faster than these integer operations: 0.156 -> 0.228 ms on the 6 -> 64 stream kernel.)
__device__ __forceinline__ unsigned f2bf2(float lo, float hi) { return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16); }
__device__ __forceinline__ float ld1(const float* p, size_t i) { return p[i]; }
__device__ __forceinline__ float ld1(const bf16_t* p, size_t i) { return bf2f(p[i]); }
__device__ __forceinline__ void st1(float* p, size_t i, float v) { p[i] = v; }
__device__ __forceinline__ void st1(bf16_t* p, size_t i, float v) { p[i] = f2bf(v); }
__device__ __forceinline__ float4 ld4(const float* p, size_t i4) { return reinterpret_cast<const float4*>(p)[i4]; }
__device__ __forceinline__ float4 ld4(const bf16_t* p, size_t i4)
{
    const uint2 r = reinterpret_cast<const uint2*>(p)[i4];
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                       __uint_as_float(r.y & 0xffff0000u));
}
__device__ __forceinline__ void st4(float* p, size_t i4, float4 v) { reinterpret_cast<float4*>(p)[i4] = v; }
__device__ __forceinline__ void st4(bf16_t* p, size_t i4, float4 v)
{
    uint2 r;
    r.x = (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16);
    r.y = (unsigned)f2bf(v.z) | ((unsigned)f2bf(v.w) << 16);
    reinterpret_cast<uint2*>(p)[i4] = r;
}

// mask_point_idx[l] as the kernels use it: the caller promises values in [0, N) (the host wrappers check what they can
// without a device sync); a stray value must give a wrong answer, never an out-of-bounds access.
__device__ __forceinline__ int mpi_at(const int32_t* __restrict__ mpi, int l, int N)
{
    const int q = mpi[l];
    return q < 0 ? 0 : (q >= N ? N - 1 : q);
}

// ---- device-side kernels' launchers (one per .hip file) ----
int launch_feat_mask(const uint8_t* mask, int H, int W, int layers, float threshold, uint8_t* feat,
                     void* ws, size_t ws_bytes, hipStream_t st);
int launch_index_prep(const uint8_t* feat, int h, int w, int patch, int stride, int mask_thred,
                      int32_t* flag, int32_t* mask_point_idx, int32_t* count, hipStream_t st);
// xT may be NULL (then the patch-major copy is not produced).
// ldx / ldn: row strides of x / xn (0 = N); with ldn > N the pad columns of xn are zero-filled.
int launch_patch_normalize(const float* x, int B, int C, int N, float* xn, float* xT, int Cp, float* inv,
                           hipStream_t st, int ldx = 0, int ldn = 0);
// shift_sz > 1 (unfold.hip): p x p windows, stride 1, rows k = (c*p+dy)*p+dx; xu row stride ld >= N' (pad columns zeroed)
int launch_unfold(const float* x, int B, int C, int h, int w, int patch, int ld, float* xu, hipStream_t st);
// addend != NULL: out = addend + fold(yu)
int launch_fold(const float* yu, int B, int C, int h, int w, int patch, float* out, hipStream_t st, const float* addend = nullptr);
// shift_sz > 1 (shifted-sum form): per-position norms -> window inverse norms + the patch-major raw windows; the window
// correlation + arg-max partials from the 1x1 correlation matrix R [B][hw][hw]
int launch_window_prepare(const float* x, int B, int C, int h, int w, int patch, float* n1, float* inv, float* xT, int Cp, hipStream_t st);
size_t window_corr_ws_bytes(int B, int Np);
int launch_window_corr_argmax(const float* R, const float* inv, int B, int h, int w, int patch, void* ws, size_t ws_bytes, hipStream_t st,
                              CorrPartials* partials);
size_t corr_argmax_ws_bytes(int B, int C, int N);
// ld: row stride of xn and ref (0 = N).  ld > N: operands zero-padded to whole 128-column tiles, patches k >= N are
// excluded from the arg-max.
// partials != NULL: skip the merge kernel and hand the k-split partials to the caller (ind/vmax are then written by the
// consumer that merges them: the attention stage kernel)
int launch_corr_argmax(const float* xn, const float* ref, int B, int C, int N, int32_t* ind, float* vmax,
                       float* S_out, void* ws, size_t ws_bytes, hipStream_t st, CorrPartials* partials = nullptr, int ld = 0);

// bf16-MFMA variant (opt-in): same fp32 operands, packed to bf16 inside; needs corr_bf16_supported(C, ld)
bool corr_bf16_supported(int C, int ld);
size_t corr_argmax_bf16_ws_bytes(int B, int C, int N, int ld = 0);
int launch_corr_argmax_bf16(const float* xn, const float* ref, int B, int C, int N, int32_t* ind, float* vmax,
                            void* ws, size_t ws_bytes, hipStream_t st, CorrPartials* partials = nullptr, int ld = 0);

// conv_gemm.hip — implicit-GEMM convolutions on the fp32 matrix cores (see the file header)
size_t conv_gemm_ws_bytes(int transposed, int B, int Cred, int M, int Hin, int Win, int Hout, int Wout, int k, int stride, int pad, int dil);
int launch_conv_gemm(bool transposed, bool w_cred_major, const float* x, const float* w, float* y, int B, int Cred, int M,
                     int Hin, int Win, int Hout, int Wout, int k, int stride, int pad, int dil, void* ws, size_t ws_bytes, hipStream_t st);

struct AttnArgs {
    const float* xT;       // [B,N,Cp] patch-major raw copy, zero padded to Cp = roundup(C,8)
    const float* inv;      // [B,N]
    int32_t* ind;          // [B,N]  written by the stage kernel (merged arg-max)
    float* vmax;           // [B,N]
    CorrPartials part;     // k-split partials of the correlation kernel
    const int32_t* mpi;    // [M] shared by the batch (mpi_stride 0), or [B][M] one row per sample (mpi_stride M)
    int mpi_stride;
    const int32_t* mcount; // NULL: every sample has M masked positions; else [B] device counts (per-sample masks), each <= M
    int B, C, Cp, N, M, Mc;   // M = capacity (row strides / CSR layout); Mc = roundup(M, 32): row stride of the compressed attention
    // workspace
    float* wn;             // [B,M]
    float* wo;             // [B,M]
    int32_t* kq;           // [B,M]   ind[mpi[l]]
    int32_t* jq;           // [B,M]   rank of kq[l] among the active columns
    int32_t* dlist;        // [B,Mc]  active columns, ascending k (padded with a valid patch index)
    int32_t* mprime;       // [B]     number of active columns
    int32_t* rankflag;     // [B,N]   rank of column k among the active ones; inactive: -(#active columns below k) - 1
    float* ac;             // [B,M,Mc] compressed attention rows
    // outputs
    float* attn;           // [B,M,N] dense attention rows, optional (NULL = not materialised)
    float* out;            // [B,C,N]
    int32_t* bwd_index;    // [B, ipsr_bwd_index_ints(N,M)], optional
};
int launch_attention(const AttnArgs& a, hipStream_t st);

int launch_backward(const float* g, const int32_t* mpi, int M, const float* attn, const int32_t* bwd_index,
                    float triple_w, int B, int C, int N, float* gin, hipStream_t st, int identity = 1);

// pointwise.hip — VGG16 feature net glue (bias / ReLU / 2x2 max-pool in one pass)
int launch_bias_act(void* x, const float* bias, int B, int C, int HW, int act, float slope, int io_bf16, void* y2, size_t y2_bstride,
                    unsigned* tickets, hipStream_t st);
int launch_bias_relu_pool2(const void* x, const float* bias, int B, int C, int H, int W, int io_bf16, void* y, hipStream_t st);
int launch_cat_relu_fwd(const void* y, const void* x, int B, int C1, int C2, int HW, int io_bf16, void* out, hipStream_t st);
int launch_cat_relu_bwd(const void* g, const void* out, int B, int C1, int C2, int HW, int io_bf16, void* dy, void* dx, hipStream_t st);

// instnorm.hip — conv-bias + InstanceNorm2d + activation, fused forward / backward (one (sample, channel) plane per workgroup)
int launch_instnorm_act_fwd(const void* x, const float* bias, const float* gamma, const float* beta, float eps, int act, float slope,
                            int B, int C, int HW, int io_bf16, void* y, float* mean, float* rstd, size_t y_bstride, void* y2, size_t y2_bstride,
                            unsigned* tickets, hipStream_t st);
int launch_instnorm_act_bwd(const void* dy, const void* y, const void* x, const float* bias, const float* gamma, const float* mean,
                            const float* rstd, int act, float slope, int B, int C, int HW, int io_bf16, void* dx, float* dgamma_p,
                            float* dbeta_p, float* dbias_p, float* sums, unsigned* tickets, size_t dy_bstride, size_t y_bstride, const void* dy2,
                            size_t dy2_bstride, hipStream_t st);
int launch_bias_act_bwd(const void* dy, const void* y, int act, float slope, int B, int C, int HW, int io_bf16, void* dx, float* dbias_p,
                        float* sums, unsigned* tickets, const void* dy2, size_t dy2_bstride, hipStream_t st);

size_t innercos_ws_bytes(int B, int Cuse, int N);
int launch_innercos_loss(const float* x, int B, int Cx, int Cuse, int N, const float* mask, const float* target,
                         float strength, float* loss, void* ws, size_t ws_bytes, hipStream_t st);
int launch_innercos_loss_fused(const float* x, int B, int Cx, int Cuse, int N, const float* mask, const float* target, float strength, float* loss,
                               void* ws, size_t ws_bytes, unsigned* ticket, hipStream_t st);
int launch_innercos_backward(const float* x, int B, int Cx, int Cuse, int N, const float* mask,
                             const float* target, float strength, const float* grad_loss, float* grad_x,
                             hipStream_t st);

}  // namespace ipsr
