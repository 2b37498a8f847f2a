// api.hip — the C-ABI of libipsr_hip.so (declared in include/ipsr_hip.h): argument checks, workspace
// carving and the launch sequence of the layer.  No allocation, no synchronisation, no global state
// besides the thread-local error string.
#include <cstdarg>
#include <cstdio>

#include "ipsr_common.h"

namespace ipsr {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char* what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(IPSR_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return IPSR_OK;
}

// ---- opt-in profiling rings ----------------------------------------------------------------------------
// region 0: the correlation + arg-max kernel; 1: a whole ipsr_forward; 2: a whole ipsr_backward(_patch);
// 3: every launch of the Winograd GEMM kernel (the convolutions' matrix-core kernel), with the launch's flop count;
// 4: every launch of the direct bf16 convolution kernels (conv_bf16.hip), likewise; 5: every launch of the InnerCos loss kernel (bytes)
constexpr int N_REGIONS = 6;
struct EvRing {
    hipEvent_t* ev = nullptr;      // 2*capacity events: start0, stop0, start1, ...
    double* work = nullptr;        // per entry: work units the caller attached (flops as EXECUTED, padded sizes), 0 if none
    double* work2 = nullptr;       // per entry: the USEFUL part of that work (unpadded sizes)
    int cap = 0, n = 0;
    bool open = false;
};
static EvRing g_ring[N_REGIONS];

void profile_mark_start(hipStream_t st, int region)
{
    EvRing& r = g_ring[region];
    if (r.cap == 0 || r.n >= r.cap) return;
    (void)hipEventRecord(r.ev[2 * r.n], st);
    r.open = true;
}

void profile_mark_stop(hipStream_t st, int region, double work, double work2)
{
    EvRing& r = g_ring[region];
    if (!r.open) return;
    (void)hipEventRecord(r.ev[2 * r.n + 1], st);
    r.work[r.n] = work;
    r.work2[r.n] = work2;
    r.open = false;
    ++r.n;
}

// brackets a whole entry point (every return path)
struct ProfileScope {
    hipStream_t st;
    int region;
    ProfileScope(hipStream_t s, int r) : st(s), region(r) { profile_mark_start(st, region); }
    ~ProfileScope() { profile_mark_stop(st, region); }
};

static int g_debug_opt[16] = {0};
int debug_option(int key) { return (key >= 0 && key < 16) ? g_debug_opt[key] : 0; }

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

struct FwdPlan {
    int N, Cp, Mc;
    int K, ld;         // patch length C*p*p and the row stride of the unfolded operands (p > 1: N rounded up to 128)
    size_t total;
};

// workspace slices in carve order (all 256-B aligned)
enum { WS_XN, WS_XT, WS_INV, WS_CORR, WS_WN, WS_WO, WS_KQ, WS_JQ, WS_DLIST, WS_MPRIME, WS_RANKFLAG, WS_AC,
       WS_XU, WS_RU, WS_OU, WS_COUNT };

// patch == 1: the feature IS the patch matrix.  patch > 1: C/N below are the unfolded K = C*p*p and the window grid
// N' = (h-p+1)(w-p+1); three more slices hold the unfolded x, the unfolded ref and the un-folded result.
static FwdPlan plan_forward(int B, int C, int h, int w, int M, int patch, size_t* sizes, bool corr_bf16 = false)
{
    FwdPlan p;
    p.K = C * patch * patch;
    p.N = (h - patch + 1) * (w - patch + 1);
    p.ld = p.N;
    p.Cp = (p.K + 7) & ~7;
    p.Mc = M > 0 ? (M + 31) & ~31 : 32;
    const size_t Mx = M > 0 ? M : 1;
    size_t sz[WS_COUNT];
    const size_t hw = (size_t)h * w;
    // patch > 1 (shifted-sum form): WS_XN = the 1x1 correlation matrix R [B][hw][hw]; WS_XU = the per-position norms n1 [B][hw];
    // WS_CORR = partials of the 1x1 correlation launch; WS_RU = partials of the window arg-max.  Nothing is unfolded but xT.
    sz[WS_XN] = patch > 1 ? (size_t)B * hw * hw * 4 : (size_t)B * p.K * p.ld * 4;
    sz[WS_XT] = (size_t)B * p.N * p.Cp * 4;
    sz[WS_INV] = (size_t)B * p.N * 4;
    if (patch > 1) sz[WS_CORR] = corr_argmax_ws_bytes(B, C, (int)hw);
    else sz[WS_CORR] = corr_bf16 ? corr_argmax_bf16_ws_bytes(B, p.K, p.N, p.ld) : corr_argmax_ws_bytes(B, p.K, p.N);
    sz[WS_XU] = patch > 1 ? (size_t)B * hw * 4 : 0;
    sz[WS_RU] = patch > 1 ? window_corr_ws_bytes(B, p.N) : 0;
    sz[WS_OU] = patch > 1 ? (size_t)B * p.K * p.N * 4 : 0;
    sz[WS_WN] = sz[WS_WO] = sz[WS_KQ] = sz[WS_JQ] = (size_t)B * Mx * 4;
    sz[WS_DLIST] = (size_t)B * p.Mc * 4;
    sz[WS_MPRIME] = (size_t)B * 4;
    sz[WS_RANKFLAG] = (size_t)B * p.N * 4;
    sz[WS_AC] = (size_t)B * Mx * p.Mc * 4;
    p.total = 256;
    for (int i = 0; i < WS_COUNT; ++i) {
        sz[i] = align_up(sz[i], 256);
        p.total += sz[i];
        if (sizes) sizes[i] = sz[i];
    }
    return p;
}

}  // namespace ipsr

using namespace ipsr;

extern "C" {

int ipsr_abi_version(void) { return 14; }

int ipsr_debug_set_option(int key, int value)
{
    if (key < 0 || key >= 16) return fail(IPSR_ERR_INVALID, "ipsr_debug_set_option: key %d outside [0, 16)", key);
    g_debug_opt[key] = value;
    return IPSR_OK;
}

const char* ipsr_last_error(void) { return g_err; }

int ipsr_profile_enable_mask(int capacity, unsigned region_mask)
{
    for (EvRing& r : g_ring) {
        for (int i = 0; i < 2 * r.cap; ++i) (void)hipEventDestroy(r.ev[i]);
        delete[] r.ev;
        delete[] r.work;
        delete[] r.work2;
        r = EvRing();
    }
    if (capacity <= 0) return IPSR_OK;
    for (int ri = 0; ri < N_REGIONS; ++ri) {
        if (!((region_mask >> ri) & 1u)) continue;
        EvRing& r = g_ring[ri];
        const int cap_r = ri >= 3 ? 256 * capacity : capacity;        // tens of GEMM / direct-convolution launches per training step
        r.work = new double[(size_t)cap_r];
        r.work2 = new double[(size_t)cap_r];
        r.ev = new hipEvent_t[2 * (size_t)cap_r];
        for (int i = 0; i < 2 * cap_r; ++i)
            if (hipEventCreate(&r.ev[i]) != hipSuccess) return fail(IPSR_ERR_LAUNCH, "ipsr_profile_enable: hipEventCreate failed");
        r.cap = cap_r;
    }
    return IPSR_OK;
}

int ipsr_profile_enable(int capacity) { return ipsr_profile_enable_mask(capacity, 0x7u); }

int ipsr_profile_read_region(int region, float* ms, int max_n)
{
    if (!ms || max_n < 0 || region < 0 || region >= N_REGIONS) return fail(IPSR_ERR_INVALID, "ipsr_profile_read_region: bad arguments");
    EvRing& r = g_ring[region];
    int n = 0;
    for (int i = 0; i < r.n && n < max_n; ++i) {
        float t = 0.0f;
        if (hipEventSynchronize(r.ev[2 * i + 1]) != hipSuccess) break;
        if (hipEventElapsedTime(&t, r.ev[2 * i], r.ev[2 * i + 1]) != hipSuccess) break;
        ms[n++] = t;
    }
    r.n = 0;
    return n;
}

int ipsr_profile_read(float* ms, int max_n) { return ipsr_profile_read_region(0, ms, max_n); }

int ipsr_profile_read_region_work2(int region, float* ms, double* work, double* useful, int max_n)
{
    if (!ms || !work || max_n < 0 || region < 0 || region >= N_REGIONS) return fail(IPSR_ERR_INVALID, "ipsr_profile_read_region_work: bad arguments");
    EvRing& r = g_ring[region];
    const int avail = r.n;
    for (int i = 0; i < avail && i < max_n; ++i) {
        work[i] = r.work[i];
        if (useful) useful[i] = r.work2[i];
    }
    return ipsr_profile_read_region(region, ms, max_n);
}

int ipsr_profile_read_region_work(int region, float* ms, double* work, int max_n) { return ipsr_profile_read_region_work2(region, ms, work, nullptr, max_n); }

size_t ipsr_feat_mask_workspace_bytes(int H, int W, int layers)
{
    (void)layers;
    if (H < 2 || W < 2) return 0;
    const size_t h1 = (size_t)((H - 2) / 2 + 1), w1 = (size_t)((W - 2) / 2 + 1);
    return 2 * align_up(h1 * w1 * sizeof(uint32_t), 256);
}

int ipsr_feat_mask(const uint8_t* mask, int H, int W, int layers, float threshold, uint8_t* feat,
                   void* ws, size_t ws_bytes, void* stream)
{
    if (!mask || !feat) return fail(IPSR_ERR_INVALID, "ipsr_feat_mask: null pointer");
    if (H < 2 || W < 2 || layers < 1 || layers > 5) return fail(IPSR_ERR_INVALID, "ipsr_feat_mask: bad size H=%d W=%d layers=%d", H, W, layers);
    if (layers > 1 && !ws) return fail(IPSR_ERR_WORKSPACE, "ipsr_feat_mask: workspace required");
    return launch_feat_mask(mask, H, W, layers, threshold, feat, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

int ipsr_index_prep(const uint8_t* feat, int h, int w, int patch, int stride, int mask_thred,
                    int32_t* flag, int32_t* mask_point_idx, int32_t* count, void* stream)
{
    if (!feat || !flag || !mask_point_idx || !count) return fail(IPSR_ERR_INVALID, "ipsr_index_prep: null pointer");
    if (patch < 1 || stride < 1 || h < patch || w < patch) return fail(IPSR_ERR_INVALID, "ipsr_index_prep: bad geometry h=%d w=%d patch=%d stride=%d", h, w, patch, stride);
    return launch_index_prep(feat, h, w, patch, stride, mask_thred, flag, mask_point_idx, count, static_cast<hipStream_t>(stream));
}

int ipsr_patch_normalize(const float* x, int B, int C, int N, float* xn, float* inv, void* stream)
{
    if (!x || !xn || !inv) return fail(IPSR_ERR_INVALID, "ipsr_patch_normalize: null pointer");
    if (B < 1 || C < 1 || N < 1) return fail(IPSR_ERR_INVALID, "ipsr_patch_normalize: bad size B=%d C=%d N=%d", B, C, N);
    return launch_patch_normalize(x, B, C, N, xn, nullptr, (C + 7) & ~7, inv, static_cast<hipStream_t>(stream));
}

size_t ipsr_corr_argmax_workspace_bytes(int B, int C, int N)
{
    if (B < 1 || C < 1 || N < 1) return 0;
    return corr_argmax_ws_bytes(B, C, N);
}

int ipsr_corr_argmax(const float* xn, const float* ref, int B, int C, int N, int32_t* ind, float* vmax,
                     float* S_out, void* ws, size_t ws_bytes, void* stream)
{
    if (!xn || !ref || !ind || !vmax || !ws) return fail(IPSR_ERR_INVALID, "ipsr_corr_argmax: null pointer");
    if (B < 1 || C < 1 || N < 1) return fail(IPSR_ERR_INVALID, "ipsr_corr_argmax: bad size B=%d C=%d N=%d", B, C, N);
    if (!aligned16(xn) || !aligned16(ref)) return fail(IPSR_ERR_INVALID, "ipsr_corr_argmax: xn/ref must be 16-byte aligned");
    return launch_corr_argmax(xn, ref, B, C, N, ind, vmax, S_out, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

size_t ipsr_bwd_index_ints(int N, int M)
{
    if (N < 1 || M < 0 || M > N) return 0;
    return 2 * ((size_t)N + 1) + (size_t)N + (size_t)M * (M + 1);
}

size_t ipsr_forward_workspace_bytes(int B, int C, int h, int w, int M, int patch, int stride)
{
    if (B < 1 || C < 1 || patch < 1 || h < patch || w < patch || M < 0 || stride != 1) return 0;
    return plan_forward(B, C, h, w, M, patch, nullptr).total;
}

static int forward_impl(const float* x, const float* ref, const int32_t* mask_point_idx, int M,
                        int B, int C, int h, int w, int patch, int stride,
                        float* out, int32_t* ind, float* vmax, float* attn_rows, int32_t* bwd_index,
                        void* ws, size_t ws_bytes, void* stream, bool corr_bf16,
                        const int32_t* mcount = nullptr, int mpi_stride = 0)
{
    if (!x || !ref || !out || !ind || !vmax || !ws) return fail(IPSR_ERR_INVALID, "ipsr_forward: null pointer");
    if (B < 1 || C < 1 || h < 1 || w < 1 || M < 0) return fail(IPSR_ERR_INVALID, "ipsr_forward: bad size B=%d C=%d h=%d w=%d M=%d", B, C, h, w, M);
    if (stride != 1 || patch < 1)
        return fail(IPSR_ERR_UNSUPPORTED, "ipsr_forward: only stride=1 is implemented (got shift_sz=%d, stride=%d)", patch, stride);
    if (h < patch || w < patch || M > (h - patch + 1) * (w - patch + 1))
        return fail(IPSR_ERR_INVALID, "ipsr_forward: bad geometry h=%d w=%d shift_sz=%d M=%d", h, w, patch, M);
    if (M > 0 && !mask_point_idx) return fail(IPSR_ERR_INVALID, "ipsr_forward: M > 0 needs mask_point_idx");
    if (!aligned16(x) || !aligned16(ref) || !aligned16(out) || !aligned16(ws) || (attn_rows && !aligned16(attn_rows)))
        return fail(IPSR_ERR_INVALID, "ipsr_forward: x/ref/out/attn_rows/ws must be 16-byte aligned");
    size_t sz[WS_COUNT];
    const FwdPlan p = plan_forward(B, C, h, w, M, patch, sz, corr_bf16);
    if (corr_bf16 && !corr_bf16_supported(p.K, p.ld))
        return fail(IPSR_ERR_UNSUPPORTED, "ipsr_forward_bf16corr: the bf16 correlation needs C*p*p %% 64 == 0 and N %% 128 == 0 for shift_sz = 1 (got %d, %d)", p.K, p.N);
    if (ws_bytes < p.total) return fail(IPSR_ERR_WORKSPACE, "ipsr_forward: workspace %zu < %zu", ws_bytes, p.total);
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfileScope scope(st, 1);

    char* slice[WS_COUNT];
    char* base = static_cast<char*>(ws);
    for (int i = 0; i < WS_COUNT; ++i) { slice[i] = base; base += sz[i]; }
    float* xn = reinterpret_cast<float*>(slice[WS_XN]);
    float* xT = reinterpret_cast<float*>(slice[WS_XT]);
    float* inv = reinterpret_cast<float*>(slice[WS_INV]);

    // shift_sz > 1: from here on a "channel" is one of the K numbers of a patch and a "position" one of the N' windows.
    // The result comes back as patches and is overlap-added at the end.
    float* outs = out;
    AttnArgs a{};                      // every field defined (mcount = NULL, mpi_stride = 0: one shared mask)
    if (patch > 1) {
        if (corr_bf16) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_forward_bf16corr: shift_sz > 1 runs the fp32 shifted-sum correlation only");
        // window norms + patch-major raw windows straight from the feature; the correlation of p x p windows = sums of shifted
        // diagonals of the 1x1 correlation R = x^T ref (8x fewer flops at p = 3, nothing unfolded): corr_argmax.hip
        float* R = xn;                                              // WS_XN slot
        float* n1 = reinterpret_cast<float*>(slice[WS_XU]);
        if (int rc = launch_window_prepare(x, B, C, h, w, patch, n1, inv, xT, p.Cp, st)) return rc;
        CorrPartials unused;
        if (int rc = launch_corr_argmax(x, ref, B, C, h * w, nullptr, nullptr, R, slice[WS_CORR], sz[WS_CORR], st, &unused, 0)) return rc;
        if (int rc = launch_window_corr_argmax(R, inv, B, h, w, patch, slice[WS_RU], sz[WS_RU], st, &a.part)) return rc;
        outs = reinterpret_cast<float*>(slice[WS_OU]);
    } else {
        if (int rc = launch_patch_normalize(x, B, p.K, p.N, xn, xT, p.Cp, inv, st, p.ld, p.ld)) return rc;
        if (corr_bf16) {
            if (int rc = launch_corr_argmax_bf16(xn, ref, B, p.K, p.N, ind, vmax, slice[WS_CORR], sz[WS_CORR], st, &a.part, p.ld)) return rc;
        } else {
            if (int rc = launch_corr_argmax(xn, ref, B, p.K, p.N, ind, vmax, nullptr, slice[WS_CORR], sz[WS_CORR], st, &a.part, p.ld)) return rc;
        }
    }
    a.xT = xT; a.inv = inv; a.ind = ind; a.vmax = vmax; a.mpi = mask_point_idx;
    a.mcount = mcount; a.mpi_stride = mpi_stride;
    a.B = B; a.C = p.K; a.Cp = p.Cp; a.N = p.N; a.M = M; a.Mc = p.Mc;
    a.wn = reinterpret_cast<float*>(slice[WS_WN]);
    a.wo = reinterpret_cast<float*>(slice[WS_WO]);
    a.kq = reinterpret_cast<int32_t*>(slice[WS_KQ]);
    a.jq = reinterpret_cast<int32_t*>(slice[WS_JQ]);
    a.dlist = reinterpret_cast<int32_t*>(slice[WS_DLIST]);
    a.mprime = reinterpret_cast<int32_t*>(slice[WS_MPRIME]);
    a.rankflag = reinterpret_cast<int32_t*>(slice[WS_RANKFLAG]);
    a.ac = reinterpret_cast<float*>(slice[WS_AC]);
    a.attn = attn_rows; a.out = outs; a.bwd_index = bwd_index;
    if (int rc = launch_attention(a, st)) return rc;
    if (patch > 1) return launch_fold(outs, B, C, h, w, patch, out, st);
    return IPSR_OK;
}

int ipsr_forward(const float* x, const float* ref, const int32_t* mask_point_idx, int M,
                 int B, int C, int h, int w, int patch, int stride,
                 float* out, int32_t* ind, float* vmax, float* attn_rows, int32_t* bwd_index,
                 void* ws, size_t ws_bytes, void* stream)
{
    return forward_impl(x, ref, mask_point_idx, M, B, C, h, w, patch, stride, out, ind, vmax, attn_rows, bwd_index, ws, ws_bytes, stream, false);
}

int ipsr_forward_masks(const float* x, const float* ref, const int32_t* mask_point_idx, int mpi_stride, const int32_t* counts, int Mcap,
                       int B, int C, int h, int w, int patch, int stride,
                       float* out, int32_t* ind, float* vmax, float* attn_rows, int32_t* bwd_index,
                       void* ws, size_t ws_bytes, void* stream, int corr_bf16)
{
    if (!counts) return fail(IPSR_ERR_INVALID, "ipsr_forward_masks: counts is NULL (use ipsr_forward for one host-known count)");
    if (mpi_stride != 0 && mpi_stride != Mcap) return fail(IPSR_ERR_INVALID, "ipsr_forward_masks: mpi_stride must be 0 or Mcap (got %d, Mcap %d)", mpi_stride, Mcap);
    if (Mcap < 1) return fail(IPSR_ERR_INVALID, "ipsr_forward_masks: Mcap must be >= 1");
    return forward_impl(x, ref, mask_point_idx, Mcap, B, C, h, w, patch, stride, out, ind, vmax, attn_rows, bwd_index, ws, ws_bytes, stream,
                        corr_bf16 != 0, counts, mpi_stride);
}

size_t ipsr_forward_bf16corr_workspace_bytes(int B, int C, int h, int w, int M, int patch, int stride)
{
    if (B < 1 || C < 1 || patch < 1 || h < patch || w < patch || M < 0 || stride != 1) return 0;
    return plan_forward(B, C, h, w, M, patch, nullptr, true).total;
}

int ipsr_forward_bf16corr(const float* x, const float* ref, const int32_t* mask_point_idx, int M,
                          int B, int C, int h, int w, int patch, int stride,
                          float* out, int32_t* ind, float* vmax, float* attn_rows, int32_t* bwd_index,
                          void* ws, size_t ws_bytes, void* stream)
{
    return forward_impl(x, ref, mask_point_idx, M, B, C, h, w, patch, stride, out, ind, vmax, attn_rows, bwd_index, ws, ws_bytes, stream, true);
}

size_t ipsr_corr_argmax_bf16_workspace_bytes(int B, int C, int N)
{
    if (B < 1 || C < 1 || N < 1) return 0;
    return corr_argmax_bf16_ws_bytes(B, C, N);
}

int ipsr_corr_argmax_bf16(const float* xn, const float* ref, int B, int C, int N, int32_t* ind, float* vmax,
                          void* ws, size_t ws_bytes, void* stream)
{
    if (!xn || !ref || !ind || !vmax || !ws) return fail(IPSR_ERR_INVALID, "ipsr_corr_argmax_bf16: null pointer");
    if (B < 1 || C < 1 || N < 1) return fail(IPSR_ERR_INVALID, "ipsr_corr_argmax_bf16: bad size B=%d C=%d N=%d", B, C, N);
    if (!aligned16(xn) || !aligned16(ref) || !aligned16(ws)) return fail(IPSR_ERR_INVALID, "ipsr_corr_argmax_bf16: xn/ref/ws must be 16-byte aligned");
    return launch_corr_argmax_bf16(xn, ref, B, C, N, ind, vmax, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

int ipsr_backward(const float* grad_out, const int32_t* mask_point_idx, int M, const float* attn_rows,
                  const int32_t* bwd_index, float triple_w, int B, int C, int h, int w, float* grad_in, void* stream)
{
    if (!grad_out || !grad_in || !bwd_index) return fail(IPSR_ERR_INVALID, "ipsr_backward: null pointer");
    if (B < 1 || C < 1 || h < 1 || w < 1 || M < 0) return fail(IPSR_ERR_INVALID, "ipsr_backward: bad size");
    ProfileScope scope(static_cast<hipStream_t>(stream), 2);
    return launch_backward(grad_out, mask_point_idx, M, attn_rows, bwd_index, triple_w, B, C, h * w, grad_in, static_cast<hipStream_t>(stream));
}

size_t ipsr_backward_workspace_bytes(int B, int C, int h, int w, int patch)
{
    if (B < 1 || C < 1 || patch < 1 || h < patch || w < patch) return 0;
    if (patch == 1) return 0;
    const size_t un = (size_t)B * C * patch * patch * (size_t)(h - patch + 1) * (w - patch + 1) * sizeof(float);
    return 2 * align_up(un, 256) + 256;
}

int ipsr_backward_patch(const float* grad_out, int M, const int32_t* bwd_index, float triple_w,
                        int B, int C, int h, int w, int patch, float* grad_in, void* ws, size_t ws_bytes, void* stream)
{
    if (!grad_out || !grad_in || !bwd_index) return fail(IPSR_ERR_INVALID, "ipsr_backward_patch: null pointer");
    if (B < 1 || C < 1 || patch < 1 || h < patch || w < patch || M < 0) return fail(IPSR_ERR_INVALID, "ipsr_backward_patch: bad size");
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfileScope scope(st, 2);
    if (patch == 1) return launch_backward(grad_out, nullptr, M, nullptr, bwd_index, triple_w, B, C, h * w, grad_in, st);
    const size_t need = ipsr_backward_workspace_bytes(B, C, h, w, patch);
    if (!ws || ws_bytes < need) return fail(IPSR_ERR_WORKSPACE, "ipsr_backward_patch: workspace %zu < %zu", ws_bytes, need);
    const int K = C * patch * patch, Np = (h - patch + 1) * (w - patch + 1);
    Carver cv(ws, ws_bytes);
    float* gu = cv.take<float>((size_t)B * K * Np);
    float* tu = cv.take<float>((size_t)B * K * Np);
    if (int rc = launch_unfold(grad_out, B, C, h, w, patch, Np, gu, st)) return rc;
    if (int rc = launch_backward(gu, nullptr, M, nullptr, bwd_index, triple_w, B, K, Np, tu, st, 0)) return rc;
    return launch_fold(tu, B, C, h, w, patch, grad_in, st, grad_out);
}

static inline bool aligned_io(const void* p, int io_bf16) { return (reinterpret_cast<uintptr_t>(p) & (io_bf16 ? 7u : 15u)) == 0; }

int ipsr_bias_act(void* x, const float* bias, int B, int C, int HW, int act, float slope, int io_bf16, unsigned* tickets, void* stream)
{
    if (!x) return fail(IPSR_ERR_INVALID, "ipsr_bias_act: null pointer");
    if (B < 1 || C < 1 || HW < 1) return fail(IPSR_ERR_INVALID, "ipsr_bias_act: bad size B=%d C=%d HW=%d", B, C, HW);
    if ((HW & 3) == 0 && !aligned_io(x, io_bf16)) return fail(IPSR_ERR_INVALID, "ipsr_bias_act: x is not vector aligned");
    return launch_bias_act(x, bias, B, C, HW, act, slope, io_bf16, nullptr, 0, tickets, static_cast<hipStream_t>(stream));
}

int ipsr_bias_act_skip(void* x, const float* bias, int B, int C, int HW, int act, float slope, int io_bf16, void* y2, size_t y2_batch_stride,
                       unsigned* tickets, void* stream)
{
    if (!x || !y2) return fail(IPSR_ERR_INVALID, "ipsr_bias_act_skip: null pointer");
    if (B < 1 || C < 1 || HW < 1 || y2_batch_stride < (size_t)C * HW) return fail(IPSR_ERR_INVALID, "ipsr_bias_act_skip: bad size B=%d C=%d HW=%d", B, C, HW);
    if ((HW & 3) == 0 && (!aligned_io(x, io_bf16) || !aligned_io(y2, io_bf16) || (y2_batch_stride & 3)))
        return fail(IPSR_ERR_INVALID, "ipsr_bias_act_skip: tensors are not vector aligned");
    return launch_bias_act(x, bias, B, C, HW, act, slope, io_bf16, y2, y2_batch_stride, tickets, static_cast<hipStream_t>(stream));
}

int ipsr_bias_relu_pool2(const void* x, const float* bias, int B, int C, int H, int W, int io_bf16, void* y, void* stream)
{
    if (!x || !y) return fail(IPSR_ERR_INVALID, "ipsr_bias_relu_pool2: null pointer");
    if (B < 1 || C < 1 || H < 2 || W < 2) return fail(IPSR_ERR_INVALID, "ipsr_bias_relu_pool2: bad size B=%d C=%d H=%d W=%d", B, C, H, W);
    if ((W & 3) == 0 && !aligned_io(x, io_bf16)) return fail(IPSR_ERR_INVALID, "ipsr_bias_relu_pool2: x is not vector aligned");
    return launch_bias_relu_pool2(x, bias, B, C, H, W, io_bf16, y, static_cast<hipStream_t>(stream));
}

int ipsr_cat_relu_forward(const void* y, const void* x, int B, int C1, int C2, int HW, int io_bf16, void* out, void* stream)
{
    if (!x || !out) return fail(IPSR_ERR_INVALID, "ipsr_cat_relu_forward: null pointer");
    if (B < 1 || C1 < 1 || C2 < 1 || HW < 1) return fail(IPSR_ERR_INVALID, "ipsr_cat_relu_forward: bad size");
    if ((HW & 3) == 0 && (!aligned_io(y, io_bf16) || !aligned_io(x, io_bf16) || !aligned_io(out, io_bf16)))
        return fail(IPSR_ERR_INVALID, "ipsr_cat_relu_forward: tensors are not vector aligned");
    return launch_cat_relu_fwd(y, x, B, C1, C2, HW, io_bf16, out, static_cast<hipStream_t>(stream));
}

int ipsr_cat_relu_backward(const void* grad_out, const void* out, int B, int C1, int C2, int HW, int io_bf16, void* dy, void* dx, void* stream)
{
    if (!grad_out || !out || !dx) return fail(IPSR_ERR_INVALID, "ipsr_cat_relu_backward: null pointer");
    if (B < 1 || C1 < 1 || C2 < 1 || HW < 1) return fail(IPSR_ERR_INVALID, "ipsr_cat_relu_backward: bad size");
    if ((HW & 3) == 0 && (!aligned_io(grad_out, io_bf16) || !aligned_io(out, io_bf16) || !aligned_io(dy, io_bf16) || !aligned_io(dx, io_bf16)))
        return fail(IPSR_ERR_INVALID, "ipsr_cat_relu_backward: tensors are not vector aligned");
    return launch_cat_relu_bwd(grad_out, out, B, C1, C2, HW, io_bf16, dy, dx, static_cast<hipStream_t>(stream));
}

int ipsr_instnorm_act_forward(const void* x, const float* bias, const float* gamma, const float* beta, float eps, int act, float slope,
                              int B, int C, int HW, int io_bf16, void* y, float* mean, float* rstd, unsigned* tickets, void* stream)
{
    if (!x || !y || !mean || !rstd) return fail(IPSR_ERR_INVALID, "ipsr_instnorm_act_forward: null pointer");
    if (B < 1 || C < 1 || HW < 2 || act < 0 || act > 2) return fail(IPSR_ERR_INVALID, "ipsr_instnorm_act_forward: bad argument B=%d C=%d HW=%d act=%d", B, C, HW, act);
    if ((HW & 3) == 0 && (!aligned_io(x, io_bf16) || !aligned_io(y, io_bf16))) return fail(IPSR_ERR_INVALID, "ipsr_instnorm_act_forward: x/y are not vector aligned");
    return launch_instnorm_act_fwd(x, bias, gamma, beta, eps, act, slope, B, C, HW, io_bf16, y, mean, rstd, 0, nullptr, 0, tickets, static_cast<hipStream_t>(stream));
}

int ipsr_instnorm_act_forward_slice(const void* x, const float* bias, const float* gamma, const float* beta, float eps, int act, float slope,
                                    int B, int C, int HW, int io_bf16, void* y, size_t y_batch_stride, void* y2, size_t y2_batch_stride,
                                    float* mean, float* rstd, unsigned* tickets, void* stream)
{
    if (!x || !y || !mean || !rstd) return fail(IPSR_ERR_INVALID, "ipsr_instnorm_act_forward_slice: null pointer");
    if (B < 1 || C < 1 || HW < 2 || act < 0 || act > 2 || y_batch_stride < (size_t)C * HW || (y2 && y2_batch_stride < (size_t)C * HW))
        return fail(IPSR_ERR_INVALID, "ipsr_instnorm_act_forward_slice: bad argument B=%d C=%d HW=%d act=%d stride=%zu", B, C, HW, act, y_batch_stride);
    if ((HW & 3) == 0 && (!aligned_io(x, io_bf16) || !aligned_io(y, io_bf16) || (y_batch_stride & 3) || !aligned_io(y2, io_bf16) || (y2_batch_stride & 3)))
        return fail(IPSR_ERR_INVALID, "ipsr_instnorm_act_forward_slice: x/y are not vector aligned");
    return launch_instnorm_act_fwd(x, bias, gamma, beta, eps, act, slope, B, C, HW, io_bf16, y, mean, rstd, y_batch_stride, y2, y2_batch_stride, tickets,
                                   static_cast<hipStream_t>(stream));
}

int ipsr_instnorm_act_backward(const void* dy, const void* y, const void* x, const float* bias, const float* gamma,
                               const float* mean, const float* rstd, int act, float slope, int B, int C, int HW, int io_bf16,
                               void* dx, float* dgamma_p, float* dbeta_p, float* dbias_p, float* sums, unsigned* tickets, void* stream)
{
    if (!dy || !y || !x || !mean || !rstd || !dx) return fail(IPSR_ERR_INVALID, "ipsr_instnorm_act_backward: null pointer");
    if (B < 1 || C < 1 || HW < 2 || act < 0 || act > 2) return fail(IPSR_ERR_INVALID, "ipsr_instnorm_act_backward: bad argument");
    if ((HW & 3) == 0 && (!aligned_io(dy, io_bf16) || !aligned_io(y, io_bf16) || !aligned_io(x, io_bf16) || !aligned_io(dx, io_bf16)))
        return fail(IPSR_ERR_INVALID, "ipsr_instnorm_act_backward: tensors are not vector aligned");
    return launch_instnorm_act_bwd(dy, y, x, bias, gamma, mean, rstd, act, slope, B, C, HW, io_bf16, dx, dgamma_p, dbeta_p, dbias_p, sums, tickets, 0, 0, nullptr, 0,
                                   static_cast<hipStream_t>(stream));
}

int ipsr_instnorm_act_backward_slice(const void* dy, size_t dy_batch_stride, const void* dy2, size_t dy2_batch_stride, const void* y,
                                     size_t y_batch_stride, const void* x, const float* bias,
                                     const float* gamma, const float* mean, const float* rstd, int act, float slope, int B, int C, int HW,
                                     int io_bf16, void* dx, float* dgamma_p, float* dbeta_p, float* dbias_p, float* sums, unsigned* tickets, void* stream)
{
    if (!dy || !y || !x || !mean || !rstd || !dx) return fail(IPSR_ERR_INVALID, "ipsr_instnorm_act_backward_slice: null pointer");
    if (B < 1 || C < 1 || HW < 2 || act < 0 || act > 2 || dy_batch_stride < (size_t)C * HW || y_batch_stride < (size_t)C * HW ||
        (dy2 && dy2_batch_stride < (size_t)C * HW))
        return fail(IPSR_ERR_INVALID, "ipsr_instnorm_act_backward_slice: bad argument");
    if ((HW & 3) == 0 && (!aligned_io(dy, io_bf16) || !aligned_io(y, io_bf16) || !aligned_io(x, io_bf16) || !aligned_io(dx, io_bf16) ||
                          !aligned_io(dy2, io_bf16) || ((dy_batch_stride | y_batch_stride | dy2_batch_stride) & 3)))
        return fail(IPSR_ERR_INVALID, "ipsr_instnorm_act_backward_slice: tensors are not vector aligned");
    return launch_instnorm_act_bwd(dy, y, x, bias, gamma, mean, rstd, act, slope, B, C, HW, io_bf16, dx, dgamma_p, dbeta_p, dbias_p, sums, tickets,
                                   dy_batch_stride, y_batch_stride, dy2, dy2_batch_stride, static_cast<hipStream_t>(stream));
}

int ipsr_bias_act_backward(const void* dy, const void* y, int act, float slope, int B, int C, int HW, int io_bf16, void* dx,
                           float* dbias_p, float* sums, unsigned* tickets, void* stream)
{
    if (!dy || !y || !dx) return fail(IPSR_ERR_INVALID, "ipsr_bias_act_backward: null pointer");
    if (B < 1 || C < 1 || HW < 1 || act < 0 || act > 2) return fail(IPSR_ERR_INVALID, "ipsr_bias_act_backward: bad argument");
    if ((HW & 3) == 0 && (!aligned_io(dy, io_bf16) || !aligned_io(y, io_bf16) || !aligned_io(dx, io_bf16)))
        return fail(IPSR_ERR_INVALID, "ipsr_bias_act_backward: tensors are not vector aligned");
    return launch_bias_act_bwd(dy, y, act, slope, B, C, HW, io_bf16, dx, dbias_p, sums, tickets, nullptr, 0, static_cast<hipStream_t>(stream));
}

int ipsr_bias_act_backward_skip(const void* dy, const void* dy2, size_t dy2_batch_stride, const void* y, int act, float slope, int B, int C, int HW,
                                int io_bf16, void* dx, float* dbias_p, float* sums, unsigned* tickets, void* stream)
{
    if (!dy || !dy2 || !y || !dx) return fail(IPSR_ERR_INVALID, "ipsr_bias_act_backward_skip: null pointer");
    if (B < 1 || C < 1 || HW < 1 || act < 0 || act > 2 || dy2_batch_stride < (size_t)C * HW) return fail(IPSR_ERR_INVALID, "ipsr_bias_act_backward_skip: bad argument");
    if ((HW & 3) == 0 && (!aligned_io(dy, io_bf16) || !aligned_io(dy2, io_bf16) || !aligned_io(y, io_bf16) || !aligned_io(dx, io_bf16) || (dy2_batch_stride & 3)))
        return fail(IPSR_ERR_INVALID, "ipsr_bias_act_backward_skip: tensors are not vector aligned");
    return launch_bias_act_bwd(dy, y, act, slope, B, C, HW, io_bf16, dx, dbias_p, sums, tickets, dy2, dy2_batch_stride, static_cast<hipStream_t>(stream));
}

size_t innercos_workspace_bytes(int B, int Cuse, int N) { return innercos_ws_bytes(B, Cuse, N); }

int innercos_loss(const float* x, int B, int Cx, int Cuse, int N, const float* mask, const float* target,
                  float strength, float* loss, void* ws, size_t ws_bytes, void* stream)
{
    if (!x || !mask || !target || !loss || !ws) return fail(IPSR_ERR_INVALID, "innercos_loss: null pointer");
    if (B < 1 || Cuse < 1 || Cx < Cuse || N < 1) return fail(IPSR_ERR_INVALID, "innercos_loss: bad size B=%d Cx=%d Cuse=%d N=%d", B, Cx, Cuse, N);
    if (!aligned16(x) || !aligned16(target) || !aligned16(mask)) return fail(IPSR_ERR_INVALID, "innercos_loss: x/target/mask must be 16-byte aligned");
    return launch_innercos_loss(x, B, Cx, Cuse, N, mask, target, strength, loss, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

int innercos_loss_fused(const float* x, int B, int Cx, int Cuse, int N, const float* mask, const float* target,
                        float strength, float* loss, void* ws, size_t ws_bytes, unsigned* ticket, void* stream)
{
    if (!x || !mask || !target || !loss || !ws || !ticket) return fail(IPSR_ERR_INVALID, "innercos_loss_fused: null pointer");
    if (B < 1 || Cuse < 1 || Cx < Cuse || N < 1) return fail(IPSR_ERR_INVALID, "innercos_loss_fused: bad size B=%d Cx=%d Cuse=%d N=%d", B, Cx, Cuse, N);
    if (!aligned16(x) || !aligned16(target) || !aligned16(mask)) return fail(IPSR_ERR_INVALID, "innercos_loss_fused: x/target/mask must be 16-byte aligned");
    return launch_innercos_loss_fused(x, B, Cx, Cuse, N, mask, target, strength, loss, ws, ws_bytes, ticket, static_cast<hipStream_t>(stream));
}

int innercos_loss_backward(const float* x, int B, int Cx, int Cuse, int N, const float* mask, const float* target,
                           float strength, const float* grad_loss, float* grad_x, void* stream)
{
    if (!x || !mask || !target || !grad_loss || !grad_x) return fail(IPSR_ERR_INVALID, "innercos_loss_backward: null pointer");
    if (B < 1 || Cuse < 1 || Cx < Cuse || N < 1) return fail(IPSR_ERR_INVALID, "innercos_loss_backward: bad size");
    return launch_innercos_backward(x, B, Cx, Cuse, N, mask, target, strength, grad_loss, grad_x, static_cast<hipStream_t>(stream));
}

}  // extern "C"
