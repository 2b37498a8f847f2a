// mask_ops.hip — K1 feature-mask pyramid and K2 index prep.
//
// Reference: util.cal_feat_mask (util/util.py:68-84) and util.cal_mask_given_mask_thred
// (util/util.py:88-161).  Both depend on the mask only, so the host caches their result per mask;
// they are byte/integer kernels far below any roofline (a 256x256 mask is 64 KB).
#include "ipsr_common.h"

namespace ipsr {

static inline int half_dim(int n) { return (n + 2 - 4) / 2 + 1; }

// One level of the pyramid: 4x4 box sum, stride 2, zero padding 1.  The reference does this in fp32
// with weights 1/16; the sums are count/16^level and exact, so integer counting gives the same bits.
// LAST: compare count * 16^-layers with the threshold and emit bytes.
template <typename TIn, bool LAST>
__global__ void __launch_bounds__(256) boxsum4s2_kernel(const TIn* __restrict__ in, int h, int w, int oh, int ow,
                                                        uint32_t* __restrict__ out_cnt, uint8_t* __restrict__ out_u8,
                                                        float scale, float threshold)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= oh * ow) return;
    const int y = idx / ow, x = idx - y * ow;
    uint32_t s = 0;
#pragma unroll
    for (int dy = 0; dy < 4; ++dy) {
        const int yy = 2 * y - 1 + dy;
        if (yy < 0 || yy >= h) continue;
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            const int xx = 2 * x - 1 + dx;
            if (xx < 0 || xx >= w) continue;
            const TIn v = in[(size_t)yy * w + xx];
            s += sizeof(TIn) == 1 ? (v ? 1u : 0u) : (uint32_t)v;
        }
    }
    if (LAST) out_u8[idx] = ((float)s * scale > threshold) ? 1 : 0;
    else out_cnt[idx] = s;
}

int launch_feat_mask(const uint8_t* mask, int H, int W, int layers, float threshold, uint8_t* feat,
                     void* ws, size_t ws_bytes, hipStream_t st)
{
    const int h1 = half_dim(H), w1 = half_dim(W);
    const size_t lvl = align_up((size_t)h1 * w1 * sizeof(uint32_t), 256);
    if (layers > 1 && ws_bytes < 2 * lvl) return fail(IPSR_ERR_WORKSPACE, "ipsr_feat_mask: workspace %zu < %zu", ws_bytes, 2 * lvl);
    uint32_t* buf[2] = {reinterpret_cast<uint32_t*>(ws), reinterpret_cast<uint32_t*>(static_cast<char*>(ws) + lvl)};
    float scale = 1.0f;
    for (int l = 0; l < layers; ++l) scale *= (1.0f / 16.0f);
    int h = H, w = W;
    const void* cur = mask;
    for (int l = 0; l < layers; ++l) {
        const int oh = half_dim(h), ow = half_dim(w);
        if (oh < 1 || ow < 1) return fail(IPSR_ERR_INVALID, "ipsr_feat_mask: mask %dx%d too small for %d layers", H, W, layers);
        const bool last = (l == layers - 1);
        const int grid = cdiv(oh * ow, 256);
        uint32_t* dst = buf[l & 1];
        if (l == 0) {
            if (last) boxsum4s2_kernel<uint8_t, true><<<grid, 256, 0, st>>>((const uint8_t*)cur, h, w, oh, ow, nullptr, feat, scale, threshold);
            else boxsum4s2_kernel<uint8_t, false><<<grid, 256, 0, st>>>((const uint8_t*)cur, h, w, oh, ow, dst, nullptr, scale, threshold);
        } else {
            if (last) boxsum4s2_kernel<uint32_t, true><<<grid, 256, 0, st>>>((const uint32_t*)cur, h, w, oh, ow, nullptr, feat, scale, threshold);
            else boxsum4s2_kernel<uint32_t, false><<<grid, 256, 0, st>>>((const uint32_t*)cur, h, w, oh, ow, dst, nullptr, scale, threshold);
        }
        if (int rc = check_launch("boxsum4s2_kernel")) return rc;
        cur = dst;
        h = oh; w = ow;
    }
    return IPSR_OK;
}

// K2: raster scan -> flag[N], ordered compaction of the flagged positions.  One workgroup walks the N
// positions in chunks of 1024; order is kept by a ballot/popcount prefix inside each wave plus an LDS
// prefix over the 16 waves.
__global__ void __launch_bounds__(1024) index_prep_kernel(const uint8_t* __restrict__ feat, int h, int w, int patch,
                                                          int stride, int mask_thred, int nW, int N,
                                                          int32_t* __restrict__ flag, int32_t* __restrict__ mpi,
                                                          int32_t* __restrict__ count)
{
    __shared__ int wave_tot[16];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int i0 = 0; i0 < N; i0 += 1024) {
        const int i = i0 + tid;
        int f = 0;
        if (i < N) {
            const int py = i / nW, px = i - py * nW;
            int s = 0;
            for (int dy = 0; dy < patch; ++dy)
                for (int dx = 0; dx < patch; ++dx) s += feat[(size_t)(py * stride + dy) * w + px * stride + dx];
            f = (s >= mask_thred) ? 1 : 0;
            flag[i] = f;
        }
        const unsigned long long bal = __ballot(f);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wv] = __popcll(bal);
        __syncthreads();
        int off = base_s;
        for (int j = 0; j < wv; ++j) off += wave_tot[j];
        if (f) mpi[off + before] = i;
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int j = 0; j < 16; ++j) t += wave_tot[j];
            base_s += t;
        }
        __syncthreads();
    }
    const int M = base_s;
    for (int i = M + tid; i < N; i += 1024) mpi[i] = -1;
    if (tid == 0) *count = M;
}

int launch_index_prep(const uint8_t* feat, int h, int w, int patch, int stride, int mask_thred,
                      int32_t* flag, int32_t* mask_point_idx, int32_t* count, hipStream_t st)
{
    const int nH = (h - patch) / stride + 1, nW = (w - patch) / stride + 1;
    index_prep_kernel<<<1, 1024, 0, st>>>(feat, h, w, patch, stride, mask_thred, nW, nH * nW, flag, mask_point_idx, count);
    return check_launch("index_prep_kernel");
}

}  // namespace ipsr
