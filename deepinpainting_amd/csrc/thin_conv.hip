// thin_conv.hip — the 3x3 stride-1 pad-1 layers with 3 or 6 channels on one side, at full resolution:
//   VGG16 conv1_1 (3 -> 64 on 256x256, models/vgg16.py:9; forward x3 per step, input gradient x1),
//   netG's first convolution (6 -> 64, models/networks.py:300-312) and its last ConvTranspose2d (128 -> 3, :255-259).
// These are not matrix-core work: 27-54 multiply-adds per output against 134-268 MB of activations on the wide side, i.e. HBM
// streams.  MIOpen runs them through its generic paths (65-230 us each, NHWC transposes included); here each is ONE pass over
// the wide tensor on the vector ALUs with the narrow tensor and the weights served from L1 / LDS:
//   few -> many   out[b][o][y][x] = sum_{i<I<=8} sum_t Wg(o,i,t) in[b][i][y+r-1][x+s-1]   (+ bias, ReLU): writes the wide tensor once
//   many -> few   the same sum with O <= 8 outputs and I wide: reads the wide tensor once (4 pixels per lane, float4 rows)
//   weight grad   G[cb][cs][u][v] = sum_{b,y,x} big[b][cb][y][x] * small[b][cs][y+u-1][x+v-1]: reads the wide tensor once
// Wg(o,i,t) = W[o*so + i*si + (flip ? 8 - t : t)] expresses Conv2d / ConvTranspose2d, forward / backward-data (as in winograd.hip):
//   Conv2d forward       W[co][ci]: o = co, i = ci, so = Ci*9, si = 9,    flip 0      Conv2d backward-data   o = ci, i = co, so = 9,    si = Ci*9, flip 1
//   ConvT  forward       W[ci][co]: o = co, i = ci, so = 9,    si = Co*9, flip 1      ConvT  backward-data   o = ci, i = co, so = Co*9, si = 9,    flip 0
// and the weight gradient of either is G with big = the wide-channel tensor of the pair (x, dy), small = the other one:
//   Conv2d dW[co][ci] = G(big = dy, small = x)[co][ci];  ConvTranspose2d dW[ci][co] = G(big = x, small = dy)[ci][co].
#include "ipsr_common.h"

namespace ipsr {

constexpr int THIN_OC = 16;          // output channels per workgroup (few -> many)
constexpr int THIN_CB = 2;           // wide channels per workgroup (weight gradient)
constexpr int THIN_ROWS = 64;        // rows per workgroup (weight gradient)

__device__ __forceinline__ float thin_relu(float v) { return v < 0.0f ? 0.0f : v; }      // NaN stays NaN (torch.relu)

// few -> many.  One thread = TWO adjacent pixels x 16 output channels: the I x 3 x 4 window values live in registers, the 16*I*9 weights in
// LDS (uniform reads: broadcast) — every weight read feeds two multiply-adds (with one pixel per thread the kernel was bound by its 432
// ds_reads per thread, 2 TB/s of output).  grid (ceil(W/512), H, B * O/16); W even.
// TIN / TOUT = float or bf16_t.  With a bf16 side (BASELINE config 5) the kernel is the autocast convolution: weights and an fp32 input are
// rounded to bf16 on the way in (what autocast's casts do), products accumulate in fp32, bias / ReLU in fp32, one rounding on the way out.
template <typename T> struct thin_is_bf16 { static constexpr bool value = false; };
template <> struct thin_is_bf16<bf16_t> { static constexpr bool value = true; };
__device__ __forceinline__ float thin_rb(float v) { return bf2f(f2bf(v)); }
__device__ __forceinline__ void thin_st2(float* p, float a, float b) { *reinterpret_cast<float2*>(p) = make_float2(a, b); }
__device__ __forceinline__ void thin_st2(bf16_t* p, float a, float b) { *reinterpret_cast<unsigned*>(p) = (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16); }

template <int I, typename TIN, typename TOUT>
__global__ void __launch_bounds__(256) thin_f2m_kernel(const TIN* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias, int relu,
                                                       TOUT* __restrict__ out, int B, int O, int H, int W, long so, long si, int flip)
{
    constexpr bool RB = thin_is_bf16<TIN>::value || thin_is_bf16<TOUT>::value;
    __shared__ float wl[THIN_OC][I * 9];
    const int nchunk = O / THIN_OC;
    const int b = blockIdx.z / nchunk, o0 = (blockIdx.z - b * nchunk) * THIN_OC;
    for (int idx = threadIdx.x; idx < THIN_OC * I * 9; idx += 256) {
        const int o = idx / (I * 9), rem = idx - o * (I * 9), i = rem / 9, t = rem - i * 9;
        const float wv = w[(long)(o0 + o) * so + (long)i * si + (flip ? 8 - t : t)];
        wl[o][rem] = RB ? thin_rb(wv) : wv;
    }
    __syncthreads();
    const int x = (blockIdx.x * 256 + threadIdx.x) * 2, y = blockIdx.y;
    if (x >= W) return;
    float win[I][3][4];                                   // columns x-1 .. x+2
#pragma unroll
    for (int i = 0; i < I; ++i) {
        const TIN* ip = in + ((size_t)b * I + i) * H * W;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int yy = y + r - 1;
            const bool yok = (unsigned)yy < (unsigned)H;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int xx = x + c - 1;
                const float v = (yok && (unsigned)xx < (unsigned)W) ? ld1(ip, (size_t)yy * W + xx) : 0.0f;
                win[i][r][c] = (RB && !thin_is_bf16<TIN>::value) ? thin_rb(v) : v;
            }
        }
    }
    TOUT* op = out + (((size_t)b * O + o0) * H + y) * W + x;
#pragma unroll 4
    for (int o = 0; o < THIN_OC; ++o) {
        float a0 = bias ? bias[o0 + o] : 0.0f, a1 = a0;
#pragma unroll
        for (int i = 0; i < I; ++i)
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const float wv = wl[o][i * 9 + r * 3 + q];
                    a0 = __builtin_fmaf(wv, win[i][r][q], a0);
                    a1 = __builtin_fmaf(wv, win[i][r][q + 1], a1);
                }
        if (relu) { a0 = thin_relu(a0); a1 = thin_relu(a1); }
        thin_st2(op + (size_t)o * H * W, a0, a1);
    }
}

// many -> few.  One thread = 4 adjacent pixels x all O outputs; per input channel three float4 rows (+ the two halo columns),
// the O*9 weights of the channel from LDS.  grid (ceil(W/1024 * 4 rows) ...): block = 4 rows x 64 lanes x 4 pixels.
template <int O, typename TIN, typename TOUT>
__global__ void __launch_bounds__(256) thin_m2f_kernel(const TIN* __restrict__ in, const float* __restrict__ w, TOUT* __restrict__ out,
                                                       int B, int I, int H, int W, long so, long si, int flip)
{
    constexpr bool RB = thin_is_bf16<TIN>::value || thin_is_bf16<TOUT>::value;
    extern __shared__ float wl[];                 // [I][O*9]
    for (int idx = threadIdx.x; idx < I * O * 9; idx += 256) {
        const int i = idx / (O * 9), rem = idx - i * (O * 9), o = rem / 9, t = rem - o * 9;
        const float wv = w[(long)o * so + (long)i * si + (flip ? 8 - t : t)];
        wl[idx] = RB ? thin_rb(wv) : wv;
    }
    __syncthreads();
    const int b = blockIdx.z;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int x0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    if (y >= H || x0 >= W) return;
    float acc[O][4];
#pragma unroll
    for (int o = 0; o < O; ++o)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[o][p] = 0.0f;
    const TIN* ib = in + (size_t)b * I * H * W;
    for (int i = 0; i < I; ++i) {
        const TIN* ip = ib + (size_t)i * H * W;
        const float* wi = wl + i * (O * 9);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int yy = y + r - 1;
            float v[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            if ((unsigned)yy < (unsigned)H) {
                const size_t e = (size_t)yy * W + x0;                // a multiple of 4 (W % 4 == 0): ld4 indexes 4-element vectors
                const float4 c = ld4(ip, e >> 2);
                v[1] = c.x; v[2] = c.y; v[3] = c.z; v[4] = c.w;
                if (x0 > 0) v[0] = ld1(ip, e - 1);
                if (x0 + 4 < W) v[5] = ld1(ip, e + 4);
                if (RB && !thin_is_bf16<TIN>::value) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) v[q] = thin_rb(v[q]);
                }
            }
#pragma unroll
            for (int o = 0; o < O; ++o)
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const float wv = wi[o * 9 + r * 3 + s];
#pragma unroll
                    for (int p = 0; p < 4; ++p) acc[o][p] = __builtin_fmaf(wv, v[p + s], acc[o][p]);
                }
        }
    }
#pragma unroll
    for (int o = 0; o < O; ++o)
        st4(out, ((((size_t)b * O + o) * H + y) * W + x0) >> 2, make_float4(acc[o][0], acc[o][1], acc[o][2], acc[o][3]));
}

// weight gradient, pass 1.  One workgroup = THIN_CB wide channels x THIN_ROWS rows of one sample; a thread walks its 4 pixels of each
// row (4 rows at a time), keeps THIN_CB*CS*9 partial sums, the workgroup reduces them through LDS (fixed order) and writes one
// partial vector.  grid (ceil(W/256), ceil(H/THIN_ROWS), B * Cb/THIN_CB).
template <int CS, typename TB, typename TS>
__global__ void __launch_bounds__(256) thin_wrw_kernel(const TB* __restrict__ big, const TS* __restrict__ small, float* __restrict__ part,
                                                       int B, int Cb, int H, int W)
{
    constexpr int NV = THIN_CB * CS * 9;
    __shared__ float red[4][NV];                                   // one row per wave
    const int ncb = Cb / THIN_CB;
    const int b = blockIdx.z / ncb, cb0 = (blockIdx.z - b * ncb) * THIN_CB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x0 = (blockIdx.x * 64 + lane) * 4;
    float acc[THIN_CB][CS][9];
#pragma unroll
    for (int c = 0; c < THIN_CB; ++c)
#pragma unroll
        for (int k = 0; k < CS; ++k)
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[c][k][t] = 0.0f;
    if (x0 < W) {
        for (int yr = 0; yr < THIN_ROWS; yr += 4) {
            const int y = blockIdx.y * THIN_ROWS + yr + wave;
            if (y >= H) break;
            float g[THIN_CB][4];
#pragma unroll
            for (int c = 0; c < THIN_CB; ++c) {
                const float4 v = ld4(big, ((((size_t)b * Cb + cb0 + c) * H + y) * W + x0) >> 2);
                g[c][0] = v.x; g[c][1] = v.y; g[c][2] = v.z; g[c][3] = v.w;
                if (thin_is_bf16<TS>::value && !thin_is_bf16<TB>::value) {       // mixed operands: the fp32 one is rounded as autocast's cast would
#pragma unroll
                    for (int p = 0; p < 4; ++p) g[c][p] = thin_rb(g[c][p]);
                }
            }
#pragma unroll
            for (int k = 0; k < CS; ++k) {
                const TS* sp = small + ((size_t)b * CS + k) * H * W;
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int yy = y + u - 1;
                    float v[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                    if ((unsigned)yy < (unsigned)H) {
                        const size_t e = (size_t)yy * W + x0;
                        const float4 c4 = ld4(sp, e >> 2);
                        v[1] = c4.x; v[2] = c4.y; v[3] = c4.z; v[4] = c4.w;
                        if (x0 > 0) v[0] = ld1(sp, e - 1);
                        if (x0 + 4 < W) v[5] = ld1(sp, e + 4);
                        if (thin_is_bf16<TB>::value && !thin_is_bf16<TS>::value) {
#pragma unroll
                            for (int q = 0; q < 6; ++q) v[q] = thin_rb(v[q]);
                        }
                    }
#pragma unroll
                    for (int c = 0; c < THIN_CB; ++c)
#pragma unroll
                        for (int s = 0; s < 3; ++s)
#pragma unroll
                            for (int p = 0; p < 4; ++p) acc[c][k][u * 3 + s] = __builtin_fmaf(g[c][p], v[p + s], acc[c][k][u * 3 + s]);
                }
            }
        }
    }
    // wave reduction (fixed butterfly), then the four waves in order
#pragma unroll
    for (int c = 0; c < THIN_CB; ++c)
#pragma unroll
        for (int k = 0; k < CS; ++k)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                float v = acc[c][k][t];
#pragma unroll
                for (int sft = 32; sft >= 1; sft >>= 1) v += __shfl_xor(v, sft);
                if (lane == 0) red[wave][(c * CS + k) * 9 + t] = v;
            }
    __syncthreads();
    if (threadIdx.x < NV) {
        const float v = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
        const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        part[blk * NV + threadIdx.x] = v;
    }
}

// pass 2: G[cb][cs][t] = sum over (sample, row block, column block) of the partial vectors, ascending.
__global__ void __launch_bounds__(256) thin_wrw_reduce_kernel(const float* __restrict__ part, float* __restrict__ G, int B, int Cb, int CS, int nblk_yx)
{
    const int NV = THIN_CB * CS * 9;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Cb * CS * 9) return;
    const int cb = idx / (CS * 9), rem = idx - cb * (CS * 9);
    const int grp = cb / THIN_CB, c = cb - grp * THIN_CB;
    const int ncb = Cb / THIN_CB;
    float acc = 0.0f;
    for (int b = 0; b < B; ++b) {
        const float* p = part + ((size_t)(b * ncb + grp) * nblk_yx) * NV + c * (CS * 9) + rem;
        for (int j = 0; j < nblk_yx; ++j) acc += p[(size_t)j * NV];
    }
    G[idx] = acc;
}


// ---------------------------------------------------------------------------------------------------
// Conv2d with ONE output channel, stride 1 (netD's last layer, 512 -> 1, k4 p1 on 31x31: models/networks.py:489-495; batch 16 in
// backward_D, 8 in backward_G).  252 MFLOP against 31.5 MB of input: a stream, which MIOpen's generic paths take 80-145 us
// (forward) and 75-127 us (weight gradient) for.  Both passes here walk input planes through LDS (zero halo of `pad`, so the taps
// need no bounds checks), a thread owning 4 adjacent output pixels of one row (Ho * ceil(Wo/4) <= 256 pixel groups):
//   forward   workgroup = (sample, chunk of ONE_CCH channels): y_part[b][chunk][oy][ox] = sum over the chunk's channels (ascending)
//             of sum_{r,s<K} w[c][r][s] x[b][c][oy+r-P][ox+s-P]; a second tiny launch adds the chunks in order
//   weight    workgroup = one channel: dw[c][r][s] = sum_b sum_{oy,ox} dy[b][oy][ox] x[b][c][oy+r-P][ox+s-P], samples ascending,
//   gradient  then the lanes (butterfly) and the four waves in order
// (the input gradient, 1 -> 512, stays on MIOpen: 38 us).  Fixed summation orders: deterministic.
constexpr int ONE_CCH = 8;             // channels per workgroup: 64 chunks x B workgroups keep several per CU in flight (each is a chain of plane loads)

template <int K>
__global__ void __launch_bounds__(256) one_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ part,
                                                      int C, int H, int W, int Ho, int Wo, int P, int nchunk)
{
    extern __shared__ float sm[];                    // two padded planes [(H+2P)][(W+2P)], then the chunk's weights [ONE_CCH][K*K]
    const int Hp = H + 2 * P, Wp = W + 2 * P, psz = Hp * Wp;
    float* plane[2] = {sm, sm + psz};
    float* wl = sm + 2 * psz;
    const int chunk = blockIdx.x, b = blockIdx.y, c0 = chunk * ONE_CCH;
    const int nc = min(ONE_CCH, C - c0);
    for (int i = threadIdx.x; i < 2 * psz; i += 256) sm[i] = 0.0f;          // the halo stays zero, the interior is overwritten
    for (int i = threadIdx.x; i < nc * K * K; i += 256) wl[i] = w[(size_t)c0 * K * K + i];
    const int ngx = (Wo + 3) / 4;
    const int oy = threadIdx.x / ngx, ox0 = (threadIdx.x - oy * ngx) * 4;
    const bool live = oy < Ho;
    const float* xb = x + ((size_t)b * C + c0) * H * W;
    __syncthreads();
    auto stage = [&](int ci, float* dst) {
        const float* xp = xb + (size_t)ci * H * W;
        for (int i = threadIdx.x; i < H * W; i += 256) { const int yy = i / W, xx = i - yy * W; dst[(yy + P) * Wp + xx + P] = xp[i]; }
    };
    stage(0, plane[0]);
    __syncthreads();
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int ci = 0; ci < nc; ++ci) {
        const float* pl = plane[ci & 1];
        if (ci + 1 < nc) stage(ci + 1, plane[(ci + 1) & 1]);
        if (live) {
            const float* wc = wl + ci * (K * K);
#pragma unroll
            for (int r = 0; r < K; ++r) {
                float v[K + 3];
                const float* row = pl + (oy + r) * Wp + ox0;
#pragma unroll
                for (int j = 0; j < K + 3; ++j) v[j] = ox0 + j < Wp ? row[j] : 0.0f;
#pragma unroll
                for (int q = 0; q < K; ++q) {
                    const float wv = wc[r * K + q];
#pragma unroll
                    for (int p = 0; p < 4; ++p) acc[p] = __builtin_fmaf(wv, v[p + q], acc[p]);
                }
            }
        }
        __syncthreads();
    }
    if (live) {
        float* dst = part + (((size_t)b * nchunk + chunk) * Ho + oy) * Wo + ox0;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (ox0 + p < Wo) dst[p] = acc[p];
    }
}

__global__ void __launch_bounds__(256) one_fwd_sum_kernel(const float* __restrict__ part, float* __restrict__ y, int n_out, int hw, int nchunk)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_out) return;
    const int b = i / hw, p = i - b * hw;
    const float* src = part + (size_t)b * nchunk * hw + p;
    float t = src[0];
    for (int j = 1; j < nchunk; ++j) t += src[(size_t)j * hw];
    y[i] = t;
}

template <int K>
__global__ void __launch_bounds__(256) one_wrw_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw,
                                                      int B, int C, int H, int W, int Ho, int Wo, int P)
{
    extern __shared__ float sm[];                    // two padded planes of x[b][c], two planes of dy[b], then the wave sums
    const int Hp = H + 2 * P, Wp = W + 2 * P, psz = Hp * Wp, gsz = Ho * Wo;
    float* plane[2] = {sm, sm + psz};
    float* gpl[2] = {sm + 2 * psz, sm + 2 * psz + gsz};
    float* red = sm + 2 * psz + 2 * gsz;             // [4][K*K]
    const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2 * psz; i += 256) sm[i] = 0.0f;
    const int ngx = (Wo + 3) / 4;
    const int oy = threadIdx.x / ngx, ox0 = (threadIdx.x - oy * ngx) * 4;
    const bool live = oy < Ho;
    __syncthreads();
    auto stage = [&](int b, int slot) {
        const float* xp = x + ((size_t)b * C + c) * H * W;
        for (int i = threadIdx.x; i < H * W; i += 256) { const int yy = i / W, xx = i - yy * W; plane[slot][(yy + P) * Wp + xx + P] = xp[i]; }
        const float* gp = dy + (size_t)b * gsz;
        for (int i = threadIdx.x; i < gsz; i += 256) gpl[slot][i] = gp[i];
    };
    stage(0, 0);
    __syncthreads();
    float acc[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) acc[t] = 0.0f;
    for (int b = 0; b < B; ++b) {
        const float* pl = plane[b & 1];
        const float* gl = gpl[b & 1];
        if (b + 1 < B) stage(b + 1, (b + 1) & 1);
        if (live) {
            float g[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) g[p] = ox0 + p < Wo ? gl[oy * Wo + ox0 + p] : 0.0f;
#pragma unroll
            for (int r = 0; r < K; ++r) {
                float v[K + 3];
                const float* row = pl + (oy + r) * Wp + ox0;
#pragma unroll
                for (int j = 0; j < K + 3; ++j) v[j] = ox0 + j < Wp ? row[j] : 0.0f;
#pragma unroll
                for (int q = 0; q < K; ++q)
#pragma unroll
                    for (int p = 0; p < 4; ++p) acc[r * K + q] = __builtin_fmaf(g[p], v[p + q], acc[r * K + q]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < K * K; ++t) {
        float v = acc[t];
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) v += __shfl_xor(v, sft);
        if (lane == 0) red[wave * (K * K) + t] = v;
    }
    __syncthreads();
    if (threadIdx.x < K * K)
        dw[(size_t)c * K * K + threadIdx.x] = ((red[threadIdx.x] + red[K * K + threadIdx.x]) + red[2 * K * K + threadIdx.x]) + red[3 * K * K + threadIdx.x];
}

}  // namespace ipsr

using namespace ipsr;

extern "C" {

// io: bit 0 = `in` is bf16, bit 1 = `out` is bf16 (as in ipsr_conv3x3_winograd_mp); 0 = the fp32 kernels of round 2.
int ipsr_conv3x3_thin_io(int op, const void* in, const float* w, const float* bias, int relu, void* out, int B, int I, int O, int H, int W,
                         long so, long si, int flip, int io, void* stream)
{
    if (!in || !w || !out) return fail(IPSR_ERR_INVALID, "ipsr_conv3x3_thin: null pointer");
    if (B < 1 || I < 1 || O < 1 || H < 1 || W < 1 || (io & ~3)) return fail(IPSR_ERR_INVALID, "ipsr_conv3x3_thin: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool ib = io & 1, ob = io & 2;
    if (op == 0) {          // few -> many
        if (O % THIN_OC != 0 || (I != 3 && I != 6)) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_conv3x3_thin: few->many needs 3 or 6 inputs and outputs %% 16 == 0 (got %d -> %d)", I, O);
        if ((size_t)B * (O / THIN_OC) > 65535 || H > 65535) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_conv3x3_thin: grid too large");
        if ((W & 1) || (reinterpret_cast<uintptr_t>(out) & (ob ? 3u : 7u))) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_conv3x3_thin: few->many needs an even width and an aligned output (W=%d)", W);
        const dim3 grid(cdiv(W, 512), H, B * (O / THIN_OC));
#define THIN_F2M(II, TI, TO) thin_f2m_kernel<II, TI, TO><<<grid, 256, 0, st>>>(static_cast<const TI*>(in), w, bias, relu, static_cast<TO*>(out), B, O, H, W, so, si, flip)
#define THIN_F2M_IO(II) do { if (ib && ob) THIN_F2M(II, bf16_t, bf16_t); else if (ib) THIN_F2M(II, bf16_t, float); else if (ob) THIN_F2M(II, float, bf16_t); else THIN_F2M(II, float, float); } while (0)
        if (I == 3) THIN_F2M_IO(3); else THIN_F2M_IO(6);
#undef THIN_F2M_IO
#undef THIN_F2M
        return check_launch("thin_f2m_kernel");
    }
    if (op == 1) {          // many -> few
        if (bias || relu) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_conv3x3_thin: many->few has no epilogue");
        if ((O != 3 && O != 6) || W % 4 != 0) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_conv3x3_thin: many->few needs 3 or 6 outputs and W %% 4 == 0 (got %d -> %d, W=%d)", I, O, W);
        if ((reinterpret_cast<uintptr_t>(in) & (ib ? 7u : 15u)) || (reinterpret_cast<uintptr_t>(out) & (ob ? 7u : 15u)))
            return fail(IPSR_ERR_INVALID, "ipsr_conv3x3_thin: tensors must be aligned to four elements");
        const size_t lds = (size_t)I * O * 9 * sizeof(float);
        if (lds > 48 * 1024) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_conv3x3_thin: %d input channels exceed the LDS weight buffer", I);
        const dim3 grid(cdiv(W, 256), cdiv(H, 4), B);
#define THIN_M2F(OO, TI, TO) thin_m2f_kernel<OO, TI, TO><<<grid, 256, lds, st>>>(static_cast<const TI*>(in), w, static_cast<TO*>(out), B, I, H, W, so, si, flip)
#define THIN_M2F_IO(OO) do { if (ib && ob) THIN_M2F(OO, bf16_t, bf16_t); else if (ib) THIN_M2F(OO, bf16_t, float); else if (ob) THIN_M2F(OO, float, bf16_t); else THIN_M2F(OO, float, float); } while (0)
        if (O == 3) THIN_M2F_IO(3); else THIN_M2F_IO(6);
#undef THIN_M2F_IO
#undef THIN_M2F
        return check_launch("thin_m2f_kernel");
    }
    return fail(IPSR_ERR_INVALID, "ipsr_conv3x3_thin: op %d", op);
}

int ipsr_conv3x3_thin(int op, const float* in, const float* w, const float* bias, int relu, float* out, int B, int I, int O, int H, int W,
                      long so, long si, int flip, void* stream)
{
    return ipsr_conv3x3_thin_io(op, in, w, bias, relu, out, B, I, O, H, W, so, si, flip, 0, stream);
}

size_t ipsr_conv3x3_thin_wrw_workspace_bytes(int B, int Cb, int Cs, int H, int W)
{
    if (B < 1 || Cb < 1 || H < 1 || W < 1 || (Cs != 3 && Cs != 6) || Cb % THIN_CB != 0 || W % 4 != 0) return 0;
    const size_t nblk = (size_t)B * (Cb / THIN_CB) * cdiv(H, THIN_ROWS) * cdiv(W, 256);
    return align_up(nblk * THIN_CB * Cs * 9 * sizeof(float), 256) + 256;
}

// io: bit 0 = `big` is bf16, bit 1 = `small` is bf16; the gradient is fp32 either way.
int ipsr_conv3x3_thin_wrw_io(const void* big, const void* small, float* g, int B, int Cb, int Cs, int H, int W, int io, void* ws, size_t ws_bytes, void* stream)
{
    if (!big || !small || !g || !ws || (io & ~3)) return fail(IPSR_ERR_INVALID, "ipsr_conv3x3_thin_wrw: null pointer / bad io code");
    const size_t need = ipsr_conv3x3_thin_wrw_workspace_bytes(B, Cb, Cs, H, W);
    if (need == 0) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_conv3x3_thin_wrw: Cb=%d Cs=%d %dx%d is not implemented", Cb, Cs, H, W);
    if (ws_bytes < need) return fail(IPSR_ERR_WORKSPACE, "ipsr_conv3x3_thin_wrw: workspace %zu < %zu", ws_bytes, need);
    const bool bb = io & 1, sb = io & 2;
    if ((reinterpret_cast<uintptr_t>(big) & (bb ? 7u : 15u)) || (reinterpret_cast<uintptr_t>(small) & (sb ? 7u : 15u)) || (reinterpret_cast<uintptr_t>(ws) & 15u))
        return fail(IPSR_ERR_INVALID, "ipsr_conv3x3_thin_wrw: tensors must be aligned to four elements, the workspace to 16 bytes");
    if ((size_t)B * (Cb / THIN_CB) > 65535) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_conv3x3_thin_wrw: grid too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(ws);
    const dim3 grid(cdiv(W, 256), cdiv(H, THIN_ROWS), B * (Cb / THIN_CB));
#define THIN_WRW(CC, TBG, TSM) thin_wrw_kernel<CC, TBG, TSM><<<grid, 256, 0, st>>>(static_cast<const TBG*>(big), static_cast<const TSM*>(small), part, B, Cb, H, W)
#define THIN_WRW_IO(CC) do { if (bb && sb) THIN_WRW(CC, bf16_t, bf16_t); else if (bb) THIN_WRW(CC, bf16_t, float); else if (sb) THIN_WRW(CC, float, bf16_t); else THIN_WRW(CC, float, float); } while (0)
    if (Cs == 3) THIN_WRW_IO(3); else THIN_WRW_IO(6);
#undef THIN_WRW_IO
#undef THIN_WRW
    if (int rc = check_launch("thin_wrw_kernel")) return rc;
    thin_wrw_reduce_kernel<<<cdiv(Cb * Cs * 9, 256), 256, 0, st>>>(part, g, B, Cb, Cs, (int)(grid.x * grid.y));
    return check_launch("thin_wrw_reduce_kernel");
}

int ipsr_conv3x3_thin_wrw(const float* big, const float* small, float* g, int B, int Cb, int Cs, int H, int W, void* ws, size_t ws_bytes, void* stream)
{
    return ipsr_conv3x3_thin_wrw_io(big, small, g, B, Cb, Cs, H, W, 0, ws, ws_bytes, stream);
}

size_t ipsr_conv_to_one_workspace_bytes(int B, int C, int H, int W, int K, int pad)
{
    const int Ho = H + 2 * pad - K + 1, Wo = W + 2 * pad - K + 1;
    if (B < 1 || C < 1 || pad < 0 || Ho < 1 || Wo < 1 || (K != 3 && K != 4) || Ho * cdiv(Wo, 4) > 256) return 0;
    return align_up((size_t)B * cdiv(C, ONE_CCH) * Ho * Wo * sizeof(float), 256) + 256;
}

int ipsr_conv_to_one(int op, const float* x, const float* other, float* out, int B, int C, int H, int W, int K, int pad,
                     void* ws, size_t ws_bytes, void* stream)
{
    if (!x || !other || !out) return fail(IPSR_ERR_INVALID, "ipsr_conv_to_one: null pointer");
    const int Ho = H + 2 * pad - K + 1, Wo = W + 2 * pad - K + 1;
    const size_t need = ipsr_conv_to_one_workspace_bytes(B, C, H, W, K, pad);
    if (need == 0) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_conv_to_one: %dx%d k=%d p=%d is not implemented (k in {3,4}, at most 256 groups of 4 output pixels)", H, W, K, pad);
    if (B > 65535) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_conv_to_one: batch %d", B);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t psz = (size_t)(H + 2 * pad) * (W + 2 * pad);
    if (op == 0) {                                    // forward: other = w [1][C][K][K], out = y [B][1][Ho][Wo]
        if (!ws || ws_bytes < need) return fail(IPSR_ERR_WORKSPACE, "ipsr_conv_to_one: workspace %zu < %zu", ws_bytes, need);
        const int nchunk = cdiv(C, ONE_CCH);
        const size_t lds = (2 * psz + (size_t)ONE_CCH * K * K) * sizeof(float);
        if (lds > 64 * 1024) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_conv_to_one: a %dx%d plane exceeds the LDS staging buffers", H, W);
        float* part = static_cast<float*>(ws);
        const dim3 grid(nchunk, B);
        if (K == 4) one_fwd_kernel<4><<<grid, 256, lds, st>>>(x, other, part, C, H, W, Ho, Wo, pad, nchunk);
        else one_fwd_kernel<3><<<grid, 256, lds, st>>>(x, other, part, C, H, W, Ho, Wo, pad, nchunk);
        if (int rc = check_launch("one_fwd_kernel")) return rc;
        one_fwd_sum_kernel<<<cdiv(B * Ho * Wo, 256), 256, 0, st>>>(part, out, B * Ho * Wo, Ho * Wo, nchunk);
        return check_launch("one_fwd_sum_kernel");
    }
    if (op == 2) {                                    // weight gradient: other = dy [B][1][Ho][Wo], out = dw [1][C][K][K]
        const size_t lds = (2 * psz + 2 * (size_t)Ho * Wo + 4 * K * K) * sizeof(float);
        if (lds > 64 * 1024) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_conv_to_one: a %dx%d plane exceeds the LDS staging buffers", H, W);
        if (K == 4) one_wrw_kernel<4><<<C, 256, lds, st>>>(x, other, out, B, C, H, W, Ho, Wo, pad);
        else one_wrw_kernel<3><<<C, 256, lds, st>>>(x, other, out, B, C, H, W, Ho, Wo, pad);
        return check_launch("one_wrw_kernel");
    }
    return fail(IPSR_ERR_INVALID, "ipsr_conv_to_one: op %d (0 forward, 2 weight gradient)", op);
}

}  // extern "C"
