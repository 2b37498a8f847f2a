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
// Weight gradient of the thin layers on the bf16 matrix cores (BASELINE config 5; bf16 wide tensor):
//      G[kb][cs][r][s] = sum_{b, y, x}  big[b][kb][y][x] * small[b][cs][y*st + r - pad][x*st + s - pad]
// Conv2d: big = dy, small = x;  ConvTranspose2d: big = x, small = dy — the same formula with `big` on the coarse grid — and G is the
// module's weight layout [dim0 = wide][dim1 = narrow][k][k] either way.  k3 s1 p1 and k4 s2 p1, 3 or 6 narrow channels.
// As a GEMM the reduction runs over PIXELS: D[kb][R] += A[kb][16 px] * B[16 px][R], R = cs * k*k + tap (27 / 54 / 48 / 96 -> one to three
// 32-column tiles).  A fragments are aligned 16-byte loads of the wide tensor (8 consecutive pixels of one channel: NCHW has them contiguous),
// B fragments are gathered from the narrow tensor, which lives in L1 / L2 (786 KB per sample): 8 scalar loads per lane and 32 columns,
// zero outside the image, rounded to bf16 like autocast's cast.  No LDS.  A wave owns whole rows of `big` and the full [kb tile][R]
// accumulator; the workgroup's four waves are added in order through LDS into one partial slab, a second launch adds the slabs in order
// (deterministic).
// The vector-ALU form above takes 0.14-0.30 ms per layer at batch 16 (VALU-bound); this one reads the wide tensor once at HBM rate.
typedef __bf16 thin_bf16x8 __attribute__((ext_vector_type(8)));
typedef float thin_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned thin_u32x4 __attribute__((ext_vector_type(4)));

// TB = float: the same reduction in the reference's own arithmetic on v_mfma_f32_32x32x2_f32 (fp32 operands, eight matrix instructions per
// 16 pixels: lane half h holds pixels 8h .. 8h+7 of the step, instruction i multiplies pixels i and 8 + i) — the fp32 training step's thin
// weight gradients (MIOpen: 0.14-0.23 ms each at batch 8).
template <int MT, int RT, typename TB, typename TS>
__global__ void __launch_bounds__(256) thin_wrw_mfma_kernel(const TB* __restrict__ big, const TS* __restrict__ small, float* __restrict__ slabs,
                                                            int B, int Kb, int Cs, int Hb, int Wb, int Hs, int Ws, int k, int st, int pad, int rows_per_wg)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 31, h = lane >> 5;
    const int b = blockIdx.y, k0 = blockIdx.z * (32 * MT);
    const int y_lo = blockIdx.x * rows_per_wg;
    const int T = k * k, R_real = Cs * T;
    // the lane's column of each 32-column tile: (narrow channel, tap row, tap column), or nothing beyond the real width
    int c_cs[RT], c_rr[RT], c_ss[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int r = n + 32 * rt;
        const bool live = r < R_real;
        const int cs = live ? r / T : 0, t = live ? r - cs * T : 0;
        c_cs[rt] = live ? cs : -1; c_rr[rt] = t / k; c_ss[rt] = t - (t / k) * k;
    }
    thin_f32x16 acc[MT][RT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][rt][e] = 0.0f;

    const size_t plane_b = (size_t)Hb * Wb, plane_s = (size_t)Hs * Ws;
    // The wave's steps: its rows (y_lo + wave, + 4, ...) x the row's 16-pixel segments, as one sequence, software-pipelined: the loads of
    // step i + 1 (2-4 wide-tensor vectors from HBM, 8-24 narrow-tensor values from L2) are issued before step i is converted and multiplied
    // — one memory latency per step would otherwise be the whole kernel (16 steps per wave).
    const int xsteps = Wb >> 4;
    int nrows = 0;
    for (int y = y_lo + wave; y < y_lo + rows_per_wg && y < Hb; y += 4) ++nrows;
    const int nsteps = nrows * xsteps;
    constexpr bool F32 = !thin_is_bf16<TB>::value;
    constexpr int NA = F32 ? 2 : 1;                            // 16-byte vectors of `big` per lane and step: 8 pixels
    thin_u32x4 la[2][MT][NA];
    float lg[2][RT][8];
    auto issue = [&](int i, int buf) {
        const int y = y_lo + wave + 4 * (i / xsteps), x0 = (i % xsteps) << 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int kb = min(k0 + 32 * mt + n, Kb - 1);                  // rows beyond Kb are masked when consumed
            const thin_u32x4* src = reinterpret_cast<const thin_u32x4*>(big + ((size_t)b * Kb + kb) * plane_b + (size_t)y * Wb + x0 + 8 * h);
#pragma unroll
            for (int v = 0; v < NA; ++v) la[buf][mt][v] = src[v];
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int sy = min(max(y * st + c_rr[rt] - pad, 0), Hs - 1);
            const TS* sp = small + ((size_t)b * Cs + max(c_cs[rt], 0)) * plane_s + (size_t)sy * Ws;
            // every load is issued, from a clamped address, and masked when consumed: a load under a per-element condition makes hipcc
            // branch around it and wait for it alone
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int sx = (x0 + 8 * h + j) * st + c_ss[rt] - pad;
                lg[buf][rt][j] = ld1(sp, (size_t)min(max(sx, 0), Ws - 1));
            }
        }
    };
    auto consume = [&](int i, int buf) {
        const int y = y_lo + wave + 4 * (i / xsteps), x0 = (i % xsteps) << 4;
        float q[RT][8];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int sy = y * st + c_rr[rt] - pad;
            const bool rowok = c_cs[rt] >= 0 && (unsigned)sy < (unsigned)Hs;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int sx = (x0 + 8 * h + j) * st + c_ss[rt] - pad;
                q[rt][j] = (rowok && (unsigned)sx < (unsigned)Ws) ? lg[buf][rt][j] : 0.0f;
            }
        }
        const thin_u32x4 z = {0u, 0u, 0u, 0u};
        if constexpr (F32) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const bool live = k0 + 32 * mt + n < Kb;
                const thin_u32x4 v0 = live ? la[buf][mt][0] : z, v1 = live ? la[buf][mt][NA - 1] : z;
                const float a8[8] = {__uint_as_float(v0.x), __uint_as_float(v0.y), __uint_as_float(v0.z), __uint_as_float(v0.w),
                                     __uint_as_float(v1.x), __uint_as_float(v1.y), __uint_as_float(v1.z), __uint_as_float(v1.w)};
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[mt][rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a8[i], q[rt][i], acc[mt][rt], 0, 0, 0);
            }
        } else {
            thin_bf16x8 fa[MT], fb[RT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) fa[mt] = __builtin_bit_cast(thin_bf16x8, (k0 + 32 * mt + n < Kb) ? la[buf][mt][0] : z);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const thin_u32x4 pk = {f2bf2(q[rt][0], q[rt][1]), f2bf2(q[rt][2], q[rt][3]), f2bf2(q[rt][4], q[rt][5]), f2bf2(q[rt][6], q[rt][7])};      // (exact for a bf16 narrow tensor)
                fb[rt] = __builtin_bit_cast(thin_bf16x8, pk);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[mt][rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt], fb[rt], acc[mt][rt], 0, 0, 0);
        }
    };
    if (nsteps > 0) issue(0, 0);
    for (int i = 0; i < nsteps; i += 2) {                      // two steps per trip: the buffer index is a compile-time constant
        if (i + 1 < nsteps) issue(i + 1, 1);
        consume(i, 0);
        if (i + 1 < nsteps) {
            if (i + 2 < nsteps) issue(i + 2, 0);
            consume(i + 1, 1);
        }
    }
    // the four waves' accumulators are added in wave order through LDS (one wave's worth at a time), wave 0 writes the workgroup's slab
    // [kb tile row][32 RT columns]
    __shared__ float red[MT * RT * 16 * 64];
    for (int w = 1; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) red[((mt * RT + rt) * 16 + e) * 64 + lane] = acc[mt][rt][e];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[mt][rt][e] += red[((mt * RT + rt) * 16 + e) * 64 + lane];
        }
        __syncthreads();
    }
    if (wave != 0) return;
    const size_t slab = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    float* out = slabs + slab * (32 * MT) * (32 * RT);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = 32 * mt + (e & 3) + 8 * (e >> 2) + 4 * h;
                out[(size_t)row * (32 * RT) + 32 * rt + n] = acc[mt][rt][e];
            }
}

// G[kb][r] = sum over the slabs of one kb tile, ascending (r < R_real).  One workgroup per kb row: thread = (r, slab group of four).
__global__ void __launch_bounds__(256) thin_wrw_mfma_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ G, int Kb, int R_real, int rows_tile,
                                                                   int cols, int slabs_per_tile)
{
    __shared__ float part[4][64];
    const int kb = blockIdx.x, tile = kb / rows_tile, row = kb - tile * rows_tile;
    const int grp = threadIdx.x >> 6;
    const size_t sstride = (size_t)rows_tile * cols;
    for (int r0 = 0; r0 < cols; r0 += 64) {
        const int r = r0 + (threadIdx.x & 63);
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;      // four independent chains per thread (a fixed order all the same): the loads overlap
        if (r < cols) {
            const float* p = slabs + (size_t)tile * slabs_per_tile * sstride + (size_t)row * cols + r;
            int sI = grp;
            for (; sI + 12 < slabs_per_tile; sI += 16) {
                a0 += p[(size_t)sI * sstride]; a1 += p[(size_t)(sI + 4) * sstride]; a2 += p[(size_t)(sI + 8) * sstride]; a3 += p[(size_t)(sI + 12) * sstride];
            }
            for (; sI < slabs_per_tile; sI += 4) a0 += p[(size_t)sI * sstride];
        }
        part[grp][threadIdx.x & 63] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (grp == 0 && r < R_real) G[(size_t)kb * R_real + r] = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// few -> many on the bf16 matrix cores (BASELINE config 5):
//      out[b][o][y][x] = sum_{cs < Cs, t < k*k}  Wg(o, cs, t) * in[b][cs][y*st + r - pad][x*st + s - pad]       (+ bias, ReLU)
// Cs in {3, 6}, (k, st) = (3, 1) or (4, 2), pad 1: VGG conv1_1, netG's first Conv2d, the first Conv2d of netP / netD, and the input
// gradient of netG's last ConvTranspose2d (Wg = W[o*so + cs*si + (flip ? T-1-t : t)] as in the vector-ALU kernels above, which are VALU-bound
// at ~40 TF: 90-180 us per layer at batch 16).  GEMM D[o][32 px] = A[o][R] * B[R][32 px] with the reduction R = Cs*k*k (27 / 54 / 48 / 96)
// padded to KS steps of 16: the A fragments (the whole weight matrix of the workgroup's o tile) live in registers for the workgroup's
// lifetime, a B fragment is the lane's pixel's window gathered from the narrow tensor (L1 / L2 resident; every load issued from a
// clamped address and masked afterwards), software-pipelined one tile ahead.  The 32 x 32MT result tile leaves through a per-wave LDS
// staging area as 16-byte row pieces.  A wave owns a 32-pixel row segment at a time; workgroup = 4 waves = 4 rows.
template <int MT, int KS, typename TIN, typename TOUT>
__global__ void __launch_bounds__(256) thin_f2m_mfma_kernel(const TIN* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias, int relu,
                                                            TOUT* __restrict__ out, int B, int Cs, int O, int Hs, int Ws, int Ho, int Wo, int k, int st, int pad,
                                                            long so, long si, int flip, int rows_per_wg)
{
    constexpr int EPW = 32 * MT * 32;                          // result elements per wave tile
    __shared__ __attribute__((aligned(16))) TOUT stage[4][EPW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 31, h = lane >> 5;
    const int b = blockIdx.y, o0 = blockIdx.z * (32 * MT);
    const int T = k * k, R_real = Cs * T;
    // A: row o0 + 32 mt + n, reduction entries 16 ks + 8 h + j
    thin_bf16x8 fa[MT][KS];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            unsigned short q[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = 16 * ks + 8 * h + j, o = o0 + 32 * mt + n;
                const int cs = r / T, t = r - cs * T;
                const bool live = r < R_real && o < O;
                const float v = w[live ? (long)o * so + (long)cs * si + (flip ? T - 1 - t : t) : 0];
                q[j] = f2bf(live ? v : 0.0f);
            }
            const thin_u32x4 pk = {(unsigned)q[0] | ((unsigned)q[1] << 16), (unsigned)q[2] | ((unsigned)q[3] << 16),
                                   (unsigned)q[4] | ((unsigned)q[5] << 16), (unsigned)q[6] | ((unsigned)q[7] << 16)};
            fa[mt][ks] = __builtin_bit_cast(thin_bf16x8, pk);
        }
    // B: the lane's window entries of step ks, element j, packed as (narrow channel | tap row << 4 | tap column << 8 | live << 12) once — the
    // divisions by the runtime tap count do not belong in the tile loop
    int code[KS][8];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = 16 * ks + 8 * h + j;
            const bool live = r < R_real;
            const int cs = live ? r / T : 0, t = live ? r - cs * T : 0, rr = t / k, ss = t - rr * k;
            code[ks][j] = cs | (rr << 4) | (ss << 8) | ((live ? 1 : 0) << 12);
        }
    const size_t plane_s = (size_t)Hs * Ws, plane_o = (size_t)Ho * Wo;
    int off[KS][8];                                            // the same entries as address offsets from the pixel's own position (interior tiles)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int cd = code[ks][j];
            off[ks][j] = (cd & 15) * (int)plane_s + (((cd >> 4) & 15) - pad) * Ws + (((cd >> 8) & 15) - pad);
        }
    float bv[MT][16];                                          // bias of the lane's result rows
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int o = o0 + 32 * mt + (e & 3) + 8 * (e >> 2) + 4 * h;
            bv[mt][e] = (bias && o < O) ? bias[o] : 0.0f;
        }
    const int xsegs = Wo >> 5;
    int nrows = 0;
    const int y_lo = blockIdx.x * rows_per_wg;
    for (int y = y_lo + wave; y < y_lo + rows_per_wg && y < Ho; y += 4) ++nrows;
    const int ntiles = nrows * xsegs;
    float lg[2][KS][8];
    // a tile whose windows all lie inside the image (wave-uniform): plain base + offset loads and no masks
    auto interior = [&](int y, int x0) {
        return y * st - pad >= 0 && y * st + (k - 1) - pad < Hs && x0 * st - pad >= 0 && (x0 + 31) * st + (k - 1) - pad < Ws;
    };
    auto issue = [&](int i, int buf) {
        const int y = y_lo + wave + 4 * (i / xsegs), x0 = (i % xsegs) << 5, x = x0 + n;
        if (interior(y, x0)) {
            const TIN* base = in + (size_t)b * Cs * plane_s + (size_t)(y * st) * Ws + x * st;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) lg[buf][ks][j] = ld1(base + off[ks][j], 0);
            return;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int cd = code[ks][j], cs = cd & 15, rr = (cd >> 4) & 15, ss = (cd >> 8) & 15;
                const int sy = min(max(y * st + rr - pad, 0), Hs - 1), sx = min(max(x * st + ss - pad, 0), Ws - 1);
                lg[buf][ks][j] = ld1(in + ((size_t)b * Cs + cs) * plane_s, (size_t)sy * Ws + sx);
            }
    };
    auto consume = [&](int i, int buf) {
        const int y = y_lo + wave + 4 * (i / xsegs), x0 = (i % xsegs) << 5, x = x0 + n;
        thin_f32x16 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][e] = bv[mt][e];
        const bool inner = interior(y, x0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            float q[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int cd = code[ks][j], rr = (cd >> 4) & 15, ss = (cd >> 8) & 15;
                const int sy = y * st + rr - pad, sx = x * st + ss - pad;
                // (entries beyond the real reduction length meet zero weights; interior tiles need no mask at all)
                const bool ok = inner || ((unsigned)sy < (unsigned)Hs && (unsigned)sx < (unsigned)Ws);
                q[j] = ok ? lg[buf][ks][j] : 0.0f;
            }
            const thin_u32x4 pk = {f2bf2(q[0], q[1]), f2bf2(q[2], q[3]), f2bf2(q[4], q[5]), f2bf2(q[6], q[7])};
            const thin_bf16x8 fb = __builtin_bit_cast(thin_bf16x8, pk);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][ks], fb, acc[mt], 0, 0, 0);
        }
        // result tile [row o][32 px] through the wave's staging area, out as 16-byte pieces of rows
        TOUT* L = stage[wave];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[mt][e];
                if (relu) v = thin_relu(v);
                st1(L, (size_t)(32 * mt + (e & 3) + 8 * (e >> 2) + 4 * h) * 32 + n, v);
            }
        constexpr int PER = 16 / sizeof(TOUT);                 // elements per 16-byte piece: 8 (bf16) or 4 (fp32)
        constexpr int PPR = 32 / PER;                          // pieces per row
#pragma unroll
        for (int c0 = 0; c0 < 32 * MT * PPR; c0 += 64) {
            const int c = c0 + lane, row = c / PPR, seg = c - row * PPR;
            const int o = o0 + row;
            const uint4 v = *reinterpret_cast<const uint4*>(L + (size_t)row * 32 + seg * PER);
            if (o < O) *reinterpret_cast<uint4*>(out + ((size_t)b * O + o) * plane_o + (size_t)y * Wo + x0 + seg * PER) = v;
        }
    };
    if (ntiles > 0) issue(0, 0);
    for (int i = 0; i < ntiles; i += 2) {
        if (i + 1 < ntiles) issue(i + 1, 1);
        consume(i, 0);
        if (i + 1 < ntiles) {
            if (i + 2 < ntiles) issue(i + 2, 0);
            consume(i + 1, 1);
        }
    }
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

// ---- few -> many on the matrix cores --------------------------------------------------------------------------------------------
static int thin_f2m_mfma_plan(int B, int Cs, int O, int Ho, int Wo, int k, int stride, int* MT, int* KS, int* rows)
{
    const bool k3 = k == 3 && stride == 1, k4 = k == 4 && stride == 2;
    if (B < 1 || O < 1 || Ho < 1 || (Cs != 3 && Cs != 6) || !(k3 || k4) || Wo % 32 != 0 || O % 8 != 0) return 0;
    *KS = (Cs * k * k + 15) / 16;                             // 2, 4, 3, 6
    *MT = O >= 128 && *KS <= 4 ? 4 : 2;
    const int otiles = (O + 32 * *MT - 1) / (32 * *MT);
    int r = 4;
    while ((long)B * otiles * ((Ho + r - 1) / r) > 2048 && r < Ho) r *= 2;
    *rows = r;
    return otiles;
}

int ipsr_conv_thin_f2m_mfma_supported(int B, int Cs, int O, int Ho, int Wo, int k, int stride)
{
    int MT, KS, rows;
    return thin_f2m_mfma_plan(B, Cs, O, Ho, Wo, k, stride, &MT, &KS, &rows) > 0;
}

// in [B,Cs,Ho*stride,Wo*stride] (io bit 0: bf16, else fp32), out [B,O,Ho,Wo] (io bit 1: bf16, else fp32); weight element (o, cs, t) at
// w[o*so + cs*si + (flip ? k*k-1-t : t)]; bias [O] or NULL, relu 0/1.
int ipsr_conv_thin_f2m_mfma(const void* in, const float* w, const float* bias, int relu, void* out, int B, int Cs, int O, int Ho, int Wo, int k, int stride,
                            long so, long si, int flip, int io, void* stream)
{
    if (!in || !w || !out || (io & ~3)) return fail(IPSR_ERR_INVALID, "ipsr_conv_thin_f2m_mfma: null pointer / bad io code");
    int MT, KS, rows;
    const int otiles = thin_f2m_mfma_plan(B, Cs, O, Ho, Wo, k, stride, &MT, &KS, &rows);
    if (!otiles) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_conv_thin_f2m_mfma: Cs=%d O=%d %dx%d k%d s%d is not implemented", Cs, O, Ho, Wo, k, stride);
    if (reinterpret_cast<uintptr_t>(out) & 15u) return fail(IPSR_ERR_INVALID, "ipsr_conv_thin_f2m_mfma: `out` must be 16-byte aligned");
    if (B > 65535 || otiles > 65535) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_conv_thin_f2m_mfma: grid too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((Ho + rows - 1) / rows, B, otiles);
    const int Hs = Ho * stride, Ws = Wo * stride;
    const bool ib = io & 1, ob = io & 2;
#define THIN_FM(MTT, KSS, TI, TO) thin_f2m_mfma_kernel<MTT, KSS, TI, TO><<<grid, 256, 0, st>>>(static_cast<const TI*>(in), w, bias, relu, static_cast<TO*>(out), \
                                                                                       B, Cs, O, Hs, Ws, Ho, Wo, k, stride, 1, so, si, flip, rows)
#define THIN_FM_IO(MTT, KSS) do { if (ib && ob) THIN_FM(MTT, KSS, bf16_t, bf16_t); else if (ib) THIN_FM(MTT, KSS, bf16_t, float); \
                                  else if (ob) THIN_FM(MTT, KSS, float, bf16_t); else THIN_FM(MTT, KSS, float, float); } while (0)
    if (MT == 4) {
        if (KS == 2) THIN_FM_IO(4, 2); else if (KS == 3) THIN_FM_IO(4, 3); else THIN_FM_IO(4, 4);
    } else {
        if (KS == 2) THIN_FM_IO(2, 2); else if (KS == 3) THIN_FM_IO(2, 3); else if (KS == 4) THIN_FM_IO(2, 4); else THIN_FM_IO(2, 6);
    }
#undef THIN_FM_IO
#undef THIN_FM
    return check_launch("thin_f2m_mfma_kernel");
}

// ---- thin weight gradients on the matrix cores (bf16 wide tensor) ---------------------------------------------------------------
static int thin_wrw_mfma_plan(int B, int Kb, int Cs, int Hb, int Wb, int k, int stride, int* MT, int* RT, int* rows_per_wg, int* gx)
{
    const bool k3 = k == 3 && stride == 1, k4 = k == 4 && stride == 2;
    if (B < 1 || Kb < 1 || Hb < 1 || (Cs != 3 && Cs != 6) || !(k3 || k4) || Wb % 16 != 0 || Wb < 16) return 0;
    const int R = Cs * k * k;                                 // 27, 54, 48, 96
    *RT = (R + 31) / 32;
    *MT = Kb >= 128 && *RT <= 2 ? 4 : 2;                      // accumulator tiles per wave: at most 8
    if (Kb % 8 != 0) return 0;
    // rows per workgroup: two to four workgroups per CU
    const int ktiles = (Kb + 32 * *MT - 1) / (32 * *MT);
    int rows = 4;
    const long wgs = *RT >= 2 ? 512 : 1024;                   // (measured: two gather tiles per step like longer runs, one tile more workgroups)
    while ((long)B * ktiles * ((Hb + rows - 1) / rows) > wgs && rows < Hb) rows *= 2;
    *rows_per_wg = rows;
    *gx = (Hb + rows - 1) / rows;
    return ktiles;
}

size_t ipsr_conv_thin_wrw_mfma_workspace_bytes(int B, int Kb, int Cs, int Hb, int Wb, int k, int stride)
{
    int MT, RT, rows, gx;
    const int ktiles = thin_wrw_mfma_plan(B, Kb, Cs, Hb, Wb, k, stride, &MT, &RT, &rows, &gx);
    if (!ktiles) return 0;
    return align_up((size_t)ktiles * B * gx * (32 * MT) * (32 * RT) * sizeof(float), 256) + 256;
}

// big [B,Kb,Hb,Wb], small [B,Cs,Hs,Ws], Hs = Hb (k3 s1 p1) or 2 Hb (k4 s2 p1); g [Kb][Cs][k][k] fp32.  io: bit 0 = `big` is bf16, bit 1 =
// `small` is bf16.  A bf16 `big` multiplies on the bf16 matrix cores (an fp32 `small` is rounded to bf16); an fp32 `big` (with an fp32
// `small`) multiplies in fp32 on v_mfma_f32_32x32x2_f32 — the reference's arithmetic.
int ipsr_conv_thin_wrw_mfma(const void* big, const void* small, float* g, int B, int Kb, int Cs, int Hb, int Wb, int k, int stride, int io,
                            void* ws, size_t ws_bytes, void* stream)
{
    if (!big || !small || !g || !ws || (io & ~3)) return fail(IPSR_ERR_INVALID, "ipsr_conv_thin_wrw_mfma: null pointer / bad io code");
    if (io == 2) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_conv_thin_wrw_mfma: an fp32 wide tensor needs an fp32 narrow one");
    const int small_bf16 = (io >> 1) & 1, big_bf16 = io & 1;
    int MT, RT, rows, gx;
    const int ktiles = thin_wrw_mfma_plan(B, Kb, Cs, Hb, Wb, k, stride, &MT, &RT, &rows, &gx);
    if (!ktiles) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_conv_thin_wrw_mfma: Kb=%d Cs=%d %dx%d k%d s%d is not implemented", Kb, Cs, Hb, Wb, k, stride);
    const size_t need = ipsr_conv_thin_wrw_mfma_workspace_bytes(B, Kb, Cs, Hb, Wb, k, stride);
    if (ws_bytes < need) return fail(IPSR_ERR_WORKSPACE, "ipsr_conv_thin_wrw_mfma: workspace %zu < %zu", ws_bytes, need);
    if ((reinterpret_cast<uintptr_t>(big) & 15u) || (reinterpret_cast<uintptr_t>(ws) & 15u)) return fail(IPSR_ERR_INVALID, "ipsr_conv_thin_wrw_mfma: `big` / workspace must be 16-byte aligned");
    if (B > 65535 || ktiles > 65535) return fail(IPSR_ERR_UNSUPPORTED, "ipsr_conv_thin_wrw_mfma: grid too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* slabs = static_cast<float*>(ws);
    const int Hs = Hb * stride, Ws = Wb * stride, pad = 1;
    const dim3 grid(gx, B, ktiles);
#define THIN_WM(MTT, RTT, TBG, TSM) thin_wrw_mfma_kernel<MTT, RTT, TBG, TSM><<<grid, 256, 0, st>>>(static_cast<const TBG*>(big), static_cast<const TSM*>(small), slabs, \
                                                                                            B, Kb, Cs, Hb, Wb, Hs, Ws, k, stride, pad, rows)
#define THIN_WM_TS(MTT, RTT) do { if (!big_bf16) THIN_WM(MTT, RTT, float, float); else if (small_bf16) THIN_WM(MTT, RTT, bf16_t, bf16_t); \
                                  else THIN_WM(MTT, RTT, bf16_t, float); } while (0)
    if (MT == 4 && RT == 1) THIN_WM_TS(4, 1);
    else if (MT == 4 && RT == 2) THIN_WM_TS(4, 2);
    else if (RT == 1) THIN_WM_TS(2, 1);
    else if (RT == 2) THIN_WM_TS(2, 2);
    else THIN_WM_TS(2, 3);
#undef THIN_WM_TS
#undef THIN_WM
    if (int rc = check_launch("thin_wrw_mfma_kernel")) return rc;
    thin_wrw_mfma_reduce_kernel<<<ktiles * 32 * MT > Kb ? Kb : ktiles * 32 * MT, 256, 0, st>>>(slabs, g, Kb, Cs * k * k, 32 * MT, 32 * RT, B * gx);
    return check_launch("thin_wrw_mfma_reduce_kernel");
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
