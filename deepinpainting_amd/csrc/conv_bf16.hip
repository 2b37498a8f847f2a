// conv_bf16.hip — direct bf16 implicit-GEMM convolutions on v_mfma_f32_32x32x16_bf16, NCHW in and out, no layout passes.
//
// BASELINE config 5 ("256x256 bf16 mixed precision: CDNA4 bf16 MFMA for patch-corr + convs") runs the conv stacks of the four nets
// and of VGG16 (models/networks.py:220-259, 404-432, 470-495, 510-515; models/vgg16.py:9-21) on bf16 activations.  The split-bf16
// Winograd engines (winograd.hip) only win on the >= 512-channel <= 32x32 layers: their transformed operands stay fp32-wide
// (4.5x the bf16 activation bytes each way).  Everything else was MIOpen's (NCHW<->NHWC transposes + ~450 TF implicit GEMMs).
// This file is the direct form for those layers: bf16 operands, fp32 accumulation, ONE launch per pass.
//
//      out[b][k][oy][ox] = sum_{c, t} Wp[k][c][t] * in[b][c][oy + dy_t][ox + dx_t]            (zero outside the image)
//
// as a GEMM  M = produced channels k,  N = pixels,  reduction = (c, t).  Forward and input gradient of Conv2d / ConvTranspose2d
// (k3 s1 p1) are all this form: only the weight re-packing differs (index strides, tap flip), as in conv_gemm.hip / winograd.hip.
//
// The MFMA wants, per lane, 8 CONSECUTIVE reduction elements: 8 channels of ONE pixel — but NCHW has a channel's pixels contiguous.
//   * weights: re-packed once per call (cast to bf16 anyway) into the image the LDS wants, [k tile][c block 16][tap][c group 2][128 k][8 c]:
//     one stage's A tile is one contiguous block, copied by LDS-DMA, fragments by ds_read_b128;
//   * activations: a stage's tile (16 channels x the tile's rows + halo rows, full image width) comes in by LDS-DMA in its natural
//     [c][row][x] form, then crosses LDS once: ds_read_b64_tr_b16 reads 4 channels x 16 pixels and hands every lane 4 channels of ONE
//     pixel, which it stores to the POSITION-major image T[c group][row][x + halo][8 c].  From T a B fragment (8 channels of pixel
//     n shifted by any tap) is one aligned ds_read_b128 whose tap shift is a constant added to the address — consecutive lanes read
//     consecutive 16-byte slots (conflict free).  The transposition costs ~2 x 16 KB of LDS traffic per stage against ~290 KB of
//     fragment reads (nine taps reuse the tile).
// Workgroup = 512 threads = 8 waves (2 x 4), tile 128 channels x 256 pixels (R = 256 / W whole image rows), a wave owns 64 x 64
// (2 x 2 MFMA tiles); stage = 16 channels x all taps; A and the raw tile double-buffered by LDS-DMA one / two stages ahead, the
// transposition of stage s+1 runs inside stage s.  L2 -> LDS traffic: A 128 x 144 x 2 B + tile ~16 KB per 256 x 128 x 288 flop.
//
// Supported: image width W in {16, 32, 64, 128} (power of two), H a multiple of 256 / W, reduction channels a multiple of 16.
// Anything else -> IPSR_ERR_UNSUPPORTED (the dispatcher leaves it where it was).
#include "ipsr_common.h"

namespace ipsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) s16x4* lds_s4_t;

constexpr int CB_K = 128, CB_P = 256, CB_C = 16, CB_THREADS = 512, CB_MAXTAP = 9;
constexpr int CB_A_BYTES = CB_MAXTAP * 2 * CB_K * 16;       // [tap][c group][128 k][8 c] bf16 = 36 KB
constexpr int CB_RAW_BYTES = 16384;                         // [16 c][rows + halo][W] bf16, W <= 128: 16 x 4 x 256 B
constexpr int CB_T_BYTES = 17408;                           // [2][positions][8 c]: (R + 2) x (W + 2) <= 4 x 130 = 520 positions (+ slack)
constexpr int CB_TR_MAX = 4;                                // transposition blocks (4 channels x 16 pixels) per 16-lane group and stage
constexpr int CB_BUF = CB_A_BYTES + CB_RAW_BYTES + CB_T_BYTES;

struct CbGeom {
    int B, C, K, H, W;          // C reduction channels (multiple of 16), K produced channels
    int wshift;                 // log2 W
    int R, NR, PW, NPOS;        // image rows per tile (256 / W), raw rows (R + 2), padded width (W + 2), positions NR * PW
    int ntap;                   // 9
    int tapoff[CB_MAXTAP];      // (dy + 1) * PW + (dx + 1)
    int ktiles, ptiles;         // ceil(K / 128), B * H / R
    int nstage;                 // C / 16
};

// weight element (k, c, t) at w[c * sc + k * sk + tap], tap = flip ? 8 - t : t  ->  Wp[kt][cb][t][cg][k & 127][c & 7] bf16 (zero for k >= K)
__global__ void __launch_bounds__(256) cb_pack_weights_kernel(const float* __restrict__ w, int C, int K, long sc, long sk, int flip, int ntap,
                                                              uint4* __restrict__ Wp, uint4* __restrict__ zero_page)
{
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < 4) zero_page[threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
    const int k = blockIdx.x * 256 + threadIdx.x;          // padded produced channel
    const int c8 = blockIdx.y, t = blockIdx.z;
    const int ktiles = (K + CB_K - 1) / CB_K;
    if (k >= ktiles * CB_K) return;
    const int tap = flip ? ntap - 1 - t : t;
    unsigned short h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = c8 * 8 + e;
        const float v = k < K ? w[(long)c * sc + (long)k * sk + tap] : 0.0f;
        h[e] = __builtin_bit_cast(unsigned short, (__bf16)v);
    }
    uint4 o;
    o.x = h[0] | ((unsigned)h[1] << 16); o.y = h[2] | ((unsigned)h[3] << 16);
    o.z = h[4] | ((unsigned)h[5] << 16); o.w = h[6] | ((unsigned)h[7] << 16);
    const int kt = k >> 7, kl = k & 127, cb = c8 >> 1, cg = c8 & 1;
    const int nstage = C / CB_C;
    Wp[((((size_t)kt * nstage + cb) * ntap + t) * 2 + cg) * CB_K + kl] = o;
}

template <int NTAP, typename TOUT>
__global__ void __launch_bounds__(CB_THREADS, 1) conv_bf16_kernel(const unsigned short* __restrict__ in, const uint4* __restrict__ Wp,
                                                                  const uint4* __restrict__ zero_page, CbGeom g, TOUT* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];          // 2 x (A | raw | T)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 31, h = lane >> 5;

    // tile: all k tiles of one pixel tile are neighbours (they share the activation tile in L2)
    const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int kt = L % g.ktiles, pt = L / g.ktiles;
    const int tiles_per_img = g.H / g.R;
    const int b = pt / tiles_per_img, y0 = (pt - b * tiles_per_img) * g.R;

    // ---- stage-invariant addresses ------------------------------------------------------------------------------------------
    // raw tile DMA: chunk q (16 bytes = 8 pixels) = (c, row, seg), LDS image linear in q
    const int segs = g.W >> 3, nchunk = CB_C * g.NR * segs;
    const unsigned short* gx[2];
    bool xlive[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = tid + CB_THREADS * j;
        xlive[j] = q < nchunk;
        const int qq = xlive[j] ? q : 0;
        const int seg = qq % segs, row = (qq / segs) % g.NR, c = qq / (segs * g.NR);
        const int y = y0 - 1 + row;
        const bool inside = (unsigned)y < (unsigned)g.H;
        gx[j] = inside ? in + (((size_t)b * g.C + c) * g.H + y) * g.W + seg * 8 : nullptr;
    }
    const size_t xstride = (size_t)CB_C * g.H * g.W;
    // A tile DMA: NTAP * 4 pieces of 1 KiB, piece = wave + 8 j
    constexpr int NPIECE = NTAP * 4, APW = (NPIECE + 7) / 8;
    const uint4* ga = Wp + ((size_t)kt * g.nstage) * (NTAP * 2 * CB_K) + lane;
    // transposition: block u = (c quad, row, 16-pixel block); lane 4q+p of a 16-lane group supplies row q, pixels 4p..4p+3
    const int grp = tid >> 4, li = tid & 15;
    const int cb16s = g.W >> 4, nblk = 4 * g.NR * cb16s;
    int tr_rd[CB_TR_MAX], tr_wr[CB_TR_MAX];
#pragma unroll
    for (int j = 0; j < CB_TR_MAX; ++j) {
        int u = grp + 32 * j;
        if (u >= nblk) u = nblk - 1;                       // duplicates the last block (same data to the same place): EXEC stays full
        const int cb16 = u % cb16s, row = (u / cb16s) % g.NR, cq = u / (cb16s * g.NR);
        tr_rd[j] = (((cq * 4 + (li >> 2)) * g.NR + row) * g.W + cb16 * 16 + (li & 3) * 4) * 2;
        tr_wr[j] = (((cq >> 1) * g.NPOS + row * g.PW + cb16 * 16 + li + 1) * 8 + (cq & 1) * 4) * 2;
    }
    const int ntr = (nblk + 31) / 32;
    // fragments: A rows wm*64 + {0,32} + r; B pixels wn*64 + {0,32} + r
    const int a_off = (h * CB_K + wm * 64 + r) * 16;
    int b_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = wn * 64 + j * 32 + r;
        b_off[j] = (h * g.NPOS + (p >> g.wshift) * g.PW + (p & (g.W - 1))) * 16;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    auto dma_a = [&](int buf, int stage) {
        const uint4* src = ga + (size_t)stage * (NTAP * 2 * CB_K);
#pragma unroll
        for (int j = 0; j < APW; ++j) {
            const int piece = wave + 8 * j;
            if (piece < NPIECE)
                __builtin_amdgcn_global_load_lds((gptr_t)(src + piece * 64), (lptr_t)(lds + buf * CB_BUF + piece * 1024), 16, 0, 0);
        }
    };
    auto dma_x = [&](int buf, int stage) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (xlive[j]) {
                const void* src = gx[j] ? static_cast<const void*>(gx[j] + (size_t)stage * xstride) : static_cast<const void*>(zero_page);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + buf * CB_BUF + CB_A_BYTES + (wave * 64 + CB_THREADS * j) * 16), 16, 0, 0);
            }
        }
    };
    auto transpose = [&](int buf) {                            // raw[buf] -> T[buf]
        unsigned char* raw = lds + buf * CB_BUF + CB_A_BYTES;
        unsigned char* T = raw + CB_RAW_BYTES;
        s16x4 v[CB_TR_MAX];
#pragma unroll
        for (int j = 0; j < CB_TR_MAX; ++j)
            if (j < ntr) v[j] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(raw + tr_rd[j]));
#pragma unroll
        for (int j = 0; j < CB_TR_MAX; ++j)
            if (j < ntr) *reinterpret_cast<s16x4*>(T + tr_wr[j]) = v[j];
    };

    // halo columns of both T images: zero, never written again (tiles span the full image width)
    for (int i = tid; i < 2 * 2 * g.NR * 2; i += CB_THREADS) {
        const int side = i & 1, row = (i >> 1) % g.NR, cg = ((i >> 1) / g.NR) & 1, buf = (i >> 1) / (2 * g.NR);
        *reinterpret_cast<uint4*>(lds + buf * CB_BUF + CB_A_BYTES + CB_RAW_BYTES + (cg * g.NPOS + row * g.PW + (side ? g.PW - 1 : 0)) * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
    // prologue: A[0], raw[0] <- stage 0; raw[1] <- stage 1
    dma_a(0, 0);
    dma_x(0, 0);
    if (g.nstage > 1) dma_x(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    transpose(0);
    __syncthreads();

    for (int s = 0; s < g.nstage; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        if (s + 1 < g.nstage) dma_a(nxt, s + 1);               // A[nxt] was last read in stage s-1
        if (s + 2 < g.nstage) dma_x(cur, s + 2);               // raw[cur] was transposed in stage s-1
        if (s + 1 < g.nstage) transpose(nxt);                  // raw[nxt] (stage s+1) landed before the barrier that ended stage s-1
        const unsigned char* A = lds + cur * CB_BUF + a_off;
        const unsigned char* T = lds + cur * CB_BUF + CB_A_BYTES + CB_RAW_BYTES;
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
            const bf16x8 fa0 = *reinterpret_cast<const bf16x8*>(A + (t * 2 * CB_K) * 16);
            const bf16x8 fa1 = *reinterpret_cast<const bf16x8*>(A + (t * 2 * CB_K + 32) * 16);
            const bf16x8 fb0 = *reinterpret_cast<const bf16x8*>(T + b_off[0] + g.tapoff[t] * 16);
            const bf16x8 fb1 = *reinterpret_cast<const bf16x8*>(T + b_off[1] + g.tapoff[t] * 16);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb1, acc[1][1], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this stage's DMAs are the next stage's operands
        __syncthreads();
    }

    // epilogue: lane = pixel, register = channel
    const size_t HW = (size_t)g.H * g.W;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = wn * 64 + j * 32 + r;
        TOUT* op = out + (size_t)b * g.K * HW + (size_t)(y0 + (p >> g.wshift)) * g.W + (p & (g.W - 1));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = kt * CB_K + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (k < g.K) st1(op, (size_t)k * HW, acc[i][j][e]);
            }
    }
}

static int cb_geometry(int B, int C, int K, int H, int W, CbGeom* g)
{
    if (W != 16 && W != 32 && W != 64 && W != 128) return fail(IPSR_ERR_UNSUPPORTED, "bf16 direct conv: image width %d (16, 32, 64 or 128)", W);
    const int R = CB_P / W;
    if (H % R != 0) return fail(IPSR_ERR_UNSUPPORTED, "bf16 direct conv: %d rows are not a multiple of the %d rows of a tile", H, R);
    if (C % CB_C != 0) return fail(IPSR_ERR_UNSUPPORTED, "bf16 direct conv: %d reduction channels are not a multiple of %d", C, CB_C);
    g->B = B; g->C = C; g->K = K; g->H = H; g->W = W;
    g->wshift = W == 16 ? 4 : (W == 32 ? 5 : (W == 64 ? 6 : 7));
    g->R = R; g->NR = R + 2; g->PW = W + 2; g->NPOS = g->NR * g->PW;
    g->ntap = 9;
    for (int t = 0; t < 9; ++t) g->tapoff[t] = (t / 3) * g->PW + (t % 3);
    g->ktiles = (K + CB_K - 1) / CB_K;
    g->ptiles = B * (H / R);
    g->nstage = C / CB_C;
    if (CB_C * g->NR * W * 2 > CB_RAW_BYTES || 2 * g->NPOS * 16 > CB_T_BYTES || 4 * g->NR * (W / 16) > 32 * CB_TR_MAX)
        return fail(IPSR_ERR_UNSUPPORTED, "bf16 direct conv: tile of %d rows x %d does not fit the LDS plan", g->NR, W);
    return IPSR_OK;
}

size_t conv_bf16_ws_bytes(int B, int C, int K, int H, int W)
{
    CbGeom g;
    if (cb_geometry(B, C, K, H, W, &g) != IPSR_OK) return 0;
    return 256 + (size_t)g.ktiles * g.nstage * 9 * 2 * CB_K * 16;
}

// in [B,C,H,W] bf16, weight fp32 with element (c, k, tap) at w[c*sc + k*sk + tap] (taps flipped when `flip`), out [B,K,H,W] bf16 / fp32
int launch_conv_bf16(const void* in, const float* w, void* out, int B, int C, int K, int H, int W, long sc, long sk, int flip, int out_bf16,
                     void* ws, size_t ws_bytes, hipStream_t st)
{
    CbGeom g;
    if (int rc = cb_geometry(B, C, K, H, W, &g)) return rc;
    const size_t need = conv_bf16_ws_bytes(B, C, K, H, W);
    if (ws_bytes < need) return fail(IPSR_ERR_WORKSPACE, "bf16 direct conv: workspace %zu < %zu", ws_bytes, need);
    uint4* zero_page = static_cast<uint4*>(ws);
    uint4* Wp = zero_page + 16;
    cb_pack_weights_kernel<<<dim3(cdiv(g.ktiles * CB_K, 256), C / 8, 9), 256, 0, st>>>(w, C, K, sc, sk, flip, 9, Wp, zero_page);
    if (int rc = check_launch("cb_pack_weights_kernel")) return rc;
    const unsigned grid = (unsigned)(g.ktiles * g.ptiles);
    const size_t smem = 2 * (size_t)CB_BUF;
    profile_mark_start(st, 3);
    if (out_bf16) {
        static bool attr_b = false;
        if (!attr_b) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_kernel<9, bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); attr_b = true; }
        conv_bf16_kernel<9, bf16_t><<<grid, CB_THREADS, smem, st>>>(static_cast<const unsigned short*>(in), Wp, zero_page, g, static_cast<bf16_t*>(out));
    } else {
        static bool attr_f = false;
        if (!attr_f) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_kernel<9, float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); attr_f = true; }
        conv_bf16_kernel<9, float><<<grid, CB_THREADS, smem, st>>>(static_cast<const unsigned short*>(in), Wp, zero_page, g, static_cast<float*>(out));
    }
    profile_mark_stop(st, 3, 2.0 * 9.0 * C * (double)(g.ktiles * CB_K) * B * H * W, 2.0 * 9.0 * C * (double)K * B * H * W);
    return check_launch("conv_bf16_kernel");
}

}  // namespace ipsr

using namespace ipsr;

extern "C" {

size_t ipsr_conv3x3_bf16_workspace_bytes(int op, int B, int Cin, int H, int W, int Cout)
{
    if (op < 0 || op > 3 || B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1) return 0;
    const bool fwd = op == 0 || op == 2;
    return conv_bf16_ws_bytes(B, fwd ? Cin : Cout, fwd ? Cout : Cin, H, W);
}

int ipsr_conv3x3_bf16(int op, const void* in, const float* weight, void* out, int B, int Cin, int H, int W, int Cout, int out_bf16,
                      void* ws, size_t ws_bytes, void* stream)
{
    if (!in || !weight || !out || !ws) return fail(IPSR_ERR_INVALID, "ipsr_conv3x3_bf16: null pointer");
    if (op < 0 || op > 3 || B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1) return fail(IPSR_ERR_INVALID, "ipsr_conv3x3_bf16: bad argument");
    if ((reinterpret_cast<uintptr_t>(ws) & 15u) || (reinterpret_cast<uintptr_t>(in) & 15u))
        return fail(IPSR_ERR_INVALID, "ipsr_conv3x3_bf16: in / workspace must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (op) {       // (sc, sk, flip) as in ipsr_conv3x3_winograd_mp: C = reduction channels, K = produced channels
        case 0: return launch_conv_bf16(in, weight, out, B, Cin, Cout, H, W, 9, (long)Cin * 9, 0, out_bf16, ws, ws_bytes, st);
        case 1: return launch_conv_bf16(in, weight, out, B, Cout, Cin, H, W, (long)Cin * 9, 9, 1, out_bf16, ws, ws_bytes, st);
        case 2: return launch_conv_bf16(in, weight, out, B, Cin, Cout, H, W, (long)Cout * 9, 9, 1, out_bf16, ws, ws_bytes, st);
        default: return launch_conv_bf16(in, weight, out, B, Cout, Cin, H, W, 9, (long)Cout * 9, 0, out_bf16, ws, ws_bytes, st);
    }
}

}  // extern "C"
