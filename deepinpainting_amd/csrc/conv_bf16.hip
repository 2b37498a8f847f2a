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
// per-thread loop bounds of a stage, by pixels per tile (256 / 512): transposition blocks (4 channels x 16 pixels) per 16-lane group, and
// 16-byte raw-tile chunks per thread
constexpr int cb_tr_max(int ptile) { return ptile == 256 ? 6 : 10; }
constexpr int cb_xj(int ptile) { return ptile == 256 ? 3 : 5; }
constexpr int CB_LDS_MAX = 160 * 1024;
// LDS plan (per geometry, cb_finish): A[2] | T[2] | raw[2 or 1].  A stage = ntap x 4 KB ([tap][c group][128 k][8 c] bf16), raw = [16 c][raw
// rows][input width] bf16, T = [planes][2 c groups][positions][8 c].  k3 at W <= 128: 2 x (36 + 16 + 16.3) KB; stride 2: 2 x (32 + 20 + 20.4);
// k3 at W = 256 (one image row per tile): 2 x 36 + 2 x 24.3 + ONE raw buffer of 24 KB — the transposition then follows the stage's
// multiplications instead of running beside them (`raw1`).

// The three forms (all: out = sum over (channel, tap) of packed weight x shifted input; lanes = pixels of the LANE grid Hl x Wl):
//   S1   k3 s1 p1                     lane grid = image; 9 taps; raw tile = R + 2 rows of the image, T = [cg][R + 2][W + 2]
//   F2C  k4 s2 p1, fine -> coarse     (Conv2d forward, ConvTranspose2d input gradient) lane grid = COARSE output, in(2o - 1 + r).  A stage =
//        16 channels x the 8 taps whose input ROW parity is ey: raw tile = the R + 1 fine rows of that parity (full fine width), the
//        transposition splits them into the two COLUMN phases, T = [ex][cg][R + 1][Wl + 1] — every tap a unit-stride shift again.
//   C2F  k4 s2 p1, coarse -> fine     (ConvTranspose2d forward, Conv2d input gradient) lane grid = COARSE input; a workgroup produces the
//        fine rows of ONE row parity ey' and both column parities (two accumulator sets of 2 x 2 taps each: 8 taps per stage), raw / T as
//        S1 on the coarse image; the epilogue interleaves the two column phases into whole fine rows.
enum { CB_S1 = 0, CB_F2C = 1, CB_C2F = 2 };

struct CbGeom {
    int B, C, K;                // C reduction channels (multiple of 16), K produced channels
    int Hin, Win;               // the tensor the kernel reads
    int Hl, Wl, wshift;         // lane grid (rows, width = power of two), log2 Wl
    int Hout, Wout;             // the tensor written
    int R, NR, PW, NPOS;        // lane-grid rows per tile (ptile / Wl), raw rows, padded width, positions NR * PW per (plane, c group)
    int ymul, rowstep, yoff[2]; // input row of raw row i of sub-stage e: ymul * y0 + yoff[e] + rowstep * i
    int nsub, nphase;           // stages per channel block (F2C: 2), output row phases = workgroups per (k tile, pixel tile) (C2F: 2)
    int ntap;
    int tapoff[2][CB_MAXTAP];   // per sub-stage (F2C) / row phase (C2F): plane * 2 * NPOS + drow * PW + dcol
    int ktiles, ptiles, nstage; // ceil(K / 128), B * Hl / R, (C / 16) * nsub
    int a_bytes, t_bytes, raw_bytes, raw1;      // LDS plan: stage sizes, one raw buffer instead of two
    int kt, ptile;              // produced channels per workgroup tile: 128, or 64 when K <= 64; pixels per tile: 256, or 512 (64-row kernel)
    int nsplit, sps;            // reduction split (small maps: too few tiles to fill the chip): runs of `sps` stages, fp32 partials added by a second launch
};

// packed weights: Wp[kt][phase][cb][sub][t][cg][k & 127][c & 7] bf16 (zero for k >= K); source element (k, c, tap) at w[c * sc + k * sk + srctap]
struct CbPack { int ntap, nsub, nphase, kt; int srctap[2][2][CB_MAXTAP]; };      // kt: rows per k tile (128 or 64); srctap[phase][sub][t]

__global__ void __launch_bounds__(256) cb_pack_weights_kernel(const float* __restrict__ w, int C, int K, long sc, long sk, CbPack pk,
                                                              uint4* __restrict__ Wp, uint4* __restrict__ zero_page)
{
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < 4) zero_page[threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
    const int k = blockIdx.x * 256 + threadIdx.x;          // padded produced channel
    const int c8 = blockIdx.y;
    const int t = blockIdx.z % pk.ntap, ps = blockIdx.z / pk.ntap, sub = ps % pk.nsub, phase = ps / pk.nsub;
    const int ktiles = (K + pk.kt - 1) / pk.kt;
    if (k >= ktiles * pk.kt) return;
    const int tap = pk.srctap[phase][sub][t];
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
    const int kt = k / pk.kt, kl = k - kt * pk.kt, cb = c8 >> 1, cg = c8 & 1;
    const int ncb = C / CB_C;
    Wp[((((((size_t)kt * pk.nphase + phase) * ncb + cb) * pk.nsub + sub) * pk.ntap + t) * 2 + cg) * pk.kt + kl] = o;
}

// KT = produced channels per workgroup tile: 128 (waves 2 x 4, a wave 64 x 64), or 64 for layers that produce <= 64 channels (waves 1 x 8, a
// wave 64 x 32) — a 128-row tile on a 64-channel layer multiplies zeros half of the time (VGG conv1_2, the outermost U-Net levels).
template <int MODE, int KT, int P, typename TOUT>
__global__ void __launch_bounds__(CB_THREADS, 1) conv_bf16_kernel(const unsigned short* __restrict__ in, const uint4* __restrict__ Wp,
                                                                  const uint4* __restrict__ zero_page, CbGeom g, TOUT* __restrict__ out)
{
    constexpr int NTAP = MODE == CB_S1 ? 9 : 8;
    constexpr int NSET = MODE == CB_C2F ? 2 : 1;               // accumulator sets (C2F: the two column phases of the fine output)
    constexpr int TPSET = NTAP / NSET;
    // P = pixels per tile: 256 under 128 produced channels; under 64 it is 512 where the map has the rows for it — a wave owns 64 x 64
    // either way (4 fragment reads per 4 MFMAs; on a 64 x 256 tile it is 64 x 32: 3 reads per 2 MFMAs = 192 B/clk/CU of the LDS's 256)
    static_assert((KT == 128 && P == 256) || (KT == 64 && (P == 256 || P == 512)), "tile");
    constexpr int WN = KT == 128 ? 4 : 8, NJ = P / 32 / WN;    // wave columns; 32-pixel MFMA tiles per wave
    constexpr int CB_TR_MAX = cb_tr_max(P);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];          // 2 x (A | raw | T)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = KT == 128 ? wave >> 2 : 0, wn = KT == 128 ? wave & 3 : wave;
    const int r = lane & 31, h = lane >> 5;

    // tile: all (k tile, row phase) workgroups of one pixel tile are neighbours (they share the activation tile in L2)
    const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int per_pt = g.ktiles * g.nphase * g.nsplit;
    const int pt = L / per_pt, kps = L - pt * per_pt;
    const int split = kps % g.nsplit, kp = kps / g.nsplit;
    const int kt = kp % g.ktiles, phase = kp / g.ktiles;
    const int s_lo = split * g.sps, s_hi = min(g.nstage, s_lo + g.sps);        // this workgroup's run of stages (the whole reduction unless split)
    out += (size_t)split * g.B * g.K * g.Hout * g.Wout;                         // split runs write their own fp32 partial
    const int tiles_per_img = g.Hl / g.R;
    const int b = pt / tiles_per_img, y0 = (pt - b * tiles_per_img) * g.R;

    // ---- stage-invariant addresses ------------------------------------------------------------------------------------------
    // raw tile DMA: chunk q (16 bytes = 8 pixels) = (c, row, seg), LDS image linear in q; per sub-stage its own row set
    const int segs = g.Win >> 3, nchunk = CB_C * g.NR * segs;
    constexpr int XJ = cb_xj(P);
    const unsigned short* gx[XJ][MODE == CB_F2C ? 2 : 1];
    bool xlive[XJ];
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
        const int q = tid + CB_THREADS * j;
        xlive[j] = q < nchunk;
        const int qq = xlive[j] ? q : 0;
        const int seg = qq % segs, row = (qq / segs) % g.NR, c = qq / (segs * g.NR);
#pragma unroll
        for (int e = 0; e < (MODE == CB_F2C ? 2 : 1); ++e) {
            const int y = g.ymul * y0 + g.yoff[e] + g.rowstep * row;
            const bool inside = (unsigned)y < (unsigned)g.Hin;
            gx[j][e] = inside ? in + (((size_t)b * g.C + c) * g.Hin + y) * g.Win + seg * 8 : nullptr;
        }
    }
    const size_t xstride = (size_t)CB_C * g.Hin * g.Win;
    // A tile DMA: NTAP * 4 pieces of 1 KiB, piece = wave + 8 j; the stages of (kt, phase) are contiguous
    constexpr int NPIECE = NTAP * KT / 32, APW = (NPIECE + 7) / 8;
    const uint4* ga = Wp + ((size_t)(kt * g.nphase + phase) * g.nstage) * (NTAP * 2 * KT) + lane;
    // transposition: block u = (c quad, row, 16-pixel block); lane 4q+p of a 16-lane group supplies row q, pixels 4p..4p+3
    const int grp = tid >> 4, li = tid & 15;
    const int cb16s = g.Win >> 4, nblk = 4 * g.NR * cb16s;
    int tr_rd[CB_TR_MAX], tr_wr[CB_TR_MAX];
#pragma unroll
    for (int j = 0; j < CB_TR_MAX; ++j) {
        int u = grp + 32 * j;
        if (u >= nblk) u = nblk - 1;                       // duplicates the last block (same data to the same place): EXEC stays full
        const int cb16 = u % cb16s, row = (u / cb16s) % g.NR, cq = u / (cb16s * g.NR);
        tr_rd[j] = (((cq * 4 + (li >> 2)) * g.NR + row) * g.Win + cb16 * 16 + (li & 3) * 4) * 2;
        const int x = cb16 * 16 + li;
        int pos;
        if (MODE == CB_F2C) { const int ex = x & 1; pos = (ex * 2 + (cq >> 1)) * g.NPOS + row * g.PW + (x >> 1) + ex; }      // x = 2 j + ex; plane 1 starts at j = -1
        else pos = (cq >> 1) * g.NPOS + row * g.PW + x + 1;
        tr_wr[j] = (pos * 8 + (cq & 1) * 4) * 2;
    }
    const int ntr = (nblk + 31) / 32;
    const int t_base = 2 * g.a_bytes, raw_base = t_base + 2 * g.t_bytes;
    // fragments: A rows wm*64 + {0,32} + r; B lane-grid pixels wn*64 + {0,32} + r
    const int a_off = (h * KT + wm * 64 + r) * 16;
    int b_off[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int p = wn * (32 * NJ) + j * 32 + r;
        b_off[j] = (h * g.NPOS + (p >> g.wshift) * g.PW + (p & (g.Wl - 1))) * 16;
    }

    f32x16 acc[NSET][2][NJ];
#pragma unroll
    for (int q = 0; q < NSET; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[q][i][j][e] = 0.0f;

    auto dma_a = [&](int buf, int stage) {
        const uint4* src = ga + (size_t)stage * (NTAP * 2 * KT);
#pragma unroll
        for (int j = 0; j < APW; ++j) {
            const int piece = wave + 8 * j;
            if (piece < NPIECE)
                __builtin_amdgcn_global_load_lds((gptr_t)(src + piece * 64), (lptr_t)(lds + buf * g.a_bytes + piece * 1024), 16, 0, 0);
        }
    };
    auto dma_x = [&](int buf, int stage) {
        const int cb = MODE == CB_F2C ? stage >> 1 : stage, e = MODE == CB_F2C ? stage & 1 : 0;
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
            if (xlive[j]) {
                const unsigned short* base = (MODE == CB_F2C && e) ? gx[j][MODE == CB_F2C ? 1 : 0] : gx[j][0];
                const void* src = base ? static_cast<const void*>(base + (size_t)cb * xstride) : static_cast<const void*>(zero_page);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + raw_base + buf * g.raw_bytes + (wave * 64 + CB_THREADS * j) * 16), 16, 0, 0);
            }
        }
    };
    auto transpose = [&](int rbuf, int tbuf) {                 // raw[rbuf] -> T[tbuf]
        unsigned char* raw = lds + raw_base + rbuf * g.raw_bytes;
        unsigned char* T = lds + t_base + tbuf * g.t_bytes;
        s16x4 v[CB_TR_MAX];
#pragma unroll
        for (int j = 0; j < CB_TR_MAX; ++j)
            if (j < ntr) v[j] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(raw + tr_rd[j]));
#pragma unroll
        for (int j = 0; j < CB_TR_MAX; ++j)
            if (j < ntr) *reinterpret_cast<s16x4*>(T + tr_wr[j]) = v[j];
    };

    // both T images start as zeros: the halo columns are never written (tiles span the full image width)
    for (int i = tid; i < 2 * (g.t_bytes / 16); i += CB_THREADS)
        *reinterpret_cast<uint4*>(lds + t_base + i * 16) = make_uint4(0u, 0u, 0u, 0u);
    // prologue: A[0], raw[0] <- stage 0; raw[1] <- stage 1 (two raw buffers)
    dma_a(0, s_lo);
    dma_x(0, s_lo);
    if (!g.raw1 && s_lo + 1 < s_hi) dma_x(1, s_lo + 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    transpose(0, 0);
    __syncthreads();

    for (int s = s_lo; s < s_hi; ++s) {
        const int cur = (s - s_lo) & 1, nxt = cur ^ 1;
        // The transposition goes first: hipcc drains every LDS-DMA in flight (s_waitcnt vmcnt(0)) before a ds_read_b64_tr_b16, so a
        // DMA issued ahead of it would be waited for here instead of behind the multiplications.
        if (!g.raw1 && s + 1 < s_hi) transpose(nxt, nxt);      // raw[nxt] (stage s+1) landed before the barrier that ended stage s-1
        if (s + 1 < s_hi) dma_a(nxt, s + 1);                   // A[nxt] was last read in stage s-1
        if (!g.raw1) {
            if (s + 2 < s_hi) dma_x(cur, s + 2);               // raw[cur] was transposed in stage s-1
        } else if (s + 1 < s_hi) {
            dma_x(0, s + 1);                                   // the one raw buffer was transposed at the end of stage s-1
        }
        const unsigned char* A = lds + cur * g.a_bytes + a_off;
        const unsigned char* T = lds + t_base + cur * g.t_bytes;
        const int* toff = g.tapoff[MODE == CB_F2C ? (s & 1) : (MODE == CB_C2F ? phase : 0)];
        // software pipeline over the taps: the fragments of tap t + 1 are read while tap t multiplies (two register sets); the
        // sched_group_barriers pin that order — left alone the compiler issues a tap's reads right before its own multiplications and
        // every tap waits out the LDS latency (s_waitcnt lgkmcnt(0) in front of 4 MFMAs)
        bf16x8 fa[2][2], fb[2][NJ];
        auto load_tap = [&](int t, int set) {
            fa[set][0] = *reinterpret_cast<const bf16x8*>(A + (t * 2 * KT) * 16);
            fa[set][1] = *reinterpret_cast<const bf16x8*>(A + (t * 2 * KT + 32) * 16);
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[set][j] = *reinterpret_cast<const bf16x8*>(T + b_off[j] + toff[t] * 16);
        };
        load_tap(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 + NJ, 0);                              // tap 0's reads: the pipeline's fill
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
            const int q = t / TPSET, set = t & 1;
            if (t + 1 < NTAP) load_tap(t + 1, set ^ 1);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                acc[q][0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][0], fb[set][j], acc[q][0][j], 0, 0, 0);
                acc[q][1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][1], fb[set][j], acc[q][1][j], 0, 0, 0);
            }
            if (t + 1 < NTAP) {
                // (with an LDS-DMA in flight hipcc only ever waits lgkmcnt(0), never a counted value: the reads must all be OLD when the next
                // tap's first multiplication asks for them, so they go right behind this tap's first one, not one per multiplication)
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                           // this tap's first MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 2 + NJ, 0);                      // the next tap's reads
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * NJ - 1, 0);                  // the rest of this tap
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this stage's DMAs are the next stage's operands
        __syncthreads();
        if (g.raw1 && s + 1 < s_hi) {                          // one raw buffer: the next stage's tile crosses LDS now, behind the multiplications
            transpose(0, nxt);
            __syncthreads();
        }
    }

    // epilogue: lane = lane-grid pixel, register = channel
    const size_t HWo = (size_t)g.Hout * g.Wout;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int p = wn * (32 * NJ) + j * 32 + r;
        const int py = y0 + (p >> g.wshift), px = p & (g.Wl - 1);
        if (MODE == CB_C2F) {
            // fine row 2 py + phase, fine columns 2 px and 2 px + 1 (the two accumulator sets): one 2-element store
            TOUT* op = out + (size_t)b * g.K * HWo + (size_t)(2 * py + phase) * g.Wout + 2 * px;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int k = kt * KT + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (k < g.K) {
                        if (sizeof(TOUT) == 2) {
                            const unsigned pk = (unsigned)f2bf(acc[0][i][j][e]) | ((unsigned)f2bf(acc[NSET - 1][i][j][e]) << 16);
                            *reinterpret_cast<unsigned*>(op + (size_t)k * HWo) = pk;
                        } else {
                            *reinterpret_cast<float2*>(op + (size_t)k * HWo) = make_float2(acc[0][i][j][e], acc[NSET - 1][i][j][e]);
                        }
                    }
                }
        } else if (sizeof(TOUT) == 2) {
            // bf16 output: through LDS (free after the last stage's barrier) as [k][P pixels], so that the tile leaves as 16-byte rows
            // instead of 64 two-byte stores per lane
            unsigned short* L = reinterpret_cast<unsigned short*>(lds);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int kl = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    L[kl * P + p] = f2bf(acc[0][i][j][e]);
                }
        } else {
            const bool live = py < g.Hout && px < g.Wout;      // (lane grid == output grid here)
            TOUT* op = out + (size_t)b * g.K * HWo + (size_t)py * g.Wout + px;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int k = kt * KT + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (live && k < g.K) st1(op, (size_t)k * HWo, acc[0][i][j][e]);
                }
        }
    }
    if (MODE != CB_C2F && sizeof(TOUT) == 2) {
        __syncthreads();
        const uint4* L4 = reinterpret_cast<const uint4*>(lds);
#pragma unroll
        for (int it = 0; it < KT * P * 2 / 16 / CB_THREADS; ++it) {
            const int chunk = tid + CB_THREADS * it;           // P / 8 chunks of 8 pixels per channel row
            const int kl = chunk / (P / 8), p0 = (chunk % (P / 8)) * 8;
            const int k = kt * KT + kl;
            if (k < g.K) {
                TOUT* op = out + ((size_t)b * g.K + k) * HWo + (size_t)(y0 + (p0 >> g.wshift)) * g.Wout + (p0 & (g.Wl - 1));
                *reinterpret_cast<uint4*>(op) = L4[chunk];
            }
        }
    }
}

static int cb_lane_grid(int Hl, int Wl, int ptile, CbGeom* g, const char* who)
{
    if (Wl != 16 && Wl != 32 && Wl != 64 && Wl != 128 && Wl != 256) return fail(IPSR_ERR_UNSUPPORTED, "%s: grid width %d (16 .. 256, a power of two)", who, Wl);
    const int R = ptile / Wl;
    g->ptile = ptile;
    if (Hl % R != 0) return fail(IPSR_ERR_UNSUPPORTED, "%s: %d rows are not a multiple of the %d rows of a tile", who, Hl, R);
    g->Hl = Hl; g->Wl = Wl; g->R = R;
    g->wshift = Wl == 16 ? 4 : (Wl == 32 ? 5 : (Wl == 64 ? 6 : (Wl == 128 ? 7 : 8)));
    return IPSR_OK;
}

static int cb_finish(CbGeom* g, const char* who)
{
    g->NPOS = g->NR * g->PW;
    g->kt = g->K <= 64 ? 64 : CB_K;
    g->ktiles = (g->K + g->kt - 1) / g->kt;
    g->ptiles = g->B * (g->Hl / g->R);
    g->nstage = (g->C / CB_C) * g->nsub;
    // Small maps leave the chip idle (a 16x16 map is ONE pixel tile per image: 64 workgroups at 512 produced channels and batch 16): the
    // reduction is cut into up to four runs of whole channel blocks, each run a workgroup of its own writing an fp32 partial.
    {
        const int wgs = g->ktiles * g->nphase * g->ptiles, nblocks = g->C / CB_C;
        int ns = 1;
        if (wgs < 128 && nblocks >= 8) ns = min(min(4, nblocks / 4), (256 + wgs - 1) / wgs);
        const int bps = (nblocks + ns - 1) / max(ns, 1);      // channel blocks per run
        g->nsplit = (nblocks + bps - 1) / bps;
        g->sps = bps * g->nsub;
    }
    const int planes = g->nsub;                               // F2C keeps the two column phases
    g->a_bytes = g->ntap * 2 * g->kt * 16;
    g->t_bytes = (int)align_up((size_t)planes * 2 * g->NPOS * 16, 256);
    g->raw_bytes = (int)align_up((size_t)CB_C * g->NR * g->Win * 2, 1024);
    g->raw1 = 2 * (g->a_bytes + g->t_bytes + g->raw_bytes) > CB_LDS_MAX;
    if (2 * (g->a_bytes + g->t_bytes) + (g->raw1 ? 1 : 2) * g->raw_bytes > CB_LDS_MAX || 4 * g->NR * (g->Win / 16) > 32 * cb_tr_max(g->ptile) ||
        CB_C * g->NR * (g->Win / 8) > cb_xj(g->ptile) * CB_THREADS)
        return fail(IPSR_ERR_UNSUPPORTED, "%s: a tile of %d rows x %d does not fit the LDS plan", who, g->NR, g->Win);
    return IPSR_OK;
}

// k3 s1 p1
static int cb_geometry_p(int B, int C, int K, int H, int W, int ptile, CbGeom* g)
{
    if (C % CB_C != 0) return fail(IPSR_ERR_UNSUPPORTED, "bf16 direct conv: %d reduction channels are not a multiple of %d", C, CB_C);
    if (int rc = cb_lane_grid(H, W, ptile, g, "bf16 direct conv")) return rc;
    g->B = B; g->C = C; g->K = K; g->Hin = H; g->Win = W; g->Hout = H; g->Wout = W;
    g->NR = g->R + 2; g->PW = W + 2;
    g->ymul = 1; g->rowstep = 1; g->yoff[0] = -1; g->yoff[1] = -1;
    g->nsub = 1; g->nphase = 1; g->ntap = 9;
    for (int t = 0; t < 9; ++t) g->tapoff[0][t] = g->tapoff[1][t] = (t / 3) * g->PW + (t % 3);
    return cb_finish(g, "bf16 direct conv");
}

// <= 64 produced channels: the 512-pixel tile where the map and the LDS plan allow it, else 256
static int cb_geometry(int B, int C, int K, int H, int W, CbGeom* g)
{
    if (K <= 64 && cb_geometry_p(B, C, K, H, W, 2 * CB_P, g) == IPSR_OK) return IPSR_OK;
    return cb_geometry_p(B, C, K, H, W, CB_P, g);
}

static int cb_geometry_s2_p(int form, int B, int C, int K, int nh, int nw, int ptile, CbGeom* g);
// k4 s2 p1: fine [.,Cf,2nh,2nw], coarse [.,Kc,nh,nw].  form 0: fine -> coarse (C = Cf reduced, K = Kc produced); 1: coarse -> fine
static int cb_geometry_s2(int form, int B, int C, int K, int nh, int nw, CbGeom* g)
{
    if (K <= 64 && cb_geometry_s2_p(form, B, C, K, nh, nw, 2 * CB_P, g) == IPSR_OK) return IPSR_OK;
    return cb_geometry_s2_p(form, B, C, K, nh, nw, CB_P, g);
}

static int cb_geometry_s2_p(int form, int B, int C, int K, int nh, int nw, int ptile, CbGeom* g)
{
    if (C % CB_C != 0) return fail(IPSR_ERR_UNSUPPORTED, "bf16 direct 4x4 stride-2 conv: %d reduction channels are not a multiple of %d", C, CB_C);
    if (int rc = cb_lane_grid(nh, nw, ptile, g, "bf16 direct 4x4 stride-2 conv")) return rc;
    g->B = B; g->C = C; g->K = K;
    if (form == 0) {
        g->Hin = 2 * nh; g->Win = 2 * nw; g->Hout = nh; g->Wout = nw;
        g->NR = g->R + 1; g->PW = nw + 1;
        g->ymul = 2; g->rowstep = 2; g->yoff[0] = 0; g->yoff[1] = -1;       // sub-stage e = input row parity: rows 2 (y0 + i) + e - 2 e
        g->nsub = 2; g->nphase = 1; g->ntap = 8;
        g->NPOS = g->NR * g->PW;
        // tap t = ri * 4 + s of sub-stage e: r = e ? {0, 2}[ri] : {1, 3}[ri]; input row 2 oy - 1 + r = 2 (oy + drow - (e ? 1 : 0)) + e  ->  drow = ri
        // column 2 ox - 1 + s: s even -> odd column (plane 1, j = ox - 1 + s / 2, stored at j + 1), s odd -> even column (plane 0, j = ox + s / 2)
        for (int e = 0; e < 2; ++e)
            for (int t = 0; t < 8; ++t) {
                const int ri = t >> 2, sx = t & 3, ex = (sx & 1) ? 0 : 1, dcol = sx >> 1;
                g->tapoff[e][t] = ex * 2 * g->NPOS + ri * g->PW + dcol;
            }
    } else {
        g->Hin = nh; g->Win = nw; g->Hout = 2 * nh; g->Wout = 2 * nw;
        g->NR = g->R + 2; g->PW = nw + 2;
        g->ymul = 1; g->rowstep = 1; g->yoff[0] = -1; g->yoff[1] = -1;
        g->nsub = 1; g->nphase = 2; g->ntap = 8;
        g->NPOS = g->NR * g->PW;
        // row phase ey' (workgroup), column phase ex' (accumulator set), taps (ai, bi): fine row 2 i + ey' takes coarse rows
        // ey' = 0: {i (r = 1), i - 1 (r = 3)};  ey' = 1: {i + 1 (r = 0), i (r = 2)}  ->  raw row (i - y0) + 1 + d, d = ey' - ai
        for (int ey = 0; ey < 2; ++ey)
            for (int t = 0; t < 8; ++t) {
                const int ex = t >> 2, ai = (t >> 1) & 1, bi = t & 1;
                g->tapoff[ey][t] = (1 + ey - ai) * g->PW + (1 + ex - bi);
            }
    }
    return cb_finish(g, "bf16 direct 4x4 stride-2 conv");
}

// fp32 partials of a split reduction, behind the packed weights
static size_t cb_partial_bytes(const CbGeom& g) { return g.nsplit > 1 ? align_up((size_t)g.nsplit * g.B * g.K * g.Hout * g.Wout * sizeof(float), 256) : 0; }

// out[i] = sum over the runs' partials, ascending
template <typename TOUT>
__global__ void __launch_bounds__(256) cb_split_reduce_kernel(const float* __restrict__ part, int nsplit, size_t n4, TOUT* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 a = reinterpret_cast<const float4*>(part)[i];
    for (int sI = 1; sI < nsplit; ++sI) {
        const float4 v = reinterpret_cast<const float4*>(part)[(size_t)sI * n4 + i];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    st4(out, i, a);
}

size_t conv_bf16_ws_bytes(int B, int C, int K, int H, int W)
{
    CbGeom g;
    if (cb_geometry(B, C, K, H, W, &g) != IPSR_OK) return 0;
    return 256 + (size_t)g.ktiles * g.nstage * 9 * 2 * g.kt * 16 + cb_partial_bytes(g);
}

size_t conv_bf16_s2_ws_bytes(int form, int B, int C, int K, int nh, int nw)
{
    CbGeom g;
    if (cb_geometry_s2(form, B, C, K, nh, nw, &g) != IPSR_OK) return 0;
    return 256 + (size_t)g.ktiles * g.nphase * g.nstage * 8 * 2 * g.kt * 16 + cb_partial_bytes(g);
}

template <int MODE, int KT, int P, typename TOUT>
static void cb_launch_kernel(const CbGeom& g, const void* in, const uint4* Wp, const uint4* zero_page, void* out, unsigned grid, size_t smem, hipStream_t st)
{
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_kernel<MODE, KT, P, TOUT>), hipFuncAttributeMaxDynamicSharedMemorySize, CB_LDS_MAX); attr = true; }
    conv_bf16_kernel<MODE, KT, P, TOUT><<<grid, CB_THREADS, smem, st>>>(static_cast<const unsigned short*>(in), Wp, zero_page, g, static_cast<TOUT*>(out));
}

template <int MODE>
static int cb_launch(const CbGeom& g, CbPack pk, const void* in, const float* w, void* out, long sc, long sk, int out_bf16, void* ws,
                     hipStream_t st, double taps_per_out, int pack_valid = 0)
{
    uint4* zero_page = static_cast<uint4*>(ws);
    uint4* Wp = zero_page + 16;
    pk.kt = g.kt;
    if (!pack_valid) {
        cb_pack_weights_kernel<<<dim3(cdiv(g.ktiles * g.kt, 256), g.C / 8, pk.ntap * pk.nsub * pk.nphase), 256, 0, st>>>(w, g.C, g.K, sc, sk, pk, Wp, zero_page);
        if (int rc = check_launch("cb_pack_weights_kernel")) return rc;
    }
    const unsigned grid = (unsigned)(g.ktiles * g.nphase * g.ptiles * g.nsplit);
    const size_t smem = 2 * (size_t)(g.a_bytes + g.t_bytes) + (size_t)(g.raw1 ? 1 : 2) * g.raw_bytes;
    void* final_out = out;
    const int final_bf16 = out_bf16;
    if (g.nsplit > 1) {                                       // the runs write fp32 partials behind the packed weights
        const size_t pack_bytes = (size_t)g.ktiles * g.nphase * g.nstage * pk.ntap * 2 * g.kt * 16;
        out = reinterpret_cast<unsigned char*>(Wp) + align_up(pack_bytes, 256);
        out_bf16 = 0;
    }
    profile_mark_start(st, 4);
    if (g.kt == 64 && g.ptile == 2 * CB_P) {
        if (out_bf16) cb_launch_kernel<MODE, 64, 2 * CB_P, bf16_t>(g, in, Wp, zero_page, out, grid, smem, st);
        else cb_launch_kernel<MODE, 64, 2 * CB_P, float>(g, in, Wp, zero_page, out, grid, smem, st);
    } else if (g.kt == 64) {
        if (out_bf16) cb_launch_kernel<MODE, 64, CB_P, bf16_t>(g, in, Wp, zero_page, out, grid, smem, st);
        else cb_launch_kernel<MODE, 64, CB_P, float>(g, in, Wp, zero_page, out, grid, smem, st);
    } else {
        if (out_bf16) cb_launch_kernel<MODE, 128, CB_P, bf16_t>(g, in, Wp, zero_page, out, grid, smem, st);
        else cb_launch_kernel<MODE, 128, CB_P, float>(g, in, Wp, zero_page, out, grid, smem, st);
    }
    const double outs = (double)g.B * g.Hout * g.Wout;
    if (g.nsplit > 1) {
        if (int rc = check_launch("conv_bf16_kernel")) return rc;
        const size_t n = (size_t)g.B * g.K * g.Hout * g.Wout;          // a multiple of 4: Wout is
        if (final_bf16) cb_split_reduce_kernel<bf16_t><<<(unsigned)cdiv(n / 4, 256), 256, 0, st>>>(static_cast<const float*>(out), g.nsplit, n / 4, static_cast<bf16_t*>(final_out));
        else cb_split_reduce_kernel<float><<<(unsigned)cdiv(n / 4, 256), 256, 0, st>>>(static_cast<const float*>(out), g.nsplit, n / 4, static_cast<float*>(final_out));
    }
    profile_mark_stop(st, 4, 2.0 * taps_per_out * g.C * (double)(g.ktiles * g.kt) * outs, 2.0 * taps_per_out * g.C * (double)g.K * outs);
    return check_launch(g.nsplit > 1 ? "cb_split_reduce_kernel" : "conv_bf16_kernel");
}

// in [B,C,H,W] bf16, weight fp32 with element (c, k, tap) at w[c*sc + k*sk + tap] (taps flipped when `flip`), out [B,K,H,W] bf16 / fp32
int launch_conv_bf16(const void* in, const float* w, void* out, int B, int C, int K, int H, int W, long sc, long sk, int flip, int out_bf16,
                     void* ws, size_t ws_bytes, hipStream_t st, int pack_valid = 0)
{
    CbGeom g;
    if (int rc = cb_geometry(B, C, K, H, W, &g)) return rc;
    const size_t need = conv_bf16_ws_bytes(B, C, K, H, W);
    if (ws_bytes < need) return fail(IPSR_ERR_WORKSPACE, "bf16 direct conv: workspace %zu < %zu", ws_bytes, need);
    CbPack pk{};
    pk.ntap = 9; pk.nsub = 1; pk.nphase = 1;
    for (int t = 0; t < 9; ++t) pk.srctap[0][0][t] = flip ? 8 - t : t;
    return cb_launch<CB_S1>(g, pk, in, w, out, sc, sk, out_bf16, ws, st, 9.0, pack_valid);
}

// k4 s2 p1.  weight [Kc][Cf][4][4]: element (coarse channel kc, fine channel cf, r, s) at w[kc * skc + cf * scf + r * 4 + s].
// form 0 (fine -> coarse): in = fine [B,Cf,2nh,2nw], out = coarse [B,Kc,nh,nw].   form 1 (coarse -> fine): in = coarse, out = fine.
int launch_conv_bf16_s2(int form, const void* in, const float* w, void* out, int B, int Kc, int Cf, int nh, int nw, long skc, long scf,
                        int out_bf16, void* ws, size_t ws_bytes, hipStream_t st)
{
    CbGeom g;
    const int C = form == 0 ? Cf : Kc, K = form == 0 ? Kc : Cf;
    if (int rc = cb_geometry_s2(form, B, C, K, nh, nw, &g)) return rc;
    const size_t need = conv_bf16_s2_ws_bytes(form, B, C, K, nh, nw);
    if (ws_bytes < need) return fail(IPSR_ERR_WORKSPACE, "bf16 direct 4x4 stride-2 conv: workspace %zu < %zu", ws_bytes, need);
    CbPack pk{};
    pk.ntap = 8;
    if (form == 0) {
        pk.nsub = 2; pk.nphase = 1;
        for (int e = 0; e < 2; ++e)
            for (int t = 0; t < 8; ++t) {
                const int ri = t >> 2, sx = t & 3, rr = e ? 2 * ri : 2 * ri + 1;
                pk.srctap[0][e][t] = rr * 4 + sx;
            }
        return cb_launch<CB_F2C>(g, pk, in, w, out, scf, skc, out_bf16, ws, st, 16.0);
    }
    pk.nsub = 1; pk.nphase = 2;
    for (int ey = 0; ey < 2; ++ey)
        for (int t = 0; t < 8; ++t) {
            const int ex = t >> 2, ai = (t >> 1) & 1, bi = t & 1;
            const int rr = ey == 0 ? (ai ? 3 : 1) : (ai ? 2 : 0), ss = ex == 0 ? (bi ? 3 : 1) : (bi ? 2 : 0);
            pk.srctap[ey][0][t] = rr * 4 + ss;
        }
    return cb_launch<CB_C2F>(g, pk, in, w, out, skc, scf, out_bf16, ws, st, 4.0);
}

// =====================================================================================================================================
// Weight gradient of the k3 s1 p1 layers on the bf16 matrix cores:
//      dW[ka][cb][t] = sum_{b, y, x}  a[b][ka][y][x] * w[b][cb][y + dy_t][x + dx_t]            (w zero outside the image)
// Conv2d: a = dy (Ka = Cout), w = x (Cb = Cin);  ConvTranspose2d: a = x (Ka = Cin), w = dy (Cb = Cout) — dW comes out in the module's
// own layout either way.  As a GEMM the reduction runs over PIXELS, which NCHW has contiguous: both operands' fragments are 8
// consecutive pixels of one channel, i.e. aligned 16-byte reads of the natural image — except the dx = +-1 taps, whose windows start
// one pixel (2 bytes) off.  A lane therefore reads its aligned chunk plus the dword before and after it and builds the two shifted
// fragments with five v_alignbit_b32 (the three taps of a row share them).
// Workgroup = 512 threads = 8 waves (4 x 2): 128 a-channels x 64 w-channels x 9 taps; a wave owns 32 x 32 x 9 = nine MFMA tiles.
// The pixel range is cut over workgroups (`nsplit` runs of whole image rows): every workgroup writes its partial [t][ka][cb] slab and
// a second small kernel adds the slabs in ascending order (deterministic) into dW[ka][cb][t].  Stage = 128 pixels (RS = 128 / W image
// rows): the a rows [128][128 px] double-buffered, the w rows in a ring of 2 RS + 2 image rows (a stage needs RS + 2, the next one's
// RS new rows arrive meanwhile), both by LDS-DMA with the 16-byte chunks of a row XOR-swizzled / the row pitch odd in 16-byte slots so
// that the 32 channels of a fragment read hit distinct banks.
constexpr int WB_K = 128, WB_C = 64, WB_THREADS = 512, WB_PX = 128;
constexpr int WB_A_BYTES = WB_K * WB_PX * 2;                 // 32 KB per buffer
constexpr int WB_X_BYTES = 96 * 1024;                        // the w-row ring (largest: W = 16 -> 18 rows x 64 c x 5 slots x 16 B = 92 KB)

struct WbGeom {
    int B, Ka, Cb, H, W, wshift;
    int RS, NSLOT, pitch;       // image rows per stage, ring rows (2 RS + 2), 16-byte slots per (row, channel): W / 8 + 3
    int stages_per_wg, nsplit;  // stages of RS rows a workgroup reduces; B * H / (RS * stages_per_wg) runs
    int ktiles, ctiles;
};

// LDS reads that run beside an LDS-DMA use ext-vector types: a HIP_vector_type (uint4) load is a struct copy that reaches the back end
// without alias metadata, and hipcc then drains the DMA queue (s_waitcnt vmcnt(0)) in front of it — the prefetch stops overlapping.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned alignbit16(unsigned hi, unsigned lo) { return __builtin_amdgcn_alignbit(hi, lo, 16); }

__global__ void __launch_bounds__(WB_THREADS, 1) conv_bf16_wrw_kernel(const unsigned short* __restrict__ a, const unsigned short* __restrict__ w,
                                                                      const uint4* __restrict__ zero_page, WbGeom g, float* __restrict__ slabs)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];          // A[2] | X ring
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave >> 1, wc = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles = g.ktiles * g.ctiles;
    const int tile = L % tiles, split = L / tiles;
    const int kt = tile % g.ktiles, ct = tile / g.ktiles;
    const int rows_per_wg = g.RS * g.stages_per_wg;
    const int runs_per_img = g.H / rows_per_wg;
    const int b = split / runs_per_img, ylo = (split - b * runs_per_img) * rows_per_wg;
    const size_t HW = (size_t)g.H * g.W;
    const int cpr = g.W >> 3;                                // 16-byte chunks per image row

    // ---- a rows: slot sigma = k * 16 + cs holds stage chunk c8 = cs ^ (k & 15) of channel k; c8 -> (row rs, chunk cx) -------------------
    const unsigned short* ga[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int sigma = (wave + 8 * j) * 64 + lane;
        const int k = sigma >> 4, c8 = (sigma & 15) ^ (k & 15);
        const int rs = c8 / cpr, cx = c8 - rs * cpr;
        const int ka = kt * WB_K + k;
        ga[j] = ka < g.Ka ? a + ((size_t)b * g.Ka + ka) * HW + (size_t)(ylo + rs) * g.W + cx * 8 : nullptr;
    }
    // ---- w rows: ring slot of image row y = (y + 1) mod NSLOT; per (ring row, channel): [halo][W / 8 chunks][halo][pad] = `pitch` slots ----
    // a group of RS rows = RS * 64 * pitch slots; slot q = (row rr, channel c, slot sl); 64 * pitch is a multiple of 64, so the 64 lanes of
    // one DMA instruction never straddle rows: wave-uniform LDS base, per-lane source
    const int xslots = g.RS * WB_C * g.pitch;
    const unsigned short* gxw[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int q = tid + WB_THREADS * j;
        const int qq = q < xslots ? q : 0;
        const int sl = qq % g.pitch, c = (qq / g.pitch) % WB_C;
        const int cb = ct * WB_C + c;
        const bool data = sl >= 1 && sl <= cpr && cb < g.Cb;
        gxw[j] = data ? w + ((size_t)b * g.Cb + cb) * HW + (sl - 1) * 8 : nullptr;
    }
    const int ring_row_slots = WB_C * g.pitch, ring_row_bytes = ring_row_slots * 16;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;

    auto dma_a = [&](int buf, int stage) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const void* src = ga[j] ? static_cast<const void*>(ga[j] + (size_t)stage * g.RS * g.W) : static_cast<const void*>(zero_page);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + buf * WB_A_BYTES + (wave + 8 * j) * 1024), 16, 0, 0);
        }
    };
    // image rows y_first .. y_first + RS - 1 into their ring slots (rows outside the image: zeros)
    auto dma_x = [&](int y_first) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int q0 = wave * 64 + WB_THREADS * j;        // uniform: first slot of this wave's instruction
            if (q0 < xslots) {
                const int rr = q0 / ring_row_slots, within = q0 - rr * ring_row_slots;
                const int y = y_first + rr;
                int slot = (y + 1) % g.NSLOT;
                if (slot < 0) slot += g.NSLOT;
                const bool ok = gxw[j] != nullptr && (unsigned)y < (unsigned)g.H;
                const void* src = ok ? static_cast<const void*>(gxw[j] + (size_t)y * g.W) : static_cast<const void*>(zero_page);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + 2 * WB_A_BYTES + slot * ring_row_bytes + within * 16), 16, 0, 0);
            }
        }
    };

    // prologue: a stage 0; w rows ylo - 1 .. ylo + RS (at least)
    dma_a(0, 0);
    for (int y = ylo - 1; y <= ylo + g.RS; y += g.RS) dma_x(y);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int a_row = (wk * 32 + r);                          // a fragment: channel row
    const int bcol = wc * 32 + r;                             // w fragment: channel column
    for (int s = 0; s < g.stages_per_wg; ++s) {
        const int cur = s & 1;
        const int y0 = ylo + s * g.RS;
        if (s + 1 < g.stages_per_wg) {
            dma_a(cur ^ 1, s + 1);
            dma_x(y0 + g.RS + 1);                             // the RS rows the next stage adds: y0 + RS + 1 .. y0 + 2 RS
        }
        const unsigned char* A = lds + cur * WB_A_BYTES;
#pragma unroll
        for (int j = 0; j < 8; ++j) {                         // k-steps of 16 pixels
            const int c8 = 2 * j + h;
            const bf16x8 fa = *reinterpret_cast<const bf16x8*>(A + ((a_row << 4) + (c8 ^ (a_row & 15))) * 16);
            const int p0 = 16 * j;                            // stage pixel of the k-step
            const int rs = p0 >> g.wshift, px = (p0 & (g.W - 1)) + 8 * h;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                int slot = (y0 + rs + dy) % g.NSLOT;          // image row y0 + rs + dy - 1 lives in ring slot (y + 1) mod NSLOT
                const unsigned char* X = lds + 2 * WB_A_BYTES + slot * ring_row_bytes + (bcol * g.pitch + 1 + (px >> 3)) * 16;
                const u32x4 c4 = *reinterpret_cast<const u32x4*>(X);    // ext-vector load (see u32x4)
                const unsigned prev = *reinterpret_cast<const unsigned*>(X - 4);
                const unsigned next = *reinterpret_cast<const unsigned*>(X + 16);
                const unsigned s0 = alignbit16(c4.x, prev), s1 = alignbit16(c4.y, c4.x), s2 = alignbit16(c4.z, c4.y), s3 = alignbit16(c4.w, c4.z),
                               s4 = alignbit16(next, c4.w);
                const uint4 left = make_uint4(s0, s1, s2, s3), right = make_uint4(s1, s2, s3, s4);
                acc[dy * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, __builtin_bit_cast(bf16x8, left), acc[dy * 3 + 0], 0, 0, 0);
                acc[dy * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, __builtin_bit_cast(bf16x8, c4), acc[dy * 3 + 1], 0, 0, 0);
                acc[dy * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, __builtin_bit_cast(bf16x8, right), acc[dy * 3 + 2], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // partial result: slab[split][t][ka][cb] (lanes along cb: coalesced)
    const int Kap = g.ktiles * WB_K, Cbp = g.ctiles * WB_C;
    float* out = slabs + (size_t)split * 9 * Kap * Cbp;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int ka = kt * WB_K + wk * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            out[((size_t)t * Kap + ka) * Cbp + ct * WB_C + bcol] = acc[t][e];
        }
}

// dW[ka][cb][t] = sum_s slab[s][t][ka][cb]
__global__ void __launch_bounds__(256) conv_bf16_wrw_reduce_kernel(const float* __restrict__ slabs, int nsplit, int Ka, int Cb, int Kap, int Cbp,
                                                                   float* __restrict__ dW)
{
    const int cb = blockIdx.x * 256 + threadIdx.x, ka = blockIdx.y;
    if (cb >= Cb) return;
    float o[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) o[t] = 0.0f;
    const size_t slab = (size_t)9 * Kap * Cbp;
    for (int s = 0; s < nsplit; ++s)
#pragma unroll
        for (int t = 0; t < 9; ++t) o[t] += slabs[(size_t)s * slab + ((size_t)t * Kap + ka) * Cbp + cb];
    float* d = dW + ((size_t)ka * Cb + cb) * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) d[t] = o[t];
}

static int wb_geometry(int B, int Ka, int Cb, int H, int W, WbGeom* g)
{
    if (W != 16 && W != 32 && W != 64 && W != 128) return fail(IPSR_ERR_UNSUPPORTED, "bf16 weight gradient: image width %d (16, 32, 64 or 128)", W);
    const int RS = WB_PX / W;
    if (H % RS != 0) return fail(IPSR_ERR_UNSUPPORTED, "bf16 weight gradient: %d rows are not a multiple of %d", H, RS);
    g->B = B; g->Ka = Ka; g->Cb = Cb; g->H = H; g->W = W;
    g->wshift = W == 16 ? 4 : (W == 32 ? 5 : (W == 64 ? 6 : 7));
    g->RS = RS; g->NSLOT = 2 * RS + 2; g->pitch = W / 8 + 3;
    g->ktiles = (Ka + WB_K - 1) / WB_K; g->ctiles = (Cb + WB_C - 1) / WB_C;
    if (g->NSLOT * WB_C * g->pitch * 16 > WB_X_BYTES || RS * WB_C * g->pitch > 5 * WB_THREADS)
        return fail(IPSR_ERR_UNSUPPORTED, "bf16 weight gradient: the row ring of a %d-wide image does not fit the LDS plan", W);
    // runs: ONE round of one workgroup per CU — every run costs a 295-KB partial slab written and read back (PMC, 128 -> 128 @128x128:
    // 512 runs = 151 MB each way against 134 MB of operands)
    const int groups = H / RS;                                // stages per image
    int spw = (int)(((long)g->ktiles * g->ctiles * B * groups + 255) / 256);
    if (spw < 1) spw = 1;
    if (spw > groups) spw = groups;
    while (groups % spw) --spw;
    g->stages_per_wg = spw;
    g->nsplit = B * (groups / spw);
    return IPSR_OK;
}

size_t conv_bf16_wrw_ws_bytes(int B, int Ka, int Cb, int H, int W)
{
    WbGeom g;
    if (wb_geometry(B, Ka, Cb, H, W, &g) != IPSR_OK) return 0;
    return 256 + (size_t)g.nsplit * 9 * g.ktiles * WB_K * g.ctiles * WB_C * 4;
}

// a [B,Ka,H,W], w [B,Cb,H,W] bf16 -> dW [Ka][Cb][3][3] fp32
int launch_conv_bf16_wrw(const void* a, const void* w, float* dW, int B, int Ka, int Cb, int H, int W, void* ws, size_t ws_bytes, hipStream_t st)
{
    WbGeom g;
    if (int rc = wb_geometry(B, Ka, Cb, H, W, &g)) return rc;
    const size_t need = conv_bf16_wrw_ws_bytes(B, Ka, Cb, H, W);
    if (ws_bytes < need) return fail(IPSR_ERR_WORKSPACE, "bf16 weight gradient: workspace %zu < %zu", ws_bytes, need);
    uint4* zero_page = static_cast<uint4*>(ws);
    float* slabs = reinterpret_cast<float*>(zero_page + 16);
    if (hipMemsetAsync(zero_page, 0, 64, st) != hipSuccess) return fail(IPSR_ERR_LAUNCH, "bf16 weight gradient: hipMemsetAsync failed");
    const size_t smem = 2 * (size_t)WB_A_BYTES + (size_t)g.NSLOT * WB_C * g.pitch * 16;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_wrw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WB_A_BYTES + WB_X_BYTES); attr = true; }
    const unsigned grid = (unsigned)(g.ktiles * g.ctiles * g.nsplit);
    profile_mark_start(st, 4);
    conv_bf16_wrw_kernel<<<grid, WB_THREADS, smem, st>>>(static_cast<const unsigned short*>(a), static_cast<const unsigned short*>(w), zero_page, g, slabs);
    profile_mark_stop(st, 4, 2.0 * 9.0 * (double)(g.ktiles * WB_K) * (g.ctiles * WB_C) * B * H * W, 2.0 * 9.0 * (double)Ka * Cb * B * H * W);
    if (int rc = check_launch("conv_bf16_wrw_kernel")) return rc;
    conv_bf16_wrw_reduce_kernel<<<dim3(cdiv(Cb, 256), Ka), 256, 0, st>>>(slabs, g.nsplit, Ka, Cb, g.ktiles * WB_K, g.ctiles * WB_C, dW);
    return check_launch("conv_bf16_wrw_reduce_kernel");
}

// =====================================================================================================================================
// Weight gradient of the k4 s2 p1 layers (Conv2d [Kc][Cf] and ConvTranspose2d [Kc][Cf] alike, in the coarse / fine terms above):
//      dW[kc][cf][r][s] = sum_{b, oy, ox}  coarse[b][kc][oy][ox] * fine[b][cf][2 oy - 1 + r][2 ox - 1 + s]
// The reduction runs over COARSE pixels: the coarse operand's fragments are aligned 16-byte reads as in the k3 kernel; the fine
// operand's 8 consecutive reduction elements are every OTHER pixel of a fine row: a lane reads the 16 fine pixels they span (two
// aligned 16-byte reads + the dword before and after) and picks the even / odd halves with four v_perm_b32 per column tap.
// Workgroup = 8 waves = 4 (kc) x 2 (row taps r in {0,1} / {2,3}): 128 kc x 32 cf x 16 taps, a wave 32 x 32 x 8 taps; stage = 64 coarse
// pixels (RS = 64 / nw coarse rows); the fine rows live in a ring of 4 RS + 2 image rows handled in PAIRS (row 2 i - 1 and 2 i:
// 64 x pitch slots, so a DMA instruction never straddles ring entries); runs of coarse rows are cut over workgroups, the partial
// [t][kc][cf] slabs added in order by the second launch.
constexpr int W2_K = 128, W2_C = 32, W2_PX = 64;
constexpr int W2_A_BYTES = W2_K * W2_PX * 2;                 // 16 KB per buffer

struct W2Geom {
    int B, Kc, Cf, nh, nw, wshift;      // coarse grid nh x nw (fine 2nh x 2nw)
    int RS, NPAIR, pitch;               // coarse rows per stage, ring entries (pairs of fine rows: 2 RS + 1), slots per (fine row, channel)
    int stages_per_wg, nsplit, ktiles, ctiles;
};

__device__ __forceinline__ unsigned pack_hi(unsigned x, unsigned y) { return __builtin_amdgcn_perm(y, x, 0x07060302u); }   // {x.hi16, y.hi16}
__device__ __forceinline__ unsigned pack_lo(unsigned x, unsigned y) { return __builtin_amdgcn_perm(y, x, 0x05040100u); }   // {x.lo16, y.lo16}

__global__ void __launch_bounds__(512, 1) conv_bf16_wrw_s2_kernel(const unsigned short* __restrict__ coarse, const unsigned short* __restrict__ fine,
                                                                   const uint4* __restrict__ zero_page, W2Geom g, float* __restrict__ slabs)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];          // A[2] | fine-row ring
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave >> 1, rh = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles = g.ktiles * g.ctiles;
    const int tile = L % tiles, split = L / tiles;
    const int kt = tile % g.ktiles, ct = tile / g.ktiles;
    const int rows_per_wg = g.RS * g.stages_per_wg;
    const int runs_per_img = g.nh / rows_per_wg;
    const int b = split / runs_per_img, ylo = (split - b * runs_per_img) * rows_per_wg;
    const int Hf = 2 * g.nh, Wf = 2 * g.nw;
    const size_t HWc = (size_t)g.nh * g.nw, HWf = (size_t)Hf * Wf;
    const int cprc = g.nw >> 3;                              // 16-byte chunks per coarse row
    const int cprf = Wf >> 3;                                // ... per fine row

    // ---- coarse rows: slot sigma = k * 8 + cs holds stage chunk c8 = cs ^ (k & 7) of channel k; c8 -> (coarse row rs, chunk cx) ----------
    const unsigned short* ga[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int sigma = (wave + 8 * j) * 64 + lane;
        const int k = sigma >> 3, c8 = (sigma & 7) ^ (k & 7);
        const int rs = c8 / cprc, cx = c8 - rs * cprc;
        const int kc = kt * W2_K + k;
        ga[j] = kc < g.Kc ? coarse + ((size_t)b * g.Kc + kc) * HWc + (size_t)(ylo + rs) * g.nw + cx * 8 : nullptr;
    }
    // ---- fine rows: ring entry of the row pair (2 i - 1, 2 i) = i mod NPAIR; per (row, channel): [halo][Wf / 8 chunks][halo][pad] ---------
    const int pair_slots = 2 * W2_C * g.pitch;               // a multiple of 64
    const unsigned short* gf[5];
    int f_row[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int q = tid + 512 * j;
        const int within = q % pair_slots;
        const int pr = within / (W2_C * g.pitch), rem = within - pr * (W2_C * g.pitch);
        const int c = rem / g.pitch, sl = rem - c * g.pitch;
        const int cf = ct * W2_C + c;
        f_row[j] = 2 * (q / pair_slots) + pr;                // fine row offset from the first row of the group
        const bool data = sl >= 1 && sl <= cprf && cf < g.Cf;
        gf[j] = data ? fine + ((size_t)b * g.Cf + cf) * HWf + (sl - 1) * 8 : nullptr;
    }
    const int pair_bytes = pair_slots * 16;

    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;

    auto dma_a = [&](int buf, int stage) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const void* src = ga[j] ? static_cast<const void*>(ga[j] + (size_t)stage * g.RS * g.nw) : static_cast<const void*>(zero_page);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + buf * W2_A_BYTES + (wave + 8 * j) * 1024), 16, 0, 0);
        }
    };
    // `npairs` row pairs starting with the pair (2 i0 - 1, 2 i0)
    auto dma_f = [&](int i0, int npairs) {
        const int nslots = npairs * pair_slots;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int q0 = wave * 64 + 512 * j;              // uniform
            if (q0 < nslots) {
                const int pp = q0 / pair_slots, within = q0 - pp * pair_slots;
                int entry = (i0 + pp) % g.NPAIR;
                if (entry < 0) entry += g.NPAIR;
                const int yf = 2 * i0 - 1 + f_row[j];
                const bool ok = gf[j] != nullptr && (unsigned)yf < (unsigned)Hf;
                const void* src = ok ? static_cast<const void*>(gf[j] + (size_t)yf * Wf) : static_cast<const void*>(zero_page);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + 2 * W2_A_BYTES + entry * pair_bytes + within * 16), 16, 0, 0);
            }
        }
    };

    // a stage on coarse rows y0 .. y0 + RS - 1 reads the fine rows 2 y0 - 1 .. 2 (y0 + RS - 1) + 2 = the pairs y0 .. y0 + RS
    dma_a(0, 0);
    dma_f(ylo, g.RS + 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int a_row = wk * 32 + r;
    const int row_bytes = W2_C * g.pitch * 16;
    for (int s = 0; s < g.stages_per_wg; ++s) {
        const int cur = s & 1;
        const int y0 = ylo + s * g.RS;
        if (s + 1 < g.stages_per_wg) {
            dma_a(cur ^ 1, s + 1);
            dma_f(y0 + g.RS + 1, g.RS);                       // the pairs the next stage adds
        }
        const unsigned char* A = lds + cur * W2_A_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {                         // k-steps of 16 coarse pixels
            const int c8 = 2 * j + h;
            const bf16x8 fa = *reinterpret_cast<const bf16x8*>(A + ((a_row << 3) + (c8 ^ (a_row & 7))) * 16);
            const int p0 = 16 * j;
            const int rs = p0 >> g.wshift, ox0 = (p0 & (g.nw - 1)) + 8 * h;
#pragma unroll
            for (int ri = 0; ri < 2; ++ri) {
                // fine row 2 (y0 + rs) - 1 + (2 rh + ri) = row (1 - ...) of a pair: row index u = 2 (y0 + rs) + 2 rh + ri  ->  pair u / 2, member u & 1
                const int u = 2 * (y0 + rs) + 2 * rh + ri;    // = fine row + 1
                const int entry = (u >> 1) % g.NPAIR;
                const unsigned char* X = lds + 2 * W2_A_BYTES + entry * pair_bytes + (u & 1) * row_bytes + (r * g.pitch + 1 + (ox0 >> 2)) * 16;
                const u32x4 lo4 = *reinterpret_cast<const u32x4*>(X);
                const u32x4 hi4 = *reinterpret_cast<const u32x4*>(X + 16);
                const unsigned prev = *reinterpret_cast<const unsigned*>(X - 4);
                const unsigned next = *reinterpret_cast<const unsigned*>(X + 32);
                const unsigned d0 = lo4.x, d1 = lo4.y, d2 = lo4.z, d3 = lo4.w, d4 = hi4.x, d5 = hi4.y, d6 = hi4.z, d7 = hi4.w;
                const uint4 f0 = make_uint4(pack_hi(prev, d0), pack_hi(d1, d2), pack_hi(d3, d4), pack_hi(d5, d6));
                const uint4 f1 = make_uint4(pack_lo(d0, d1), pack_lo(d2, d3), pack_lo(d4, d5), pack_lo(d6, d7));
                const uint4 f2 = make_uint4(pack_hi(d0, d1), pack_hi(d2, d3), pack_hi(d4, d5), pack_hi(d6, d7));
                const uint4 f3 = make_uint4(pack_lo(d1, d2), pack_lo(d3, d4), pack_lo(d5, d6), pack_lo(d7, next));
                acc[ri * 4 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, __builtin_bit_cast(bf16x8, f0), acc[ri * 4 + 0], 0, 0, 0);
                acc[ri * 4 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, __builtin_bit_cast(bf16x8, f1), acc[ri * 4 + 1], 0, 0, 0);
                acc[ri * 4 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, __builtin_bit_cast(bf16x8, f2), acc[ri * 4 + 2], 0, 0, 0);
                acc[ri * 4 + 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, __builtin_bit_cast(bf16x8, f3), acc[ri * 4 + 3], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // partial result: slab[split][t = r * 4 + s][kc][cf] (lanes along cf)
    const int Kp = g.ktiles * W2_K, Cp = g.ctiles * W2_C;
    float* out = slabs + (size_t)split * 16 * Kp * Cp;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int kc = kt * W2_K + wk * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            out[((size_t)(rh * 8 + t) * Kp + kc) * Cp + ct * W2_C + r] = acc[t][e];
        }
}

// dW[kc][cf][t] = sum_s slab[s][t][kc][cf], t = 0..15
__global__ void __launch_bounds__(256) conv_bf16_wrw_s2_reduce_kernel(const float* __restrict__ slabs, int nsplit, int Kc, int Cf, int Kp, int Cp,
                                                                      float* __restrict__ dW)
{
    const int cf = blockIdx.x * 256 + threadIdx.x, kc = blockIdx.y;
    if (cf >= Cf) return;
    float o[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) o[t] = 0.0f;
    const size_t slab = (size_t)16 * Kp * Cp;
    for (int s = 0; s < nsplit; ++s)
#pragma unroll
        for (int t = 0; t < 16; ++t) o[t] += slabs[(size_t)s * slab + ((size_t)t * Kp + kc) * Cp + cf];
    float4* d = reinterpret_cast<float4*>(dW + ((size_t)kc * Cf + cf) * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = make_float4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
}

static int w2_geometry(int B, int Kc, int Cf, int nh, int nw, W2Geom* g)
{
    if (nw != 16 && nw != 32 && nw != 64) return fail(IPSR_ERR_UNSUPPORTED, "bf16 4x4 stride-2 weight gradient: coarse width %d (16, 32 or 64)", nw);
    const int RS = W2_PX / nw;
    if (nh % RS != 0) return fail(IPSR_ERR_UNSUPPORTED, "bf16 4x4 stride-2 weight gradient: %d coarse rows are not a multiple of %d", nh, RS);
    g->B = B; g->Kc = Kc; g->Cf = Cf; g->nh = nh; g->nw = nw;
    g->wshift = nw == 16 ? 4 : (nw == 32 ? 5 : 6);
    g->RS = RS; g->NPAIR = 2 * RS + 1; g->pitch = 2 * nw / 8 + 3;
    g->ktiles = (Kc + W2_K - 1) / W2_K; g->ctiles = (Cf + W2_C - 1) / W2_C;
    if ((RS + 1) * 2 * W2_C * g->pitch > 5 * 512) return fail(IPSR_ERR_UNSUPPORTED, "bf16 4x4 stride-2 weight gradient: row ring of a %d-wide grid", nw);
    const int groups = nh / RS;
    int spw = (int)(((long)g->ktiles * g->ctiles * B * groups + 255) / 256);
    if (spw < 1) spw = 1;
    if (spw > groups) spw = groups;
    while (groups % spw) --spw;
    g->stages_per_wg = spw;
    g->nsplit = B * (groups / spw);
    return IPSR_OK;
}

size_t conv_bf16_wrw_s2_ws_bytes(int B, int Kc, int Cf, int nh, int nw)
{
    W2Geom g;
    if (w2_geometry(B, Kc, Cf, nh, nw, &g) != IPSR_OK) return 0;
    return 256 + (size_t)g.nsplit * 16 * g.ktiles * W2_K * g.ctiles * W2_C * 4;
}

// fine [B,Cf,2nh,2nw], coarse [B,Kc,nh,nw] bf16 -> dW [Kc][Cf][4][4] fp32
int launch_conv_bf16_wrw_s2(const void* fine, const void* coarse, float* dW, int B, int Kc, int Cf, int nh, int nw, void* ws, size_t ws_bytes, hipStream_t st)
{
    W2Geom g;
    if (int rc = w2_geometry(B, Kc, Cf, nh, nw, &g)) return rc;
    const size_t need = conv_bf16_wrw_s2_ws_bytes(B, Kc, Cf, nh, nw);
    if (ws_bytes < need) return fail(IPSR_ERR_WORKSPACE, "bf16 4x4 stride-2 weight gradient: workspace %zu < %zu", ws_bytes, need);
    uint4* zero_page = static_cast<uint4*>(ws);
    float* slabs = reinterpret_cast<float*>(zero_page + 16);
    if (hipMemsetAsync(zero_page, 0, 64, st) != hipSuccess) return fail(IPSR_ERR_LAUNCH, "bf16 4x4 stride-2 weight gradient: hipMemsetAsync failed");
    const size_t smem = 2 * (size_t)W2_A_BYTES + (size_t)g.NPAIR * 2 * W2_C * g.pitch * 16;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_wrw_s2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, CB_LDS_MAX); attr = true; }
    if (smem > (size_t)CB_LDS_MAX) return fail(IPSR_ERR_UNSUPPORTED, "bf16 4x4 stride-2 weight gradient: %zu bytes of LDS", smem);
    const unsigned grid = (unsigned)(g.ktiles * g.ctiles * g.nsplit);
    profile_mark_start(st, 4);
    conv_bf16_wrw_s2_kernel<<<grid, 512, smem, st>>>(static_cast<const unsigned short*>(coarse), static_cast<const unsigned short*>(fine), zero_page, g, slabs);
    profile_mark_stop(st, 4, 2.0 * 16.0 * (double)(g.ktiles * W2_K) * (g.ctiles * W2_C) * B * nh * nw, 2.0 * 16.0 * (double)Kc * Cf * B * nh * nw);
    if (int rc = check_launch("conv_bf16_wrw_s2_kernel")) return rc;
    conv_bf16_wrw_s2_reduce_kernel<<<dim3(cdiv(Cf, 256), Kc), 256, 0, st>>>(slabs, g.nsplit, Kc, Cf, g.ktiles * W2_K, g.ctiles * W2_C, dW);
    return check_launch("conv_bf16_wrw_s2_reduce_kernel");
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
    return ipsr_conv3x3_bf16_packed(op, in, weight, out, B, Cin, H, W, Cout, out_bf16, 0, ws, ws_bytes, stream);
}

int ipsr_conv3x3_bf16_packed(int op, const void* in, const float* weight, void* out, int B, int Cin, int H, int W, int Cout, int out_bf16,
                             int pack_valid, void* ws, size_t ws_bytes, void* stream)
{
    if (!in || !weight || !out || !ws) return fail(IPSR_ERR_INVALID, "ipsr_conv3x3_bf16: null pointer");
    if (op < 0 || op > 3 || B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1) return fail(IPSR_ERR_INVALID, "ipsr_conv3x3_bf16: bad argument");
    if ((reinterpret_cast<uintptr_t>(ws) & 15u) || (reinterpret_cast<uintptr_t>(in) & 15u))
        return fail(IPSR_ERR_INVALID, "ipsr_conv3x3_bf16: in / workspace must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (op) {       // (sc, sk, flip) as in ipsr_conv3x3_winograd_mp: C = reduction channels, K = produced channels
        case 0: return launch_conv_bf16(in, weight, out, B, Cin, Cout, H, W, 9, (long)Cin * 9, 0, out_bf16, ws, ws_bytes, st, pack_valid);
        case 1: return launch_conv_bf16(in, weight, out, B, Cout, Cin, H, W, (long)Cin * 9, 9, 1, out_bf16, ws, ws_bytes, st, pack_valid);
        case 2: return launch_conv_bf16(in, weight, out, B, Cin, Cout, H, W, (long)Cout * 9, 9, 1, out_bf16, ws, ws_bytes, st, pack_valid);
        default: return launch_conv_bf16(in, weight, out, B, Cout, Cin, H, W, 9, (long)Cout * 9, 0, out_bf16, ws, ws_bytes, st, pack_valid);
    }
}

size_t ipsr_conv4x4s2_bf16_workspace_bytes(int mode, int B, int Kc, int Cf, int nh, int nw)
{
    if (mode < 0 || mode > 1 || B < 1 || Kc < 1 || Cf < 1 || nh < 1 || nw < 1) return 0;
    return conv_bf16_s2_ws_bytes(mode, B, mode == 0 ? Cf : Kc, mode == 0 ? Kc : Cf, nh, nw);
}

int ipsr_conv4x4s2_bf16(int mode, const void* in, const float* weight, void* out, int B, int Kc, int Cf, int nh, int nw, int out_bf16,
                        void* ws, size_t ws_bytes, void* stream)
{
    if (!in || !weight || !out || !ws) return fail(IPSR_ERR_INVALID, "ipsr_conv4x4s2_bf16: null pointer");
    if (mode < 0 || mode > 1 || B < 1 || Kc < 1 || Cf < 1 || nh < 1 || nw < 1) return fail(IPSR_ERR_INVALID, "ipsr_conv4x4s2_bf16: bad argument");
    if ((reinterpret_cast<uintptr_t>(ws) & 15u) || (reinterpret_cast<uintptr_t>(in) & 15u) || (reinterpret_cast<uintptr_t>(out) & 7u))
        return fail(IPSR_ERR_INVALID, "ipsr_conv4x4s2_bf16: in / out / workspace must be 16-byte aligned");
    // weight [Kc][Cf][4][4] in both modules (Conv2d: [Cout][Cin], ConvTranspose2d: [Cin][Cout]), as in ipsr_conv4x4s2_winograd
    return launch_conv_bf16_s2(mode, in, weight, out, B, Kc, Cf, nh, nw, (long)Cf * 16, 16, out_bf16, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

size_t ipsr_conv4x4s2_bf16_wrw_workspace_bytes(int B, int Kc, int Cf, int nh, int nw)
{
    if (B < 1 || Kc < 1 || Cf < 1 || nh < 1 || nw < 1) return 0;
    return conv_bf16_wrw_s2_ws_bytes(B, Kc, Cf, nh, nw);
}

int ipsr_conv4x4s2_bf16_wrw(const void* fine, const void* coarse, float* dw, int B, int Kc, int Cf, int nh, int nw, void* ws, size_t ws_bytes, void* stream)
{
    if (!fine || !coarse || !dw || !ws) return fail(IPSR_ERR_INVALID, "ipsr_conv4x4s2_bf16_wrw: null pointer");
    if (B < 1 || Kc < 1 || Cf < 1 || nh < 1 || nw < 1) return fail(IPSR_ERR_INVALID, "ipsr_conv4x4s2_bf16_wrw: bad argument");
    if ((reinterpret_cast<uintptr_t>(ws) & 15u) || (reinterpret_cast<uintptr_t>(fine) & 15u) || (reinterpret_cast<uintptr_t>(coarse) & 15u) ||
        (reinterpret_cast<uintptr_t>(dw) & 15u))
        return fail(IPSR_ERR_INVALID, "ipsr_conv4x4s2_bf16_wrw: operands / workspace must be 16-byte aligned");
    return launch_conv_bf16_wrw_s2(fine, coarse, dw, B, Kc, Cf, nh, nw, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

size_t ipsr_conv3x3_bf16_wrw_workspace_bytes(int transposed, int B, int Cin, int H, int W, int Cout)
{
    if (B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1) return 0;
    return transposed ? conv_bf16_wrw_ws_bytes(B, Cin, Cout, H, W) : conv_bf16_wrw_ws_bytes(B, Cout, Cin, H, W);
}

int ipsr_conv3x3_bf16_wrw(int transposed, const void* x, const void* dy, float* dw, int B, int Cin, int H, int W, int Cout,
                          void* ws, size_t ws_bytes, void* stream)
{
    if (!x || !dy || !dw || !ws) return fail(IPSR_ERR_INVALID, "ipsr_conv3x3_bf16_wrw: null pointer");
    if (B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1) return fail(IPSR_ERR_INVALID, "ipsr_conv3x3_bf16_wrw: bad argument");
    if ((reinterpret_cast<uintptr_t>(ws) & 15u) || (reinterpret_cast<uintptr_t>(x) & 15u) || (reinterpret_cast<uintptr_t>(dy) & 15u))
        return fail(IPSR_ERR_INVALID, "ipsr_conv3x3_bf16_wrw: x / dy / workspace must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    // Conv2d: dW[co][ci][t] = sum dy[co][p] x[ci][p + t];  ConvTranspose2d: dW[ci][co][t] = sum x[ci][p] dy[co][p + t]
    if (transposed) return launch_conv_bf16_wrw(x, dy, dw, B, Cin, Cout, H, W, ws, ws_bytes, st);
    return launch_conv_bf16_wrw(dy, x, dw, B, Cout, Cin, H, W, ws, ws_bytes, st);
}

}  // extern "C"
