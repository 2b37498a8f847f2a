// winograd.hip — 3x3 stride-1 pad-1 convolutions (and their input gradients / the transposed twins) by Winograd
// F(4x4, 3x3) on the fp32 matrix cores.
//
// Reference layers (models/vgg16.py:9-21 — all thirteen VGG convolutions; models/networks.py:220-243 — `downconv_3`
// Conv2d(k3 s1 p1) and `upconv_3` ConvTranspose2d(k3 s1 p1) of every netG level): together the largest share of the
// training step's time.  A direct fp32 implicit GEMM (conv_gemm.hip, ~100 TF) can at best tie MIOpen's F(2x2,3x3)
// Winograd assembly there (100-120 TF direct-equivalent); the way past it is the larger tile: F(4x4,3x3) needs 36
// multiplies per 16 outputs and channel pair instead of 144 — 4x fewer MFMA flops than the direct form.
//
//      Y = A^T [ (G g G^T) (.) (B^T d B) ] A          per 4x4 output tile, summed over input channels
//
// Non-fused, four launches (all operands of the multiply stage are then plain reduction-major matrices):
//   wino_filter_kernel   U[xi][c][k] = (G g G^T)[xi]                 from the weight tensor in any of the four roles
//                        (Conv2d / ConvTranspose2d, forward / backward-data: index strides + tap flip, as in conv_gemm.hip)
//   wino_input_kernel    V[xi][c][t] = (B^T d B)[xi]                 d = 6x6 input window of tile t = (b, ty, tx), zero padded
//   wino_gemm_kernel     M[xi][k][t] = sum_c U[xi][c][k] * V[xi][c][t]   36 independent GEMMs; both operands are [C][*]
//                        with the reduction outermost — exactly the layout the correlation kernel streams by LDS-DMA, so
//                        this is that kernel's pipeline (128x128 tile, 4-slot DMA ring, 32x32x2 fp32 MFMA) without the arg-max
//   wino_output_kernel   y[b][k][4ty+i][4tx+j] = (A^T M A)[i][j]
// HBM traffic is 2.25x the activations each way (36 numbers per 16 pixels) — worth it from 128 channels up, where the
// 4x smaller GEMM dominates; below that the direct kernels / MIOpen stay in charge (the Python dispatcher decides per shape).
//
// Numerics: the standard interpolation points (0, +-1, +-2, inf); fp32 error ~4e-6 of the output scale at 512 channels
// (tests/test_gpu_conv.py measures it per shape against an fp64 convolution).
#include <type_traits>

#include "ipsr_common.h"

namespace ipsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int WG_BM = 128, WG_BN = 128, WG_BK = 16, WG_NBUF = 4, WG_THREADS = 256;

// How the reduction of the 36 GEMMs is cut (wino_choose_split): the GEMMs xi < xi_split ("head") in `nsplit` ranges of `sps`
// stages, the others ("tail") in `nsplit_t` ranges of `sps_t`.  Range ks of GEMM xi writes slab ks of M
// (M + (ks*36 + xi) * plane); the output transforms add the slabs of each xi in ascending order (deterministic).
struct WinoSplit {
    int nsplit, sps, xi_split, nsplit_t, sps_t;
    int nxi = 36;           // GEMMs in the launch (36 Winograd points; 1 for the plain GEMMs of the small-map convolutions)
    __host__ __device__ int slabs() const { return xi_split >= nxi ? nsplit : (xi_split <= 0 ? nsplit_t : (nsplit > nsplit_t ? nsplit : nsplit_t)); }
    __host__ __device__ int of(int xi) const { return xi < xi_split ? nsplit : nsplit_t; }
    __host__ __device__ int head_xi() const { return xi_split < nxi ? xi_split : nxi; }
    __host__ int workgroups(int tiles_per_xi) const { return tiles_per_xi * (head_xi() * nsplit + (nxi - head_xi()) * nsplit_t); }
};

// m[i][j] = sum over the slabs of GEMM xi = 6i+j of M[slab][xi][off]
__device__ __forceinline__ void wino_load_sum(const float* __restrict__ M, const WinoSplit sp, size_t plane, size_t off, float m[6][6])
{
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) m[i][j] = M[(size_t)(i * 6 + j) * plane + off];
    // slabs every GEMM has: straight-line adds; the slabs only the finer-cut group has: predicated
    const int nmin = sp.xi_split >= 36 ? sp.nsplit : (sp.nsplit < sp.nsplit_t ? sp.nsplit : sp.nsplit_t), nmax = sp.slabs();
    for (int s = 1; s < nmin; ++s) {
        const float* Ms = M + (size_t)s * 36 * plane;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) m[i][j] += Ms[(size_t)(i * 6 + j) * plane + off];
    }
    for (int s = nmin < 1 ? 1 : nmin; s < nmax; ++s) {
        const float* Ms = M + (size_t)s * 36 * plane;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (s < sp.of(i * 6 + j)) m[i][j] += Ms[(size_t)(i * 6 + j) * plane + off];
    }
}

// ---- 1-D transforms --------------------------------------------------------------------------------
__device__ __forceinline__ void wino_bt(const float d[6], float v[6])        // B^T d
{
    v[0] = 4.0f * d[0] - 5.0f * d[2] + d[4];
    v[1] = -4.0f * d[1] - 4.0f * d[2] + d[3] + d[4];
    v[2] = 4.0f * d[1] - 4.0f * d[2] - d[3] + d[4];
    v[3] = -2.0f * d[1] - d[2] + 2.0f * d[3] + d[4];
    v[4] = 2.0f * d[1] - d[2] - 2.0f * d[3] + d[4];
    v[5] = 4.0f * d[1] - 5.0f * d[3] + d[5];
}
__device__ __forceinline__ void wino_g(const float g[3], float u[6])         // G g
{
    u[0] = 0.25f * g[0];
    u[1] = (-1.0f / 6.0f) * (g[0] + g[1] + g[2]);
    u[2] = (-1.0f / 6.0f) * (g[0] - g[1] + g[2]);
    u[3] = (1.0f / 24.0f) * g[0] + (1.0f / 12.0f) * g[1] + (1.0f / 6.0f) * g[2];
    u[4] = (1.0f / 24.0f) * g[0] - (1.0f / 12.0f) * g[1] + (1.0f / 6.0f) * g[2];
    u[5] = g[2];
}
__device__ __forceinline__ void wino_at(const float m[6], float y[4])        // A^T m
{
    y[0] = m[0] + m[1] + m[2] + m[3] + m[4];
    y[1] = m[1] - m[2] + 2.0f * m[3] - 2.0f * m[4];
    y[2] = m[1] + m[2] + 4.0f * m[3] + 4.0f * m[4];
    y[3] = m[1] - m[2] + 8.0f * m[3] - 8.0f * m[4] + m[5];
}

// weight gradient, F(3x3, 4x4): dW(3x3) = A'^T [ (G' e G'^T) (.) (B^T d B) ] A'   (e = 4x4 tile of dy, d = 6x6 window of x)
__device__ __forceinline__ void wino_g4(const float e[4], float u[6])        // G' e
{
    u[0] = 0.25f * e[0];
    u[1] = (-1.0f / 6.0f) * (e[0] + e[1] + e[2] + e[3]);
    u[2] = (-1.0f / 6.0f) * (e[0] - e[1] + e[2] - e[3]);
    u[3] = (1.0f / 24.0f) * e[0] + (1.0f / 12.0f) * e[1] + (1.0f / 6.0f) * e[2] + (1.0f / 3.0f) * e[3];
    u[4] = (1.0f / 24.0f) * e[0] - (1.0f / 12.0f) * e[1] + (1.0f / 6.0f) * e[2] - (1.0f / 3.0f) * e[3];
    u[5] = e[3];
}
__device__ __forceinline__ void wino_at3(const float m[6], float y[3])       // A'^T m
{
    y[0] = m[0] + m[1] + m[2] + m[3] + m[4];
    y[1] = m[1] - m[2] + 2.0f * m[3] - 2.0f * m[4];
    y[2] = m[1] + m[2] + 4.0f * m[3] + 4.0f * m[4] + m[5];
}

// ---------------------------------------------------------------------------------------------------
// SPLIT operands for the bf16 matrix cores (BASELINE config 5: "CDNA4 bf16 MFMA for ... convs"; also an opt-in arithmetic for
// the fp32 nets).  F(4x4,3x3) amplifies rounding ~100x (transform coefficients up to 8 and 1/24 with heavy cancellation in
// A^T M A), so operands merely ROUNDED to bf16 would leave 3-6 % error in the output.  Instead every transformed operand value
// v is stored as NPL bf16 numbers whose sum is v to 2^-16 (NPL = 2: hi + lo) or 2^-24 (NPL = 3: hi + mid + lo), and the GEMM
// multiplies the planes pairwise on v_mfma_f32_32x32x16_bf16 with fp32 accumulation:
//      NPL = 2:  hi*hi + hi*lo + lo*hi                               3 MFMAs of 32 cycles per 16 channels (fp32: 8 x 64 cycles)
//      NPL = 3:  hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi   6 MFMAs, every dropped term <= 2^-24 of the product
// Layout: the bf16 MFMA wants 8 consecutive reduction indices per lane, so an operand is stored
//      Op[xi][plane][red / 8][row][8]      (red = the GEMM's reduction index: channel, or tile for the weight gradients;
//                                           row = produced channel k / tile t / channel c; 16 bytes per (red block, row))
// which makes a lane's MFMA fragment ONE 16-byte LDS read and a 128-row tile of one reduction block 2 KB of contiguous memory
// for the LDS-DMA.  The transform kernels produce it through an LDS exchange (a thread computes one (red, row) element of all
// 36 points; 8 threads' results make one 16-byte vector).
typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int SPLIT_ROWS = 32;                       // rows (and 8 reduction indices) per workgroup of the transform kernels
constexpr int SPLIT_CHUNK = 12;                      // points staged in LDS at a time: 12 KB (NPL = 2) / 18 KB (NPL = 3) per workgroup
template <int NPL> constexpr int split_stage_elems() { return SPLIT_CHUNK * NPL * SPLIT_ROWS * 8; }

// one thread's 36 values (reduction slot pk of 8, row rl of 32) -> the workgroup's 36 * NPL * 32 vectors of 8 bf16 at
// dst[((xi * NPL + plane) * nblk + blk) * rows + row0 + rl][8].  The points cross LDS twelve at a time: the whole exchange at
// once (37 / 55 KB) left two workgroups per CU and the kernels latency bound.
template <int NPL>
__device__ __forceinline__ void store_split(unsigned short* __restrict__ stage, const float (&v)[6][6], int pk, int rl,
                                            unsigned short* __restrict__ dst, size_t nblk, size_t rows, size_t blk, size_t row0)
{
    constexpr int NVEC = SPLIT_CHUNK * NPL * SPLIT_ROWS;           // 16-byte vectors per chunk
    const uint4* src = reinterpret_cast<const uint4*>(stage);
#pragma unroll
    for (int ch = 0; ch < 36 / SPLIT_CHUNK; ++ch) {
#pragma unroll
        for (int e = 0; e < SPLIT_CHUNK; ++e) {
            const int xi = ch * SPLIT_CHUNK + e;
            float r = v[xi / 6][xi % 6];
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                const unsigned short b = f2bf(r);
                stage[((e * NPL + p) * SPLIT_ROWS + rl) * 8 + pk] = b;
                r -= bf2f(b);                          // exact: the residual of a rounding fits the format
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < (NVEC + 255) / 256; ++it) {
            const int idx = threadIdx.x + it * 256;
            if (NVEC % 256 == 0 || idx < NVEC) {
                const int xp = idx / SPLIT_ROWS, r2 = idx - xp * SPLIT_ROWS;
                *reinterpret_cast<uint4*>(dst + (((size_t)(ch * SPLIT_CHUNK * NPL + xp) * nblk + blk) * rows + row0 + r2) * 8) = src[idx];
            }
        }
        __syncthreads();                               // the stage is refilled by the next chunk / the caller's next phase
    }
}

// Thread <-> (tile t, channel c) maps of the operand-producing transform kernels, and where a thread's 36 values go.
//   TMAJOR = false (forward / input-gradient operands: the GEMM reduces over channels)
//       MODE 0: workgroup = 256 tiles of one channel, grid (Tp/256, C)      -> fp32 planes  V[xi][c][t]
//       MODE n: workgroup = 8 channels x 32 tiles,    grid (Tp/32, C/8)     -> split bf16   Vs[xi][plane][c/8][t][8]
//   TMAJOR = true  (weight-gradient operands: the GEMM reduces over tiles)
//       MODE 0: workgroup = 16 tiles x 16 channels,   grid (Tp/16, Cp/16)   -> fp32         Vt[xi][t][c]   (through LDS: 64-byte runs)
//       MODE n: workgroup = 8 tiles x 32 channels,    grid (Tp/8, Cp/32)    -> split bf16   Vs[xi][plane][t/8][c][8]
struct OpIdx { int t, c, pk, rl; unsigned bx, by; };

template <bool TMAJOR, int MODE> constexpr int op_smem_bytes()
{
    return MODE > 0 ? SPLIT_CHUNK * (MODE > 0 ? MODE : 1) * SPLIT_ROWS * 8 * 2 : (TMAJOR ? 36 * 16 * 17 * 4 : 16);
}

template <bool TMAJOR, int MODE>
__device__ __forceinline__ OpIdx op_index(bool remap)
{
    OpIdx ix;
    remap2d(remap, ix.bx, ix.by);
    const int tid = threadIdx.x;
    if (!TMAJOR && MODE == 0) { ix.pk = 0; ix.rl = 0; ix.t = ix.bx * 256 + tid; ix.c = ix.by; }
    else if (!TMAJOR) { ix.pk = tid >> 5; ix.rl = tid & 31; ix.t = ix.bx * SPLIT_ROWS + ix.rl; ix.c = ix.by * 8 + ix.pk; }
    else if (MODE == 0) { ix.pk = tid & 15; ix.rl = tid >> 4; ix.t = ix.bx * 16 + ix.pk; ix.c = ix.by * 16 + ix.rl; }
    else { ix.pk = tid & 7; ix.rl = tid >> 3; ix.t = ix.bx * 8 + ix.pk; ix.c = ix.by * SPLIT_ROWS + ix.rl; }
    return ix;
}

static dim3 op_grid(bool tmajor, int mode, int Tp, int Cn)      // Cn: channels (TMAJOR: padded to 128)
{
    if (!tmajor) return mode == 0 ? dim3(cdiv(Tp, 256), Cn) : dim3(Tp / SPLIT_ROWS, Cn / 8);
    return mode == 0 ? dim3(Tp / 16, Cn / 16) : dim3(Tp / 8, Cn / SPLIT_ROWS);
}

__device__ __forceinline__ void wino_store_tmajor(float (*stage)[16][17], const float v[6][6], int tl, int cl, float* __restrict__ dst,
                                                  int t0, int c0, int Tp, int Cp);

// Cn = channels of the operand (TMAJOR: the padded row length Cp), Tp = padded tiles
template <bool TMAJOR, int MODE>
__device__ __forceinline__ void op_emit(unsigned char* smem, const float (&v)[6][6], const OpIdx& ix, void* __restrict__ out, int Cn, int Tp)
{
    if (MODE > 0) {
        unsigned short* stage = reinterpret_cast<unsigned short*>(smem);
        if (!TMAJOR) store_split<(MODE > 0 ? MODE : 1)>(stage, v, ix.pk, ix.rl, static_cast<unsigned short*>(out), Cn / 8, Tp, ix.by, (size_t)ix.bx * SPLIT_ROWS);
        else store_split<(MODE > 0 ? MODE : 1)>(stage, v, ix.pk, ix.rl, static_cast<unsigned short*>(out), Tp / 8, Cn, ix.bx, (size_t)ix.by * SPLIT_ROWS);
    } else if (TMAJOR) {
        wino_store_tmajor(reinterpret_cast<float (*)[16][17]>(smem), v, ix.pk, ix.rl, static_cast<float*>(out), ix.bx * 16, ix.by * 16, Tp, Cn);
    } else {
        float* V = static_cast<float*>(out);
        const size_t plane = (size_t)Cn * Tp;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) V[(size_t)(i * 6 + j) * plane + (size_t)ix.c * Tp + ix.t] = v[i][j];
    }
}

// U[xi][c][k], k < Kp (zero beyond K), from W[c*sc + k*sm + r*3 + s] (flip: r -> 2-r, s -> 2-s)
// MODE 0: fp32 planes U[xi][c][k].  MODE 2 / 3: split bf16, Us[xi][plane][c/8][k][8] (grid = (Kp/32, C/8)).
template <int MODE>
__global__ void __launch_bounds__(256) wino_filter_kernel(const float* __restrict__ W, int C, int K, int Kp, long sc, long sm, int flip,
                                                          void* __restrict__ Uout)
{
    __shared__ __attribute__((aligned(16))) unsigned short stage[MODE > 0 ? split_stage_elems<MODE ? MODE : 1>() : 8];
    int k, c;
    if (MODE > 0) { k = blockIdx.x * SPLIT_ROWS + (threadIdx.x & 31); c = blockIdx.y * 8 + (threadIdx.x >> 5); }
    else { k = blockIdx.x * 256 + threadIdx.x; c = blockIdx.y; if (k >= Kp) return; }
    float g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int rr = flip ? 2 - r : r, ss = flip ? 2 - s : s;
            g[r][s] = k < K ? W[(long)c * sc + (long)k * sm + rr * 3 + ss] : 0.0f;
        }
    float t[6][3], u[6][6];
#pragma unroll
    for (int s = 0; s < 3; ++s) {                 // columns: t[:,s] = G g[:,s]
        const float col[3] = {g[0][s], g[1][s], g[2][s]};
        float o[6];
        wino_g(col, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) t[i][s] = o[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) wino_g(t[i], u[i]);          // rows: u[i,:] = G t[i,:]
    if (MODE > 0) {
        store_split<MODE ? MODE : 1>(stage, u, threadIdx.x >> 5, threadIdx.x & 31, static_cast<unsigned short*>(Uout), C / 8, Kp, blockIdx.y,
                                     (size_t)blockIdx.x * SPLIT_ROWS);
        return;
    }
    float* U = static_cast<float*>(Uout);
    const size_t plane = (size_t)C * Kp;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) U[(size_t)(i * 6 + j) * plane + (size_t)c * Kp + k] = u[i][j];
}

// MODE 0: fp32 planes V[xi][c][t].  MODE 2 / 3: split bf16, Vs[xi][plane][c/8][t][8] (grid = (Tp/32, C/8)).  TIN: float, or
// unsigned short = bf16 activations (BASELINE config 5).
template <int MODE, typename TIN>
__global__ void __launch_bounds__(256) wino_input_kernel(const TIN* __restrict__ x, int B, int C, int H, int Wd, int TY, int TX, int Tp,
                                                         void* __restrict__ Vout, int remap)
{
    __shared__ __attribute__((aligned(16))) unsigned short stage[MODE > 0 ? split_stage_elems<MODE ? MODE : 1>() : 8];
    unsigned bx, by;
    remap2d(remap != 0, bx, by);
    int t, c;
    if (MODE > 0) { t = bx * SPLIT_ROWS + (threadIdx.x & 31); c = by * 8 + (threadIdx.x >> 5); }
    else { t = bx * 256 + threadIdx.x; c = by; if (t >= Tp) return; }
    const int T = B * TY * TX;
    float d[6][6];
    if (t < T) {
        const int b = t / (TY * TX), rem = t - b * TY * TX;
        const int ty = rem / TX, tx = rem - ty * TX;
        const TIN* xp = x + ((size_t)b * C + c) * H * Wd;
        const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
        if ((Wd & 3) == 0) {
            // the window's inner four columns are the tile's own, vector aligned: one 4-element load + two halo scalars per row
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int yy = y0 + i;
                const bool yok = (unsigned)yy < (unsigned)H;
                const TIN* rp = xp + (size_t)(yok ? yy : 0) * Wd;
                const float4 mid = yok ? ld4(rp, (size_t)(x0 + 1) >> 2) : make_float4(0.f, 0.f, 0.f, 0.f);
                d[i][0] = (yok && x0 >= 0) ? ld1(rp, (size_t)x0) : 0.0f;
                d[i][1] = mid.x; d[i][2] = mid.y; d[i][3] = mid.z; d[i][4] = mid.w;
                d[i][5] = (yok && x0 + 5 < Wd) ? ld1(rp, (size_t)(x0 + 5)) : 0.0f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int yy = y0 + i;
                const bool yok = (unsigned)yy < (unsigned)H;
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const int xx = x0 + j;
                    d[i][j] = (yok && (unsigned)xx < (unsigned)Wd) ? ld1(xp, (size_t)yy * Wd + xx) : 0.0f;
                }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) d[i][j] = 0.0f;
    }
    float w[6][6], v[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {                 // columns: w[:,j] = B^T d[:,j]
        const float col[6] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]};
        float o[6];
        wino_bt(col, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) w[i][j] = o[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) wino_bt(w[i], v[i]);         // rows
    if (MODE > 0) {
        store_split<MODE ? MODE : 1>(stage, v, threadIdx.x >> 5, threadIdx.x & 31, static_cast<unsigned short*>(Vout), C / 8, Tp, by,
                                     (size_t)bx * SPLIT_ROWS);
        return;
    }
    float* V = static_cast<float*>(Vout);
    const size_t plane = (size_t)C * Tp;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) V[(size_t)(i * 6 + j) * plane + (size_t)c * Tp + t] = v[i][j];
}

// y[b][k][4ty+i][4tx+j] = (A^T M A)[i][j], M = the sum of the GEMM's `nsplit` partial results (ascending: deterministic).
// EPI: 0 plain; 1 = + bias[k], ReLU; 2 = + bias[k], ReLU, 2x2 max-pool (y is then [B,K,H/2,W/2]) — the VGG16 chain
// Conv2d(bias) -> ReLU(inplace) [-> MaxPool2d(2,2)] (models/vgg16.py:9-21) without a second pass over the activations.
// NaN-propagating like torch's relu / max_pool2d.
template <int EPI, typename TOUT>
__global__ void __launch_bounds__(256) wino_output_kernel(const float* __restrict__ Mo, WinoSplit split, const float* __restrict__ bias,
                                                          int B, int K, int Kp, int H, int Wd, int TY, int TX, int Tp, TOUT* __restrict__ y, int remap)
{
    unsigned bx, by;
    remap2d(remap != 0, bx, by);
    const int t = bx * 256 + threadIdx.x, k = by;
    const int T = B * TY * TX;
    if (t >= T) return;
    float m[6][6];
    const size_t plane = (size_t)Kp * Tp;
    wino_load_sum(Mo, split, plane, (size_t)k * Tp + t, m);
    float w[4][6], o[4][4];
#pragma unroll
    for (int j = 0; j < 6; ++j) {                 // columns: w[:,j] = A^T m[:,j]
        const float col[6] = {m[0][j], m[1][j], m[2][j], m[3][j], m[4][j], m[5][j]};
        float r4[4];
        wino_at(col, r4);
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i][j] = r4[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) wino_at(w[i], o[i]);
    if (EPI >= 1) {
        const float bv = bias ? bias[k] : 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float v = o[i][j] + bv; o[i][j] = v < 0.0f ? 0.0f : v; }
    }
    const int b = t / (TY * TX), rem = t - b * TY * TX;
    const int ty = rem / TX, tx = rem - ty * TX;
    if (EPI == 2) {                                // H, W even (host-checked): the 4x4 tile pools to 2x2
        const int Hh = H >> 1, Wh = Wd >> 1;
        TOUT* yp = y + ((size_t)b * K + k) * Hh * Wh;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float a0 = o[2 * i][2 * j], a1 = o[2 * i][2 * j + 1], a2 = o[2 * i + 1][2 * j], a3 = o[2 * i + 1][2 * j + 1];
                const float m01 = (a0 > a1 || a0 != a0) ? a0 : a1, m23 = (a2 > a3 || a2 != a2) ? a2 : a3;
                const int yy = 2 * ty + i, xx = 2 * tx + j;
                if (yy < Hh && xx < Wh) st1(yp, (size_t)yy * Wh + xx, (m01 > m23 || m01 != m01) ? m01 : m23);
            }
        return;
    }
    TOUT* yp = y + ((size_t)b * K + k) * H * Wd;
    const int y0 = 4 * ty, x0 = 4 * tx;
    if ((Wd & 3) == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (y0 + i < H) st4(yp, ((size_t)(y0 + i) * Wd + x0) >> 2, make_float4(o[i][0], o[i][1], o[i][2], o[i][3]));
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (y0 + i < H && x0 + j < Wd) st1(yp, (size_t)(y0 + i) * Wd + x0 + j, o[i][j]);
    }
}

// ---- weight-gradient transforms: outputs are TILE-major ([xi][t][channel]) because the GEMM then reduces over tiles ----
// One workgroup = 16 tiles x 16 channels: a thread transforms one (tile, channel) with its lanes along the tiles (coalesced
// reads of neighbouring windows), the 36 results cross LDS and leave with the lanes along the channels (64-byte runs).
__device__ __forceinline__ void wino_store_tmajor(float (*stage)[16][17], const float v[6][6], int tl, int cl, float* __restrict__ dst,
                                                  int t0, int c0, int Tp, int Cp)
{
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) stage[i * 6 + j][tl][cl] = v[i][j];
    __syncthreads();
    const int c = threadIdx.x & 15, t = threadIdx.x >> 4;
    const size_t plane = (size_t)Tp * Cp;
#pragma unroll
    for (int xi = 0; xi < 36; ++xi) dst[(size_t)xi * plane + (size_t)(t0 + t) * Cp + c0 + c] = stage[xi][t][c];
}

// window operand (x for Conv2d, dy for ConvTranspose2d):  Vt[xi][t][c] = (B^T d B)[xi], zero for t >= T or c >= C
template <int MODE, typename TIN>
__global__ void __launch_bounds__(256) wino_wrw_window_kernel(const TIN* __restrict__ x, int B, int C, int H, int Wd, int TY, int TX,
                                                              int Tp, int Cp, void* __restrict__ Vt)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[op_smem_bytes<true, MODE>()];
    const OpIdx ix = op_index<true, MODE>(false);
    const int t = ix.t, c = ix.c, T = B * TY * TX;
    float d[6][6];
    const bool live = t < T && c < C;
    int b = 0, ty = 0, tx = 0;
    if (live) { b = t / (TY * TX); const int rem = t - b * TY * TX; ty = rem / TX; tx = rem - ty * TX; }
    const TIN* xp = x + ((size_t)b * C + (live ? c : 0)) * H * Wd;
    const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int yy = y0 + i;
        const bool yok = live && (unsigned)yy < (unsigned)H;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int xx = x0 + j;
            d[i][j] = (yok && (unsigned)xx < (unsigned)Wd) ? ld1(xp, (size_t)yy * Wd + xx) : 0.0f;
        }
    }
    float w[6][6], v[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const float col[6] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]};
        float o[6];
        wino_bt(col, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) w[i][j] = o[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) wino_bt(w[i], v[i]);
    op_emit<true, MODE>(smem, v, ix, Vt, Cp, Tp);
}

// tile operand (dy for Conv2d, x for ConvTranspose2d):  Et[xi][t][k] = (G' e G'^T)[xi], e = the 4x4 tile, zero beyond T / K
template <int MODE, typename TIN>
__global__ void __launch_bounds__(256) wino_wrw_tile_kernel(const TIN* __restrict__ dy, int B, int K, int H, int Wd, int TY, int TX,
                                                            int Tp, int Kp, void* __restrict__ Et)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[op_smem_bytes<true, MODE>()];
    const OpIdx ix = op_index<true, MODE>(false);
    const int t = ix.t, k = ix.c, T = B * TY * TX;
    const bool live = t < T && k < K;
    int b = 0, ty = 0, tx = 0;
    if (live) { b = t / (TY * TX); const int rem = t - b * TY * TX; ty = rem / TX; tx = rem - ty * TX; }
    const TIN* dp = dy + ((size_t)b * K + (live ? k : 0)) * H * Wd;
    float e[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int yy = 4 * ty + i, xx = 4 * tx + j;
            e[i][j] = (live && yy < H && xx < Wd) ? ld1(dp, (size_t)yy * Wd + xx) : 0.0f;
        }
    float w[6][4], v[6][6];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float col[4] = {e[0][j], e[1][j], e[2][j], e[3][j]};
        float o[6];
        wino_g4(col, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) w[i][j] = o[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) wino_g4(w[i], v[i]);
    op_emit<true, MODE>(smem, v, ix, Et, Kp, Tp);
}

// dW[k][c][r][s] = (A'^T Mw[:][k][c] A')[r][s]
__global__ void __launch_bounds__(256) wino_wrw_output_kernel(const float* __restrict__ Mw, WinoSplit split, int K, int C, int Kp, int Cp,
                                                              float* __restrict__ dW)
{
    const int c = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
    if (c >= C) return;
    float m[6][6];
    const size_t plane = (size_t)Kp * Cp;
    wino_load_sum(Mw, split, plane, (size_t)k * Cp + c, m);
    float w[3][6], o[3][3];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const float col[6] = {m[0][j], m[1][j], m[2][j], m[3][j], m[4][j], m[5][j]};
        float r3[3];
        wino_at3(col, r3);
#pragma unroll
        for (int i = 0; i < 3; ++i) w[i][j] = r3[i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) wino_at3(w[i], o[i]);
    float* dst = dW + ((size_t)k * C + c) * 9;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) dst[i * 3 + j] = o[i][j];
}

// ---- the 36 GEMMs ----------------------------------------------------------------------------------
// M[xi][k][t] = sum_c U[xi][c][k] * V[xi][c][t];  C % 16 == 0, Kp % 128 == 0, Tp % 128 == 0.
// corr_argmax_fast_kernel's pipeline (see there for the measurements behind each choice): operand tiles HBM/L2 -> LDS by
// global_load_lds_dwordx4 three stages ahead in a 4-slot ring, counted vmcnt + raw s_barrier per stage, a wave owns the
// 32x32 sub-tiles {wm*32, +64} x {wn*32, +64} so that its two A (B) fragments of a k-step are ONE ds_read2st64_b32.
// nsplit > 1: the reduction is cut into nsplit ranges of `sps` stages; range `ks` writes its own partial Mo + ks*36*Kp*Tp (summed,
// in order, by the output transform).  Small layers have too few 128x128 tiles to fill 256 CUs x 2 otherwise (36 x 4 x 1 = 144
// workgroups for a 512-channel 16x16 map), and tile counts just above a multiple of 512 leave a nearly empty second round.
// BM = rows (produced channels) per workgroup tile: 128, or 64 for the layers that produce 64 channels (VGG conv1_2, the outermost
// U-Net levels) — padding those to 128 rows made half of the GEMM's flops, and half of the M it writes, zeros.  With 64 rows a wave
// owns 32 rows x {32, +64} columns: one A fragment and one ds_read2st64 of B per two MFMAs, three DMA pieces per wave and stage
// (one of A: four 256-byte rows, two of B).
template <int BM>
__global__ void __launch_bounds__(WG_THREADS, 2) wino_gemm_kernel(const float* __restrict__ U, const float* __restrict__ V, int C, int Kp, int Tp,
                                                                  int ktiles, int ttiles, WinoSplit split, float* __restrict__ Mo)
{
    static_assert(BM == 128 || BM == 64, "row tile");
    constexpr int SLOT = WG_BK * (BM + WG_BN);                 // floats per ring slot: A [16][BM] then B [16][128]
    constexpr int MI = BM / 64;                                // 32-row fragments per wave
    __shared__ __attribute__((aligned(16))) float lds[WG_NBUF * SLOT];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    // all tiles of one xi (they share U[xi] and V[xi]) get consecutive logical ids -> one XCD's L2
    // head workgroups (the long ones) are dispatched first, the tail's short ones fill the last round; each group is spread
    // over the XCDs on its own
    const int tiles = ktiles * ttiles;
    const unsigned head = (unsigned)(split.head_xi() * tiles * split.nsplit);
    const bool is_tail = blockIdx.x >= head;
    const unsigned L = is_tail ? xcd_remap(blockIdx.x - head, gridDim.x - head) : xcd_remap(blockIdx.x, head);
    const int nsplit = is_tail ? split.nsplit_t : split.nsplit, sps = is_tail ? split.sps_t : split.sps;
    const int per_xi = tiles * nsplit;
    const int xi0 = L / per_xi, rem = L - xi0 * per_xi;
    const int xi = xi0 + (is_tail ? split.xi_split : 0);
    const int ks = rem % nsplit, kt = (rem / nsplit) % ktiles, tt = rem / (nsplit * ktiles);
    const int k0 = kt * BM, t0 = tt * WG_BN;
    const int s_lo = ks * sps;
    const float* A = U + (size_t)xi * C * Kp + (size_t)s_lo * WG_BK * Kp;
    const float* Bm = V + (size_t)xi * C * Tp + (size_t)s_lo * WG_BK * Tp;

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int nstage = min(C / WG_BK - s_lo, sps);        // stages of this workgroup (>= 1)
    // DMA pieces of 1 KiB per wave and stage.  BM = 128: 2 of A + 2 of B, each two 512-byte rows (lane -> row parity, 16-byte
    // column).  BM = 64: 1 of A = four 256-byte rows (lane -> row lane>>4, column lane&15) + 2 of B.
    constexpr int NPA = BM == 128 ? 2 : 1, NP = NPA + 2;
    constexpr int AHEAD = WG_NBUF - 1;
    // One pointer per DMA piece, advanced by a constant per stage; the prefetch is unconditional (past the last stage it re-reads
    // the last one into a slot nobody reads again): no branch in the loop body and one constant vmcnt (see wino_gemm_split_kernel).
    const float* gp[NP];
    int loff[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const bool isA = p < NPA;
        if (isA && BM == 64) {
            const int row = 4 * wave + (lane >> 4);                       // 16 rows of 64 floats: wave w owns rows 4w..4w+3
            loff[p] = 4 * wave * BM;
            gp[p] = A + (size_t)row * Kp + k0 + (lane & 15) * 4;
        } else {
            const int pair = wave + 4 * (isA ? p : p - NPA);
            loff[p] = (isA ? 0 : WG_BK * BM) + pair * 2 * 128;
            const size_t row = (size_t)2 * pair + (lane >> 5);
            gp[p] = isA ? (A + row * Kp + k0 + (lane & 31) * 4) : (Bm + row * Tp + t0 + (lane & 31) * 4);
        }
    }
    const size_t strideA = (size_t)WG_BK * Kp, strideB = (size_t)WG_BK * Tp;
    auto dma_piece = [&](int p, int slot, bool more) {
        __builtin_amdgcn_global_load_lds((gptr_t)gp[p], (lptr_t)&lds[slot * SLOT + loff[p]], 16, 0, 0);
        gp[p] += more ? (p < NPA ? strideA : strideB) : 0;
    };
    int issued = 0, pf_slot = 0;
#pragma unroll
    for (int a = 0; a < AHEAD; ++a) {
#pragma unroll
        for (int p = 0; p < NP; ++p) dma_piece(p, pf_slot, issued + 1 < nstage);
        ++issued;
        pf_slot = (pf_slot + 1) & (WG_NBUF - 1);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * NP) : "memory");
    __builtin_amdgcn_s_barrier();

    for (int s = 0; s < nstage; ++s) {
        const int cur = s & (WG_NBUF - 1);
        const bool more = issued + 1 < nstage;
        const float* ta = lds + (size_t)cur * SLOT + h * BM + wm * 32 + r;
        const float* tb = lds + (size_t)cur * SLOT + WG_BK * BM + h * WG_BN + wn * 32 + r;
        float fa[MI][3], fb0[3], fb1[3];
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[i][0] = ta[64 * i];
        fb0[0] = tb[0]; fb1[0] = tb[64];
#pragma unroll
        for (int kk = 0; kk < WG_BK / 2; ++kk) {
            const int cs = kk % 3, nx = (kk + 1) % 3;
            if (kk + 1 < WG_BK / 2) {
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[i][nx] = ta[(kk + 1) * 2 * BM + 64 * i];
                fb0[nx] = tb[(kk + 1) * 2 * WG_BN]; fb1[nx] = tb[(kk + 1) * 2 * WG_BN + 64];
            }
            // the stage's DMA pieces are spread over its k-steps (4 pieces: every second step; 3: steps 0, 3, 6)
            if (NP == 4 ? (kk % 2 == 0) : (kk % 3 == 0 && kk / 3 < NP)) dma_piece(NP == 4 ? kk / 2 : kk / 3, pf_slot, more);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][cs], fb0[cs], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][cs], fb1[cs], acc[i][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        ++issued;
        pf_slot = (pf_slot + 1) & (WG_NBUF - 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * NP) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the redundant tail prefetches must not outlive the workgroup's LDS

    float* out = Mo + ((size_t)ks * split.nxi + xi) * Kp * Tp;
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
        const int t = t0 + wn * 32 + jn * 64 + r;
#pragma unroll
        for (int im = 0; im < MI; ++im)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = k0 + wm * 32 + im * 64 + (e & 3) + 8 * (e >> 2) + 4 * h;
                out[(size_t)k * Tp + t] = acc[im][jn][e];
            }
    }
}

// The same 36 GEMMs on the bf16 matrix cores with SPLIT operands (see store_split): M[xi][k][t] = sum_c U[xi][c][k] V[xi][c][t] with
// U, V given as NPL bf16 planes each.  Same workgroup tile (128 x 128), same 4-slot LDS-DMA ring of 16-channel stages, same
// reduction cuts and the same fp32 result layout as wino_gemm_kernel — only the inside of a stage differs: the stage holds
// [plane][channel block of 8][128 rows][8] per operand (2 KB per (plane, block): the image the LDS-DMA writes lane-linearly),
// a lane's fragment for `v_mfma_f32_32x32x16_bf16` (row r = lane & 31, channels 8h .. 8h+7, h = lane >> 5) is one ds_read_b128,
// and one k-step of 16 channels is 3 (NPL = 2) or 6 (NPL = 3) MFMAs of 32 cycles per 32x32 tile instead of 8 of 64.
template <int NPL>
__global__ void __launch_bounds__(WG_THREADS, 2) wino_gemm_split_kernel(const unsigned short* __restrict__ U, const unsigned short* __restrict__ V,
                                                                        int C, int Kp, int Tp, int ktiles, int ttiles, WinoSplit split,
                                                                        float* __restrict__ Mo)
{
    constexpr int OPB = NPL * 2 * WG_BM * 8;                  // bf16 elements of one operand in a stage: planes x 2 channel blocks x 128 rows x 8
    constexpr int SLOT = 2 * OPB;
    constexpr int NSLOT = NPL == 2 ? 4 : 3;                   // 64 KB / 72 KB of LDS: two workgroups per CU either way
    __shared__ __attribute__((aligned(16))) unsigned short lds[NSLOT * SLOT];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    const int tiles = ktiles * ttiles;
    const unsigned head = (unsigned)(split.head_xi() * tiles * split.nsplit);
    const bool is_tail = blockIdx.x >= head;
    const unsigned L = is_tail ? xcd_remap(blockIdx.x - head, gridDim.x - head) : xcd_remap(blockIdx.x, head);
    const int nsplit = is_tail ? split.nsplit_t : split.nsplit, sps = is_tail ? split.sps_t : split.sps;
    const int per_xi = tiles * nsplit;
    const int xi0 = L / per_xi, rem = L - xi0 * per_xi;
    const int xi = xi0 + (is_tail ? split.xi_split : 0);
    const int ks = rem % nsplit, kt = (rem / nsplit) % ktiles, tt = rem / (nsplit * ktiles);
    const int k0 = kt * WG_BM, t0 = tt * WG_BN;
    const int s_lo = ks * sps;
    const int nblk = C / 8;
    // plane p, channel block cb of this workgroup's rows: base + ((p * nblk + cb) * rows + row) * 8
    const unsigned short* A = U + ((size_t)xi * NPL * nblk * Kp + k0) * 8;
    const unsigned short* Bm = V + ((size_t)xi * NPL * nblk * Tp + t0) * 8;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int nstage = min(C / WG_BK - s_lo, sps);
    constexpr int NPIECE = 8 * NPL;                           // 1-KiB DMA pieces per stage: 2 operands x NPL planes x 2 blocks x 2 halves of 64 rows
    constexpr int NP = NPIECE / 4;                            // per wave
    constexpr int AHEAD = NSLOT - 1;                          // stages in flight
    // A stage costs only 3 / 6 MFMAs of 32 cycles per tile here (the fp32 kernel: 8 of 64), so everything else in the loop has to be
    // cheap: every wave keeps one pointer per DMA piece and advances it by a constant per stage, the prefetch is UNCONDITIONAL
    // (past the last stage it re-reads the last one into a slot nobody reads again) so that the loop body has no branch and the
    // number of DMAs in flight is the same in every iteration: one constant vmcnt.
    const unsigned short* gp[NP];
    int loff[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int q = wave + 4 * i;                           // uniform
        const int opnd = q / (4 * NPL), within = q - opnd * 4 * NPL;
        const int pl = within >> 2, cb = (within >> 1) & 1, half = within & 1;
        loff[i] = opnd * OPB + ((pl * 2 + cb) * WG_BM + half * 64) * 8;
        const size_t blk = (size_t)s_lo * 2 + cb;
        gp[i] = opnd == 0 ? A + (((size_t)pl * nblk + blk) * Kp + half * 64 + lane) * 8 : Bm + (((size_t)pl * nblk + blk) * Tp + half * 64 + lane) * 8;
    }
    const size_t strideA = (size_t)2 * Kp * 8, strideB = (size_t)2 * Tp * 8;          // one stage = two channel blocks
    auto dma_stage_piece = [&](int i, int slot) {
        __builtin_amdgcn_global_load_lds((gptr_t)gp[i], (lptr_t)&lds[slot * SLOT + loff[i]], 16, 0, 0);
    };
    auto advance = [&](int i, bool more) {                    // -> the next stage's piece, or stay on the last stage
        const int q = wave + 4 * i;
        const size_t st_ = q < 4 * NPL ? strideA : strideB;
        gp[i] += more ? st_ : 0;
    };
    int issued = 0;                                           // stages whose DMAs have been issued (clamped to nstage - 1 as a source)
    int pf_slot = 0;
#pragma unroll
    for (int a = 0; a < AHEAD; ++a) {
#pragma unroll
        for (int i = 0; i < NP; ++i) { dma_stage_piece(i, pf_slot); advance(i, issued + 1 < nstage); }
        ++issued;
        pf_slot = pf_slot + 1 == NSLOT ? 0 : pf_slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * NP) : "memory");
    __builtin_amdgcn_s_barrier();

    int cur = 0;
    for (int s = 0; s < nstage; ++s) {
        const unsigned short* sa = lds + cur * SLOT + (h * WG_BM + wm * 32 + r) * 8;
        const unsigned short* sb = lds + cur * SLOT + OPB + (h * WG_BM + wn * 32 + r) * 8;
        // fragments in the order the tiles need them — (0,0) first — so that the first MFMAs start while the later reads are in flight
        // (the compiler places the counted lgkmcnt waits)
        bf16x8 fa[NPL][2], fb[NPL][2];
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
            fa[p][0] = *reinterpret_cast<const bf16x8*>(sa + (p * 2 * WG_BM) * 8);
            fb[p][0] = *reinterpret_cast<const bf16x8*>(sb + (p * 2 * WG_BM) * 8);
        }
#pragma unroll
        for (int p = 0; p < NPL; ++p) fb[p][1] = *reinterpret_cast<const bf16x8*>(sb + (p * 2 * WG_BM + 64) * 8);
#pragma unroll
        for (int p = 0; p < NPL; ++p) fa[p][1] = *reinterpret_cast<const bf16x8*>(sa + (p * 2 * WG_BM + 64) * 8);
        const bool more = issued + 1 < nstage;
        // products in ascending magnitude: the small cross terms first, hi*hi last
        constexpr int NPROD = NPL == 2 ? 3 : 6;
        constexpr int PA[6] = {NPL == 2 ? 1 : 2, 0, NPL == 2 ? 0 : 1, 1, 0, 0};      // NPL=2: (1,0)(0,1)(0,0)   NPL=3: (2,0)(0,2)(1,1)(1,0)(0,1)(0,0)
        constexpr int PB[6] = {0, NPL == 2 ? 1 : 2, NPL == 2 ? 0 : 1, 0, 1, 0};
        int piece = 0;
#pragma unroll
        for (int tile = 0; tile < 4; ++tile) {
            const int i = tile >> 1, j = tile & 1;
#pragma unroll
            for (int q = 0; q < NPROD; ++q) {
                if (q % 3 == 0 && piece < NP) {               // this wave's DMA pieces of stage s + AHEAD, spread between the MFMAs
                    dma_stage_piece(piece, pf_slot);
                    advance(piece, more);
                    ++piece;
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[q]][i], fb[PB[q]][j], acc[i][j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 4 * ((NPROD + 2) / 3); i < NP; ++i) { dma_stage_piece(i, pf_slot); advance(i, more); }
        ++issued;
        pf_slot = pf_slot + 1 == NSLOT ? 0 : pf_slot + 1;
        cur = cur + 1 == NSLOT ? 0 : cur + 1;
        // stage s + 1 has landed once all but the AHEAD - 1 youngest stages' DMAs are done
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * NP) : "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the redundant tail prefetches must not outlive the workgroup's LDS

    float* out = Mo + ((size_t)ks * split.nxi + xi) * Kp * Tp;
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
        const int t = t0 + wn * 32 + jn * 64 + r;
#pragma unroll
        for (int im = 0; im < 2; ++im)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = k0 + wm * 32 + im * 64 + (e & 3) + 8 * (e >> 2) + 4 * h;
                out[(size_t)k * Tp + t] = acc[im][jn][e];
            }
    }
}

// launches the 36 GEMMs in the arithmetic `math`: 0 = fp32 operands on v_mfma_f32_32x32x2_f32; 2 / 3 = split bf16 operands
// `useful`: the multiply-adds of the UNPADDED problem x 2 (rows / cols / reduction before rounding up to the tile sizes)
// Rows (produced channels) of the GEMM operand U / of M as stored: a multiple of 128, or exactly 64 for the 64-channel layers in
// fp32 arithmetic (wino_gemm_kernel<64>); math < 0 = "whatever arithmetic" for workspace sizing (the larger padding).
static int wino_rows_padded(int K, int math) { return (math == 0 && K <= 64) ? 64 : (K + WG_BM - 1) / WG_BM * WG_BM; }
static int wino_row_tiles(int Kp) { return Kp == 64 ? 1 : Kp / WG_BM; }

static void launch_wino_gemm(int math, const void* A, const void* Bv, int red, int rows, int cols, const WinoSplit& sp, float* Mo, hipStream_t st,
                             double useful = 0.0)
{
    const int kt = wino_row_tiles(rows), tt = cols / WG_BN;
    const unsigned grid = sp.workgroups(kt * tt);
    profile_mark_start(st, 3);
    if (math == 2)
        wino_gemm_split_kernel<2><<<grid, WG_THREADS, 0, st>>>(static_cast<const unsigned short*>(A), static_cast<const unsigned short*>(Bv), red, rows, cols, kt, tt, sp, Mo);
    else if (math == 3)
        wino_gemm_split_kernel<3><<<grid, WG_THREADS, 0, st>>>(static_cast<const unsigned short*>(A), static_cast<const unsigned short*>(Bv), red, rows, cols, kt, tt, sp, Mo);
    else if (rows == 64)
        wino_gemm_kernel<64><<<grid, WG_THREADS, 0, st>>>(static_cast<const float*>(A), static_cast<const float*>(Bv), red, rows, cols, kt, tt, sp, Mo);
    else
        wino_gemm_kernel<128><<<grid, WG_THREADS, 0, st>>>(static_cast<const float*>(A), static_cast<const float*>(Bv), red, rows, cols, kt, tt, sp, Mo);
    profile_mark_stop(st, 3, 72.0 * red * rows * cols, useful);
}

// ---------------------------------------------------------------------------------------------------
// how to cut the reduction of the 36 GEMMs (measurements: tools/sweep_wino_split.py -> profiles/r02_sweep_wino_split.txt).
//  * uniform cut (weight gradients: thousands of stages, a handful of output tiles): estimated time = rounds of 512 resident
//    workgroups x (stages + pipeline fill) plus the extra pass over the partial results; the lowest estimate wins.
//  * head / tail: a grid of 36 x 8 or 36 x 16 tiles is 288 / 576 workgroups — one round of the chip plus an eighth, i.e. a
//    second round that is nearly empty (56 % of the matrix cores busy on a 512-channel 32x32 map).  There the GEMMs of the
//    first 32 points run uncut and fill the round exactly, and the last 4 are cut 2-4 ways so that their short workgroups
//    fill the tail evenly: whole convolution -14 % (512@32x32), -23 % (1024 -> 256 @32x32), -10 % (512 -> 128 @64x64).
//  * small grids with short reductions (<= 144 workgroups, <= 32 stages) stay uncut: the cut shortens the GEMM by 3-8 us and
//    costs the output transform 10 us per extra slab on a map with that few tiles.
static int g_force_split[3] = {0, 0, 0};                     // ipsr_debug_force_wino_split: {nsplit, xi_split, nsplit_t}, 0 = automatic

static WinoSplit wino_choose_split(int tiles36, int stages, size_t m_bytes)
{
    const int tiles = tiles36 / 36;
    WinoSplit best{1, stages, 36, 1, stages};
    if (g_force_split[0] > 0) {                              // tuning aid (tools/sweep_wino_split.py), off unless asked for
        const int a = g_force_split[0], x = g_force_split[1], b = g_force_split[2];
        if (a <= stages && b <= stages) {
            const int pa = cdiv(stages, a), pb = cdiv(stages, b);
            return WinoSplit{cdiv(stages, pa), pa, x, cdiv(stages, pb), pb};
        }
    }
    if (tiles36 <= 144 && stages <= 32) return best;
    if ((tiles == 8 || tiles == 16) && stages >= 16 && stages <= 128) {
        const int nt = stages >= 32 ? 4 : 2, per = cdiv(stages, nt);
        return WinoSplit{1, stages, 32, cdiv(stages, per), per};
    }
    double best_t = -1.0;
    const int cand[] = {1, 2, 3, 4, 6, 8, 12, 16};
    for (int ns : cand) {
        if (ns > 1 && stages / ns < 4) break;
        const int per = cdiv(stages, ns), real = cdiv(stages, per);
        const long wgs = (long)tiles36 * real;
        const long rounds = (wgs + 511) / 512;
        // a stage of two co-resident workgroups takes ~1.7 us; a last round that fills less than half the chip runs one
        // workgroup per CU, ~1.5x faster each
        const long tail = wgs - (rounds - 1) * 512;
        double t = (rounds - 1) * (per + 3) * 1.7 + (per + 3) * (tail <= 256 ? 1.15 : 1.7);
        t += (real - 1) * (double)m_bytes * 2.0 / 4.0e6;              // partials written + read at ~4 TB/s, in us
        if (best_t < 0.0 || t < best_t) { best_t = t; best = WinoSplit{real, per, 36, real, per}; }
    }
    return best;
}

// ---- arithmetic / element types of one convolution call -------------------------------------------------------------------------
// math: 0 = fp32 operands on v_mfma_f32_32x32x2_f32 (the reference's arithmetic); 2 / 3 = SPLIT bf16 operands (store_split) on
// v_mfma_f32_32x32x16_bf16.  in_bf16 / out_bf16: the activation tensors read / written are bf16 (BASELINE config 5) instead of fp32;
// weights, weight gradients and all transform arithmetic stay fp32.
struct ConvArith { int math; bool in_bf16, out_bf16; };
static inline bool arith_ok(const ConvArith& a) { return a.math == 0 || a.math == 2 || a.math == 3; }

template <typename F> static inline void with_mode(int math, F&& f)
{
    if (math == 2) f(std::integral_constant<int, 2>{});
    else if (math == 3) f(std::integral_constant<int, 3>{});
    else f(std::integral_constant<int, 0>{});
}
template <typename F> static inline void with_type(bool bf16, F&& f)
{
    if (bf16) f(static_cast<bf16_t*>(nullptr));
    else f(static_cast<float*>(nullptr));
}
#define ELEM_T(tag) std::remove_pointer_t<decltype(tag)>

struct WinoPlan { int TY, TX, T, Tp, Kp; WinoSplit sp; size_t u_floats, v_floats, m_floats, total_bytes; };

static int wino_plan(int B, int C, int K, int H, int W, WinoPlan* p, int math)
{
    if (C % WG_BK != 0) return fail(IPSR_ERR_UNSUPPORTED, "winograd: %d reduction channels are not a multiple of %d", C, WG_BK);
    p->TY = (H + 3) / 4; p->TX = (W + 3) / 4;
    p->T = B * p->TY * p->TX;
    p->Tp = (p->T + WG_BN - 1) / WG_BN * WG_BN;
    p->Kp = wino_rows_padded(K, math);
    p->u_floats = (size_t)36 * C * p->Kp * 3 / 2;          // room for three bf16 planes (the split arithmetic with NPL = 3)
    p->v_floats = (size_t)36 * C * p->Tp * 3 / 2;
    const size_t m1 = (size_t)36 * p->Kp * p->Tp;
    p->sp = wino_choose_split(36 * wino_row_tiles(p->Kp) * (p->Tp / WG_BN), C / WG_BK, m1 * 4);
    p->m_floats = m1 * p->sp.slabs();
    p->total_bytes = align_up(p->u_floats * 4, 256) + align_up(p->v_floats * 4, 256) + align_up(p->m_floats * 4, 256) + 256;
    return IPSR_OK;
}

size_t winograd_ws_bytes(int B, int C, int K, int H, int W)
{
    WinoPlan p, q;                                    // whatever arithmetic the call will ask for
    if (wino_plan(B, C, K, H, W, &p, -1) != IPSR_OK || wino_plan(B, C, K, H, W, &q, 0) != IPSR_OK) return 0;
    return p.total_bytes > q.total_bytes ? p.total_bytes : q.total_bytes;
}

size_t winograd_filter_floats(int C, int K) { return (size_t)36 * C * ((K + WG_BM - 1) / WG_BM * WG_BM) * 3 / 2; }

// y[B,K,H,W] = conv3x3(x[B,C,H,W]) with weight element (c, k, r, s) at w[c*sc + k*sm + r*3 + s], taps flipped when `flip`.
// u_cache (optional, winograd_filter_floats(C, K) floats owned by the caller): the transformed filter; computed into it when
// !u_valid, reused as is otherwise (frozen weights: VGG16).  epilogue: 0 none, 1 bias + ReLU, 2 bias + ReLU + 2x2 max-pool.
int launch_winograd(const void* x, const float* w, void* y, int B, int C, int K, int H, int W, long sc, long sm, int flip,
                    void* ws, size_t ws_bytes, hipStream_t st, const float* bias = nullptr, int epilogue = 0,
                    float* u_cache = nullptr, int u_valid = 0, ConvArith ar = ConvArith{0, false, false})
{
    WinoPlan p;
    if (int rc = wino_plan(B, C, K, H, W, &p, ar.math)) return rc;
    if (ws_bytes < p.total_bytes) return fail(IPSR_ERR_WORKSPACE, "winograd: workspace %zu < %zu", ws_bytes, p.total_bytes);
    if (epilogue < 0 || epilogue > 2 || (epilogue == 2 && ((H | W) & 1)))
        return fail(IPSR_ERR_INVALID, "winograd: epilogue %d on a %dx%d map", epilogue, H, W);
    if (!arith_ok(ar)) return fail(IPSR_ERR_INVALID, "winograd: arithmetic %d (0 = fp32 MFMA, 2 / 3 = split bf16)", ar.math);
    Carver cv(ws, ws_bytes);
    float* U = cv.take<float>(p.u_floats);
    float* V = cv.take<float>(p.v_floats);
    float* Mo = cv.take<float>(p.m_floats);
    if (u_cache) U = u_cache;
    const int remap = !debug_option(1);          // XCD-contiguous (channel, tile block) ranges; debug option 1 = off
    with_mode(ar.math, [&](auto M) {
        constexpr int MODE = decltype(M)::value;
        if (!(u_cache && u_valid)) wino_filter_kernel<MODE><<<op_grid(false, MODE, p.Kp, C), 256, 0, st>>>(w, C, K, p.Kp, sc, sm, flip, U);
        with_type(ar.in_bf16, [&](auto* tag) {
            using T = ELEM_T(tag);
            wino_input_kernel<MODE, T><<<op_grid(false, MODE, p.Tp, C), 256, 0, st>>>(static_cast<const T*>(x), B, C, H, W, p.TY, p.TX, p.Tp, V, remap);
        });
    });
    if (int rc = check_launch("wino_input_kernel")) return rc;
    launch_wino_gemm(ar.math, U, V, C, p.Kp, p.Tp, p.sp, Mo, st, 72.0 * C * K * p.T);
    if (int rc = check_launch("wino_gemm_kernel")) return rc;
    const dim3 og(cdiv(p.T, 256), K);
    with_type(ar.out_bf16, [&](auto* tag) {
        using T = ELEM_T(tag);
        T* yo = static_cast<T*>(y);
        if (epilogue == 2) wino_output_kernel<2, T><<<og, 256, 0, st>>>(Mo, p.sp, bias, B, K, p.Kp, H, W, p.TY, p.TX, p.Tp, yo, remap);
        else if (epilogue == 1) wino_output_kernel<1, T><<<og, 256, 0, st>>>(Mo, p.sp, bias, B, K, p.Kp, H, W, p.TY, p.TX, p.Tp, yo, remap);
        else wino_output_kernel<0, T><<<og, 256, 0, st>>>(Mo, p.sp, bias, B, K, p.Kp, H, W, p.TY, p.TX, p.Tp, yo, remap);
    });
    return check_launch("wino_output_kernel");
}

struct WinoWrwPlan { int TY, TX, T, Tp, Kp, Cp; WinoSplit sp; size_t e_floats, v_floats, m_floats, total_bytes; };

static int wino_wrw_plan(int B, int K, int C, int H, int W, WinoWrwPlan* p, int math)
{
    p->TY = (H + 3) / 4; p->TX = (W + 3) / 4;
    p->T = B * p->TY * p->TX;
    p->Tp = (p->T + WG_BN - 1) / WG_BN * WG_BN;            // reduction of the GEMM: a multiple of 16 (and of the 16-tile blocks)
    p->Kp = wino_rows_padded(K, math);
    p->Cp = (C + WG_BN - 1) / WG_BN * WG_BN;
    p->e_floats = (size_t)36 * p->Tp * p->Kp * 3 / 2;            // room for three bf16 planes
    p->v_floats = (size_t)36 * p->Tp * p->Cp * 3 / 2;
    const size_t m1 = (size_t)36 * p->Kp * p->Cp;
    p->sp = wino_choose_split(36 * wino_row_tiles(p->Kp) * (p->Cp / WG_BN), p->Tp / WG_BK, m1 * 4);
    p->m_floats = m1 * p->sp.slabs();
    p->total_bytes = align_up(p->e_floats * 4, 256) + align_up(p->v_floats * 4, 256) + align_up(p->m_floats * 4, 256) + 256;
    return IPSR_OK;
}

size_t winograd_wrw_ws_bytes(int B, int K, int C, int H, int W)
{
    WinoWrwPlan p, q;
    wino_wrw_plan(B, K, C, H, W, &p, -1);
    wino_wrw_plan(B, K, C, H, W, &q, 0);
    return p.total_bytes > q.total_bytes ? p.total_bytes : q.total_bytes;
}

// dW[K][C][3][3] = sum over tiles:  tile operand `et` [B,K,H,W] (4x4 tiles), window operand `dt` [B,C,H,W] (6x6 windows)
int launch_winograd_wrw(const void* et, const void* dt, float* dW, int B, int K, int C, int H, int W, void* ws, size_t ws_bytes, hipStream_t st,
                        ConvArith ar = ConvArith{0, false, false})
{
    WinoWrwPlan p;
    wino_wrw_plan(B, K, C, H, W, &p, ar.math);
    if (ws_bytes < p.total_bytes) return fail(IPSR_ERR_WORKSPACE, "winograd wrw: workspace %zu < %zu", ws_bytes, p.total_bytes);
    if (!arith_ok(ar)) return fail(IPSR_ERR_INVALID, "winograd wrw: arithmetic %d", ar.math);
    Carver cv(ws, ws_bytes);
    float* Et = cv.take<float>(p.e_floats);
    float* Vt = cv.take<float>(p.v_floats);
    float* Mw = cv.take<float>(p.m_floats);
    with_mode(ar.math, [&](auto M) {
        constexpr int MODE = decltype(M)::value;
        with_type(ar.in_bf16, [&](auto* tag) {
            using T = ELEM_T(tag);
            wino_wrw_tile_kernel<MODE, T><<<op_grid(true, MODE, p.Tp, p.Kp), 256, 0, st>>>(static_cast<const T*>(et), B, K, H, W, p.TY, p.TX, p.Tp, p.Kp, Et);
            wino_wrw_window_kernel<MODE, T><<<op_grid(true, MODE, p.Tp, p.Cp), 256, 0, st>>>(static_cast<const T*>(dt), B, C, H, W, p.TY, p.TX, p.Tp, p.Cp, Vt);
        });
    });
    if (int rc = check_launch("wino_wrw_window_kernel")) return rc;
    // M[xi][k][c] = sum_t Et[xi][t][k] * Vt[xi][t][c]: the same GEMM with the tiles as the reduction
    launch_wino_gemm(ar.math, Et, Vt, p.Tp, p.Kp, p.Cp, p.sp, Mw, st, 72.0 * p.T * K * C);
    if (int rc = check_launch("wino_gemm_kernel")) return rc;
    wino_wrw_output_kernel<<<dim3(cdiv(C, 256), K), 256, 0, st>>>(Mw, p.sp, K, C, p.Kp, p.Cp, dW);
    return check_launch("wino_wrw_output_kernel");
}

// ===================================================================================================
// The DILATED down convolution of every netG level — Conv2d(k4, stride 2, pad 3, dilation 2), models/networks.py:226 — by
// Winograd F(3x3, 4x4).  With dilation 2 and stride 2 the layer only ever reads the odd rows / columns of its input:
//      y[o] = sum_r w[r] x[2o - 3 + 2r] = sum_r w[r] X[o + r],      X[i] = x[2i - 3]   (zero outside)
// i.e. a plain 4-tap stride-1 correlation on the sub-sampled image X.  F(3,4) produces 3 outputs from a 6-wide window with
// 6 multiplies instead of 12: 4x fewer matrix-core flops in 2-D, with the SAME interpolation points as F(4,3) above, hence
// the same input transform B^T and the same 36 GEMMs; only the filter transform (G' 6x4) and the output transform (A'^T 3x6)
// change — they are the ones the 3x3 weight gradient already uses.
//   forward        window of X at origin 3t (x read with stride 2, offset -3), filter G' w G'^T, output A'^T M A' -> y
//   backward-data  dx[2q+1] = sum_r' w[3-r'] dy[q - 1 + r'] (even rows/columns get no gradient: zero): the same pipeline on
//                  windows of dy (stride 1, offset -1) with flipped taps and (Cout, Cin) swapped, written with stride 2
//   weight grad    dW[r] = sum_o dy[o] X[o + r]: F(4x4, 3x3) with the roles swapped — 3x3 tiles of dy through G, 6x6 windows
//                  of X through B^T, reduced over all tiles by the GEMM, A^T (4x6) back to the 4x4 taps.

// generic window transform: V[xi][c][t] (TMAJOR = false) or V[xi][t][c] (true) of windows at origin OS*t read as
// x[c][IS*(origin + i) + off]
template <int OS, int IS, bool TMAJOR, int MODE, typename TIN>
__global__ void __launch_bounds__(256) wino_window_kernel(const TIN* __restrict__ x, int B, int C, int H, int Wd, int off, int TY, int TX,
                                                          int Tp, int Cp, void* __restrict__ V, int remap)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[op_smem_bytes<TMAJOR, MODE>()];
    const OpIdx ix = op_index<TMAJOR, MODE>(!TMAJOR && remap != 0);
    const int t = ix.t, c = ix.c;
    if (!TMAJOR && MODE == 0 && t >= Tp) return;
    const int T = B * TY * TX;
    const bool live = t < T && c < C;
    int b = 0, ty = 0, tx = 0;
    if (live) { b = t / (TY * TX); const int rem = t - b * TY * TX; ty = rem / TX; tx = rem - ty * TX; }
    const TIN* xp = x + ((size_t)b * C + (live ? c : 0)) * H * Wd;
    float d[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int yy = IS * (OS * ty + i) + off;
        const bool yok = live && (unsigned)yy < (unsigned)H;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int xx = IS * (OS * tx + j) + off;
            d[i][j] = (yok && (unsigned)xx < (unsigned)Wd) ? ld1(xp, (size_t)yy * Wd + xx) : 0.0f;
        }
    }
    float w[6][6], v[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const float col[6] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]};
        float o[6];
        wino_bt(col, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) w[i][j] = o[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) wino_bt(w[i], v[i]);
    op_emit<TMAJOR, MODE>(smem, v, ix, V, TMAJOR ? Cp : C, Tp);
}

// U[xi][c][k] = (G' g G'^T)[xi] for 4x4 taps: W[c*sc + k*sm + r*4 + s] (flip: r -> 3-r, s -> 3-s)
template <int MODE>
__global__ void __launch_bounds__(256) wino4_filter_kernel(const float* __restrict__ W, int C, int K, int Kp, long sc, long sm, int flip,
                                                           void* __restrict__ U)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[op_smem_bytes<false, MODE>()];
    const OpIdx ix = op_index<false, MODE>(false);
    const int k = ix.t, c = ix.c;
    if (MODE == 0 && k >= Kp) return;
    float g[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            const int rr = flip ? 3 - r : r, ss = flip ? 3 - s2 : s2;
            g[r][s2] = k < K ? W[(long)c * sc + (long)k * sm + rr * 4 + ss] : 0.0f;
        }
    float t[6][4], u[6][6];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
        const float col[4] = {g[0][s2], g[1][s2], g[2][s2], g[3][s2]};
        float o[6];
        wino_g4(col, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) t[i][s2] = o[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) wino_g4(t[i], u[i]);
    op_emit<false, MODE>(smem, u, ix, U, C, Kp);
}

// y[b][k][os*(3ty+i)+oo][os*(3tx+j)+oo] = (A'^T M A')[i][j] for 3ty+i < Ho, 3tx+j < Wo  (y is [B,K,Hy,Wy])
template <typename TOUT>
__global__ void __launch_bounds__(256) wino3_output_kernel(const float* __restrict__ Mo, WinoSplit split, int B, int K, int Kp, int Ho, int Wo,
                                                           int TY, int TX, int Tp, int Hy, int Wy, int os, int oo, TOUT* __restrict__ y)
{
    const int t = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
    const int T = B * TY * TX;
    if (t >= T) return;
    float m[6][6];
    const size_t plane = (size_t)Kp * Tp;
    wino_load_sum(Mo, split, plane, (size_t)k * Tp + t, m);
    float w[3][6], o[3][3];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const float col[6] = {m[0][j], m[1][j], m[2][j], m[3][j], m[4][j], m[5][j]};
        float r3[3];
        wino_at3(col, r3);
#pragma unroll
        for (int i = 0; i < 3; ++i) w[i][j] = r3[i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) wino_at3(w[i], o[i]);
    const int b = t / (TY * TX), rem = t - b * TY * TX;
    const int ty = rem / TX, tx = rem - ty * TX;
    TOUT* yp = y + ((size_t)b * K + k) * Hy * Wy;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int oy = 3 * ty + i, ox = 3 * tx + j;
            if (oy < Ho && ox < Wo) st1(yp, (size_t)(os * oy + oo) * Wy + os * ox + oo, o[i][j]);
        }
}

// weight gradient: 3x3 tiles of dy -> Et[xi][t][k] = (G e G^T)[xi]
template <int MODE, typename TIN>
__global__ void __launch_bounds__(256) wino_wrw_tile3_kernel(const TIN* __restrict__ dy, int B, int K, int Ho, int Wo, int TY, int TX,
                                                             int Tp, int Kp, void* __restrict__ Et)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[op_smem_bytes<true, MODE>()];
    const OpIdx ix = op_index<true, MODE>(false);
    const int t = ix.t, k = ix.c, T = B * TY * TX;
    const bool live = t < T && k < K;
    int b = 0, ty = 0, tx = 0;
    if (live) { b = t / (TY * TX); const int rem = t - b * TY * TX; ty = rem / TX; tx = rem - ty * TX; }
    const TIN* dp = dy + ((size_t)b * K + (live ? k : 0)) * Ho * Wo;
    float e[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int yy = 3 * ty + i, xx = 3 * tx + j;
            e[i][j] = (live && yy < Ho && xx < Wo) ? ld1(dp, (size_t)yy * Wo + xx) : 0.0f;
        }
    float w[6][3], v[6][6];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float col[3] = {e[0][j], e[1][j], e[2][j]};
        float o[6];
        wino_g(col, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) w[i][j] = o[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) wino_g(w[i], v[i]);
    op_emit<true, MODE>(smem, v, ix, Et, Kp, Tp);
}

// dW[k][c][r][s] (4x4) = (A^T Mw[:][k][c] A)[r][s]
__global__ void __launch_bounds__(256) wino_wrw_output4_kernel(const float* __restrict__ Mw, WinoSplit split, int K, int C, int Kp, int Cp,
                                                               float* __restrict__ dW)
{
    const int c = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
    if (c >= C) return;
    float m[6][6];
    const size_t plane = (size_t)Kp * Cp;
    wino_load_sum(Mw, split, plane, (size_t)k * Cp + c, m);
    float w[4][6], o[4][4];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const float col[6] = {m[0][j], m[1][j], m[2][j], m[3][j], m[4][j], m[5][j]};
        float r4[4];
        wino_at(col, r4);
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i][j] = r4[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) wino_at(w[i], o[i]);
    float4* dst = reinterpret_cast<float4*>(dW + ((size_t)k * C + c) * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i] = make_float4(o[i][0], o[i][1], o[i][2], o[i][3]);
}

struct DilPlan { int Ho, Wo, Gy, Gx, TY, TX, T, Tp, Kp, Cp; WinoSplit sp; size_t a_floats, b_floats, m_floats, total_bytes; };

// mode 0: forward (grid = output Ho x Wo, reduce C=Cin, produce K=Cout); 1: backward-data (geom 0: grid = H/2 x W/2 odd positions
// of dx; geom 1: grid = H x W; reduce Cout, produce Cin); 2: weight gradient (tiles of dy, reduction over tiles).
// geom 0: Conv2d(k4 s2 p3 d2), output H/2 x W/2.   geom 1: Conv2d(k4 s1 p1) — netD's fourth convolution, models/networks.py:483-489
// — output (H-1) x (W-1): the same 4-tap stride-1 correlation, on the image itself (X[i] = x[i - 1]).
static int dil_plan(int geom, int mode, int B, int Cin, int H, int W, int Cout, DilPlan* p, int math)
{
    if (geom != 0 && geom != 1) return fail(IPSR_ERR_INVALID, "4x4 winograd: geometry %d", geom);
    if (geom == 0 && ((H | W) & 1)) return fail(IPSR_ERR_UNSUPPORTED, "dilated winograd: odd extent %dx%d", H, W);
    if (geom == 1 && (H < 4 || W < 4)) return fail(IPSR_ERR_UNSUPPORTED, "4x4 winograd: extent %dx%d below the kernel", H, W);
    p->Ho = geom == 0 ? H / 2 : H - 1; p->Wo = geom == 0 ? W / 2 : W - 1;
    p->Gy = (geom == 1 && mode == 1) ? H : p->Ho; p->Gx = (geom == 1 && mode == 1) ? W : p->Wo;
    p->TY = (p->Gy + 2) / 3; p->TX = (p->Gx + 2) / 3;
    p->T = B * p->TY * p->TX;
    p->Tp = (p->T + WG_BN - 1) / WG_BN * WG_BN;
    if (mode == 2) {
        p->Kp = wino_rows_padded(Cout, math);
        p->Cp = (Cin + WG_BN - 1) / WG_BN * WG_BN;
        p->a_floats = (size_t)36 * p->Tp * p->Kp * 3 / 2;        // room for three bf16 planes (split arithmetic)
        p->b_floats = (size_t)36 * p->Tp * p->Cp * 3 / 2;
        const size_t m1 = (size_t)36 * p->Kp * p->Cp;
        p->sp = wino_choose_split(36 * wino_row_tiles(p->Kp) * (p->Cp / WG_BN), p->Tp / WG_BK, m1 * 4);
        p->m_floats = m1 * p->sp.slabs();
    } else {
        const int red = mode == 0 ? Cin : Cout, prod = mode == 0 ? Cout : Cin;
        if (red % WG_BK != 0) return fail(IPSR_ERR_UNSUPPORTED, "4x4 winograd: %d reduction channels are not a multiple of %d", red, WG_BK);
        p->Kp = wino_rows_padded(prod, math);
        p->Cp = red;
        p->a_floats = (size_t)36 * red * p->Kp * 3 / 2;
        p->b_floats = (size_t)36 * red * p->Tp * 3 / 2;
        const size_t m1 = (size_t)36 * p->Kp * p->Tp;
        p->sp = wino_choose_split(36 * wino_row_tiles(p->Kp) * (p->Tp / WG_BN), red / WG_BK, m1 * 4);
        p->m_floats = m1 * p->sp.slabs();
    }
    p->total_bytes = align_up(p->a_floats * 4, 256) + align_up(p->b_floats * 4, 256) + align_up(p->m_floats * 4, 256) + 256;
    return IPSR_OK;
}

size_t winograd_dil_ws_bytes(int geom, int mode, int B, int Cin, int H, int W, int Cout)
{
    DilPlan p, q;
    if (dil_plan(geom, mode, B, Cin, H, W, Cout, &p, -1) != IPSR_OK || dil_plan(geom, mode, B, Cin, H, W, Cout, &q, 0) != IPSR_OK) return 0;
    return p.total_bytes > q.total_bytes ? p.total_bytes : q.total_bytes;
}

template <bool TMAJOR, int MODE, typename T>
static void launch_window3(int is, hipStream_t st, const T* x, int B, int C, int H, int W, int off, const DilPlan& p, int Cp, void* V)
{
    const dim3 grid = op_grid(TMAJOR, MODE, p.Tp, TMAJOR ? Cp : C);
    const int remap = !debug_option(1);          // XCD-contiguous (channel, tile block) ranges; debug option 1 = off
    if (is == 2) wino_window_kernel<3, 2, TMAJOR, MODE, T><<<grid, 256, 0, st>>>(x, B, C, H, W, off, p.TY, p.TX, p.Tp, Cp, V, remap);
    else wino_window_kernel<3, 1, TMAJOR, MODE, T><<<grid, 256, 0, st>>>(x, B, C, H, W, off, p.TY, p.TX, p.Tp, Cp, V, remap);
}

// x [B,Cin,H,W], w [Cout,Cin,4,4], y / dy [B,Cout,Ho,Wo]
int launch_winograd_dil(int geom, int mode, const void* a, const void* b2, void* out, int B, int Cin, int H, int W, int Cout,
                        void* ws, size_t ws_bytes, hipStream_t st, ConvArith ar = ConvArith{0, false, false})
{
    DilPlan p;
    if (int rc = dil_plan(geom, mode, B, Cin, H, W, Cout, &p, ar.math)) return rc;
    if (ws_bytes < p.total_bytes) return fail(IPSR_ERR_WORKSPACE, "4x4 winograd: workspace %zu < %zu", ws_bytes, p.total_bytes);
    if (!arith_ok(ar)) return fail(IPSR_ERR_INVALID, "4x4 winograd: arithmetic %d", ar.math);
    const int Ho = p.Ho, Wo = p.Wo;
    const int xis = geom == 0 ? 2 : 1, xoff = geom == 0 ? -3 : -1;        // X[i] = x[xis * i + xoff]
    Carver cv(ws, ws_bytes);
    float* A = cv.take<float>(p.a_floats);
    float* Bv = cv.take<float>(p.b_floats);
    float* Mo = cv.take<float>(p.m_floats);
    if (mode == 0 || mode == 1) {
        // mode 0: a = x, b2 = w, out = y.   mode 1: a = dy, b2 = w, out = dx:
        //   geom 0: only the odd rows / columns receive gradient, dx[2q+1] = sum_r' w[3-r'] dy[q - 1 + r']
        //   geom 1: dx[i] = sum_r' w[3-r'] dy[i - 2 + r'] at every position
        //   reduction over Cout: element (c = co, k = ci) of w[co][ci][r][s] at co*Cin*16 + ci*16, taps flipped
        const int red = mode == 0 ? Cin : Cout, prod = mode == 0 ? Cout : Cin;
        if (mode == 1 && geom == 0 && hipMemsetAsync(out, 0, (size_t)B * Cin * H * W * (ar.out_bf16 ? 2 : 4), st) != hipSuccess)
            return fail(IPSR_ERR_LAUNCH, "dilated winograd: hipMemsetAsync failed");
        const float* w = static_cast<const float*>(b2);
        with_mode(ar.math, [&](auto M) {
            constexpr int MODE = decltype(M)::value;
            if (mode == 0) wino4_filter_kernel<MODE><<<op_grid(false, MODE, p.Kp, Cin), 256, 0, st>>>(w, Cin, Cout, p.Kp, 16, (long)Cin * 16, 0, A);
            else wino4_filter_kernel<MODE><<<op_grid(false, MODE, p.Kp, Cout), 256, 0, st>>>(w, Cout, Cin, p.Kp, (long)Cin * 16, 16, 1, A);
            with_type(ar.in_bf16, [&](auto* tag) {
                using T = ELEM_T(tag);
                if (mode == 0) launch_window3<false, MODE, T>(xis, st, static_cast<const T*>(a), B, Cin, H, W, xoff, p, 0, Bv);
                else launch_window3<false, MODE, T>(1, st, static_cast<const T*>(a), B, Cout, Ho, Wo, geom == 0 ? -1 : -2, p, 0, Bv);
            });
        });
        if (int rc = check_launch("wino_window_kernel")) return rc;
        launch_wino_gemm(ar.math, A, Bv, red, p.Kp, p.Tp, p.sp, Mo, st, 72.0 * red * prod * p.T);
        if (int rc = check_launch("wino_gemm_kernel")) return rc;
        with_type(ar.out_bf16, [&](auto* tag) {
            using T = ELEM_T(tag);
            if (mode == 0) wino3_output_kernel<T><<<dim3(cdiv(p.T, 256), prod), 256, 0, st>>>(Mo, p.sp, B, Cout, p.Kp, Ho, Wo, p.TY, p.TX, p.Tp, Ho, Wo, 1, 0, static_cast<T*>(out));
            else wino3_output_kernel<T><<<dim3(cdiv(p.T, 256), prod), 256, 0, st>>>(Mo, p.sp, B, Cin, p.Kp, p.Gy, p.Gx, p.TY, p.TX, p.Tp, H, W,
                                                                                    geom == 0 ? 2 : 1, geom == 0 ? 1 : 0, static_cast<T*>(out));
        });
        return check_launch("wino3_output_kernel");
    }
    // mode 2: a = x, b2 = dy, out = dW [Cout][Cin][4][4] (fp32)
    with_mode(ar.math, [&](auto M) {
        constexpr int MODE = decltype(M)::value;
        with_type(ar.in_bf16, [&](auto* tag) {
            using T = ELEM_T(tag);
            wino_wrw_tile3_kernel<MODE, T><<<op_grid(true, MODE, p.Tp, p.Kp), 256, 0, st>>>(static_cast<const T*>(b2), B, Cout, Ho, Wo, p.TY, p.TX, p.Tp, p.Kp, A);
            launch_window3<true, MODE, T>(xis, st, static_cast<const T*>(a), B, Cin, H, W, xoff, p, p.Cp, Bv);
        });
    });
    if (int rc = check_launch("wino_window_kernel")) return rc;
    launch_wino_gemm(ar.math, A, Bv, p.Tp, p.Kp, p.Cp, p.sp, Mo, st, 72.0 * p.T * Cout * Cin);
    if (int rc = check_launch("wino_gemm_kernel")) return rc;
    wino_wrw_output4_kernel<<<dim3(cdiv(Cin, 256), Cout), 256, 0, st>>>(Mo, p.sp, Cout, Cin, p.Kp, p.Cp, static_cast<float*>(out));
    return check_launch("wino_wrw_output4_kernel");
}

// ===================================================================================================
// The 4x4 STRIDE-2 pad-1 layers — every down convolution of netP / netD / netF (Conv2d, models/networks.py:404-432, 470-495,
// 510-515) and every up convolution of netP / netG (ConvTranspose2d, :235-243, 420-428): a third of the training step — by
// Winograd F(5x5, 2x2) on the two polyphase components of the fine grid.
//
// One geometry serves both modules.  "fine" = the 2n-grid tensor (x of Conv2d, y of ConvTranspose2d), "coarse" = the n-grid
// one; a fine position f and a coarse position o meet through tap r when f = 2o + r - 1; the weight is [coarse ch][fine ch][4][4]
// for both modules (Conv2d: [Cout][Cin], ConvTranspose2d: [Cin][Cout]).  In 1-D:
//   mode 0  fine -> coarse  (Conv2d forward, ConvTranspose2d backward-data)
//           out[o] = sum_e sum_a w[2a+e+1] P_e[o+a],  P_e[i] = fine[2i+e], e in {-1,0}: a 2-tap stride-1 correlation on each of the
//           two phases (four in 2-D), reduced over 4 x channels.  F(5,2): 5 outputs from a 6-wide window, 6 multiplies instead of 10.
//   mode 1  coarse -> fine  (ConvTranspose2d forward, Conv2d backward-data)
//           fine[2i]   = w[3] in[i-1] + w[1] in[i]       both phases read the SAME window in[5t-1 .. 5t+4]: one input transform,
//           fine[2i+1] = w[2] in[i]   + w[0] in[i+1]     four filter sets (the GEMM produces 4 x channels), phase e of tile t
//                                                          holds the outputs i = 5t + m - e, m = 0..4
//   mode 2  weight gradient  dW[2a+e+1] = sum_o coarse[o] P_e[o+a]: F(2x2, 5x5) — 5x5 tiles of the coarse tensor through G5,
//           the windows of mode 0 (tile-major), reduced over all tiles by the GEMM, A2^T back to the 2x2 taps of each phase.
// 100 multiply-adds per 5x5 tile, channel pair and phase become 36: 2.8x fewer matrix-core flops than the direct form, through
// the same 36 GEMMs (same interpolation points 0, +-1, +-2, inf, hence the same B^T) as everything above.
__device__ __forceinline__ void wino_g2(const float g[2], float u[6])        // G2 g   (6x2)
{
    u[0] = 0.25f * g[0];
    u[1] = (-1.0f / 6.0f) * (g[0] + g[1]);
    u[2] = (-1.0f / 6.0f) * (g[0] - g[1]);
    u[3] = (1.0f / 24.0f) * g[0] + (1.0f / 12.0f) * g[1];
    u[4] = (1.0f / 24.0f) * g[0] - (1.0f / 12.0f) * g[1];
    u[5] = g[1];
}
__device__ __forceinline__ void wino_at5(const float m[6], float y[5])       // A5^T m (5x6)
{
    y[0] = m[0] + m[1] + m[2] + m[3] + m[4];
    y[1] = m[1] - m[2] + 2.0f * m[3] - 2.0f * m[4];
    y[2] = m[1] + m[2] + 4.0f * m[3] + 4.0f * m[4];
    y[3] = m[1] - m[2] + 8.0f * m[3] - 8.0f * m[4];
    y[4] = m[1] + m[2] + 16.0f * m[3] + 16.0f * m[4] + m[5];
}
__device__ __forceinline__ void wino_g5(const float e[5], float u[6])        // G5 e   (6x5)
{
    u[0] = 0.25f * e[0];
    u[1] = (-1.0f / 6.0f) * (e[0] + e[1] + e[2] + e[3] + e[4]);
    u[2] = (-1.0f / 6.0f) * (e[0] - e[1] + e[2] - e[3] + e[4]);
    u[3] = (1.0f / 24.0f) * e[0] + (1.0f / 12.0f) * e[1] + (1.0f / 6.0f) * e[2] + (1.0f / 3.0f) * e[3] + (2.0f / 3.0f) * e[4];
    u[4] = (1.0f / 24.0f) * e[0] - (1.0f / 12.0f) * e[1] + (1.0f / 6.0f) * e[2] - (1.0f / 3.0f) * e[3] + (2.0f / 3.0f) * e[4];
    u[5] = e[4];
}
__device__ __forceinline__ void wino_at2(const float m[6], float y[2])       // A2^T m (2x6)
{
    y[0] = m[0] + m[1] + m[2] + m[3] + m[4];
    y[1] = m[1] - m[2] + 2.0f * m[3] - 2.0f * m[4] + m[5];
}

// U[xi][r][q] = (G2 g G2^T)[xi] of the 2x2 sub-filter of each phase.  W = [coarse ch kc][fine ch cf][4][4].
//   form 0 (mode 0): reduction row r = (phase, cf) of 4*Cf, produced column q = kc < Kc;   g[a][b] = w[2a+ey][2b+ex]
//   form 1 (mode 1): reduction row r = kc,              produced column q = (phase, cf) of 4*Cf; g[a][b] = w[3-ey-2a][3-ex-2b]
// One thread per (kc, cf): its 16 taps are one 64-byte read, the four phases' 4 x 36 numbers go out with the q index
// fastest across the threads (form 0: kc, form 1: cf), i.e. coalesced.  Columns q in [Q, Qp) are zero-filled by the same threads.
__global__ void __launch_bounds__(256) wino52_filter_kernel(const float* __restrict__ W, int Kc, int Cf, int Qp, int form, float* __restrict__ U)
{
    const int fast = blockIdx.x * 256 + threadIdx.x, slow = blockIdx.y;
    const int nfast = form == 0 ? Kc : Cf;
    const int R = form == 0 ? 4 * Cf : Kc, Q = form == 0 ? Kc : 4 * Cf;
    const size_t plane = (size_t)R * Qp;
    if (fast >= nfast) {
        // padding columns: form 0: q = fast in [Kc, Qp) of rows (ph, slow); form 1: the tail [4*Cf, Qp) of row kc = slow, once
        if (form == 0) {
            if (fast < Qp)
                for (int ph = 0; ph < 4; ++ph)
                    for (int xi = 0; xi < 36; ++xi) U[(size_t)xi * plane + (size_t)(ph * Cf + slow) * Qp + fast] = 0.0f;
        } else {
            const int q = Q + (fast - nfast);
            if (q < Qp)
                for (int xi = 0; xi < 36; ++xi) U[(size_t)xi * plane + (size_t)slow * Qp + q] = 0.0f;
        }
        return;
    }
    const int kc = form == 0 ? fast : slow, cf = form == 0 ? slow : fast;
    float w[4][4];
    const float4* wp = reinterpret_cast<const float4*>(W + ((size_t)kc * Cf + cf) * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float4 v = wp[i]; w[i][0] = v.x; w[i][1] = v.y; w[i][2] = v.z; w[i][3] = v.w; }
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
        const int ey = ph >> 1, ex = ph & 1;
        float g[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) g[a][b] = form == 0 ? w[2 * a + ey][2 * b + ex] : w[3 - ey - 2 * a][3 - ex - 2 * b];
        float t[6][2], u[6][6];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float col[2] = {g[0][b], g[1][b]};
            float o[6];
            wino_g2(col, o);
#pragma unroll
            for (int i = 0; i < 6; ++i) t[i][b] = o[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) wino_g2(t[i], u[i]);
        const size_t r = form == 0 ? (size_t)ph * Cf + cf : kc, q = form == 0 ? kc : (size_t)ph * Cf + cf;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) U[(size_t)(i * 6 + j) * plane + r * Qp + q] = u[i][j];
    }
}

// split-bf16 twin (see store_split): workgroup = 8 reduction indices x 32 columns of (kc, cf) pairs, the four phases stored one after
// the other.  form 0: reduction (ph, cf) -> blocks of 8 cf, columns kc (grid (Qp/32, Cf/8), padding columns are zero);
// form 1: reduction kc -> blocks of 8 kc, columns (ph, cf) (grid (Cf/32, Kc/8); host requires Cf % 32 == 0 and Qp == 4*Cf).
template <int NPL>
__global__ void __launch_bounds__(256) wino52_filter_split_kernel(const float* __restrict__ W, int Kc, int Cf, int Qp, int form,
                                                                  unsigned short* __restrict__ U)
{
    __shared__ __attribute__((aligned(16))) unsigned short stage[split_stage_elems<NPL>()];
    const int pk = threadIdx.x >> 5, rl = threadIdx.x & 31;
    const int kc = form == 0 ? blockIdx.x * SPLIT_ROWS + rl : blockIdx.y * 8 + pk;
    const int cf = form == 0 ? blockIdx.y * 8 + pk : blockIdx.x * SPLIT_ROWS + rl;
    const bool live = kc < Kc && cf < Cf;
    float w[4][4];
    const float4* wp = reinterpret_cast<const float4*>(W + ((size_t)(live ? kc : 0) * Cf + (live ? cf : 0)) * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 v = live ? wp[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        w[i][0] = v.x; w[i][1] = v.y; w[i][2] = v.z; w[i][3] = v.w;
    }
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
        const int ey = ph >> 1, ex = ph & 1;
        float g[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) g[a][b] = form == 0 ? w[2 * a + ey][2 * b + ex] : w[3 - ey - 2 * a][3 - ex - 2 * b];
        float t[6][2], u[6][6];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float col[2] = {g[0][b], g[1][b]};
            float o[6];
            wino_g2(col, o);
#pragma unroll
            for (int i = 0; i < 6; ++i) t[i][b] = o[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) wino_g2(t[i], u[i]);
        if (form == 0) store_split<NPL>(stage, u, pk, rl, U, (size_t)4 * Cf / 8, Qp, (size_t)ph * Cf / 8 + blockIdx.y, (size_t)blockIdx.x * SPLIT_ROWS);
        else store_split<NPL>(stage, u, pk, rl, U, (size_t)Kc / 8, Qp, blockIdx.y, (size_t)ph * Cf + (size_t)blockIdx.x * SPLIT_ROWS);
    }
}

// windows at tile stride 5: V[xi][cc][t] (TMAJOR = false) or V[xi][t][cc] (true), cc = (phase, c) of nphase*C.
//   nphase = 4, IS = 2: P_e[5t + i] = x[2(5t+i) + e], e = (ey-1, ex-1)  (modes 0 and 2, x = the fine tensor)
//   nphase = 1, IS = 1: x[5t + i - 1]                                    (mode 1, x = the coarse tensor)
template <int IS, bool TMAJOR, int MODE, typename TIN>
__global__ void __launch_bounds__(256) wino5_window_kernel(const TIN* __restrict__ x, int B, int C, int H, int Wd, int nphase, int TY, int TX,
                                                           int Tp, int Cp, void* __restrict__ V, int remap)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[op_smem_bytes<TMAJOR, MODE>()];
    const int Ctot = nphase * C;
    const OpIdx ix = op_index<TMAJOR, MODE>(!TMAJOR && remap != 0);
    const int t = ix.t, cc = ix.c;
    if (!TMAJOR && MODE == 0 && t >= Tp) return;
    const int T = B * TY * TX;
    const bool live = t < T && cc < Ctot;
    int b = 0, ty = 0, tx = 0, ph = 0, c = 0;
    if (live) { b = t / (TY * TX); const int rem = t - b * TY * TX; ty = rem / TX; tx = rem - ty * TX; ph = cc / C; c = cc - ph * C; }
    const int offy = nphase == 4 ? (ph >> 1) - 1 : -1, offx = nphase == 4 ? (ph & 1) - 1 : -1;
    const TIN* xp = x + ((size_t)b * C + c) * H * Wd;
    float d[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int yy = IS * (5 * ty + i) + offy;
        const bool yok = live && (unsigned)yy < (unsigned)H;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int xx = IS * (5 * tx + j) + offx;
            d[i][j] = (yok && (unsigned)xx < (unsigned)Wd) ? ld1(xp, (size_t)yy * Wd + xx) : 0.0f;
        }
    }
    float w[6][6], v[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const float col[6] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]};
        float o[6];
        wino_bt(col, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) w[i][j] = o[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) wino_bt(w[i], v[i]);
    op_emit<TMAJOR, MODE>(smem, v, ix, V, TMAJOR ? Cp : Ctot, Tp);
}

__device__ __forceinline__ void wino_at5_2d(const float m[6][6], float o[5][5])
{
    float w[5][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const float col[6] = {m[0][j], m[1][j], m[2][j], m[3][j], m[4][j], m[5][j]};
        float r5[5];
        wino_at5(col, r5);
#pragma unroll
        for (int i = 0; i < 5; ++i) w[i][j] = r5[i];
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) wino_at5(w[i], o[i]);
}

// mode 0: y[b][k][5ty+i][5tx+j] = (A5^T M A5)[i][j]
template <typename TOUT>
__global__ void __launch_bounds__(256) wino5_output_kernel(const float* __restrict__ Mo, WinoSplit split, int B, int K, int Kp, int Ho, int Wo,
                                                           int TY, int TX, int Tp, TOUT* __restrict__ y)
{
    const int t = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
    const int T = B * TY * TX;
    if (t >= T) return;
    float m[6][6], o[5][5];
    wino_load_sum(Mo, split, (size_t)Kp * Tp, (size_t)k * Tp + t, m);
    wino_at5_2d(m, o);
    const int b = t / (TY * TX), rem = t - b * TY * TX;
    const int ty = rem / TX, tx = rem - ty * TX;
    TOUT* yp = y + ((size_t)b * K + k) * Ho * Wo;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int oy = 5 * ty + i, ox = 5 * tx + j;
            if (oy < Ho && ox < Wo) st1(yp, (size_t)oy * Wo + ox, o[i][j]);
        }
}

// mode 1: the four phases of tile t and fine channel k (GEMM rows ph*K + k) -> the 10 x 10 block of y [B,K,2n_h,2n_w] they
// interleave into:  y[2(5ty+m) - ey][2(5tx+m') - ex] = (A5^T M_ph A5)[m][m']  wherever the coarse index 5t + m - e lies in [0, n),
// i.e. the block starts at (10ty - 1, 10tx - 1).  One phase per thread (wave = phase, lane = one of 64 consecutive tiles: the 36
// loads of a lane are coalesced over t); the results meet in LDS and leave as ROWS — 64 tiles x 10 columns of one fine row per
// sweep, contiguous wherever the tiles are neighbours — instead of 25 stores per thread at 8-byte pitch inside a 40-byte thread
// stride (1.7-2.4 TB/s effective).  Measured per layer (profiles/r03_hipconv_k4s2.txt): ConvTranspose2d forward 64 -> 64 @128 -> 256
// 0.299 -> 0.210 ms, 128 -> 128 @64 0.161 -> 0.132, Conv2d input gradient 64 <- 128 @128 0.109 -> 0.096.  The same trick on the
// dense 5 x 5 / 3 x 3 output tiles (5 or 3 consecutive floats per thread and row already) measured no gain and was not kept.
constexpr int W5R_TILES = 64;
template <typename TOUT>
__global__ void __launch_bounds__(256) wino5_output_rows_kernel(const float* __restrict__ Mo, WinoSplit split, int B, int K, int Kp, int nh, int nw,
                                                                int TY, int TX, int Tp, TOUT* __restrict__ y)
{
    __shared__ float blk[W5R_TILES][101];                     // [tile][10 x 10], odd pitch: lane-per-tile writes hit distinct banks
    __shared__ int org_y[W5R_TILES], org_x[W5R_TILES], org_b[W5R_TILES];
    const int tl = threadIdx.x & 63, ph = threadIdx.x >> 6, ey = ph >> 1, ex = ph & 1;
    const int t0 = blockIdx.x * W5R_TILES, t = t0 + tl, k = blockIdx.y;
    const int T = B * TY * TX;
    if (t < T) {
        float m[6][6], o[5][5];
        wino_load_sum(Mo, split, (size_t)Kp * Tp, (size_t)(ph * K + k) * Tp + t, m);
        wino_at5_2d(m, o);
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) blk[tl][(2 * i - ey + 1) * 10 + (2 * j - ex + 1)] = o[i][j];
        if (ph == 0) {
            const int b = t / (TY * TX), rem = t - b * TY * TX;
            const int ty = rem / TX, tx = rem - ty * TX;
            org_b[tl] = b; org_y[tl] = 10 * ty - 1; org_x[tl] = 10 * tx - 1;
        }
    }
    __syncthreads();
    const int Hy = 2 * nh, Wy = 2 * nw;
    const int ntile = min(W5R_TILES, T - t0);
    for (int e = threadIdx.x; e < 10 * 10 * W5R_TILES; e += 256) {
        const int r = e / (10 * W5R_TILES), xs = e - r * (10 * W5R_TILES);
        const int tile = xs / 10, c = xs - tile * 10;
        if (tile >= ntile) continue;
        const int fy = org_y[tile] + r, fx = org_x[tile] + c;
        if (fy < 0 || fy >= Hy || fx < 0 || fx >= Wy) continue;
        st1(y + ((size_t)org_b[tile] * K + k) * (size_t)Hy * Wy, (size_t)fy * Wy + fx, blk[tile][r * 10 + c]);
    }
}

// mode 2: 5x5 tiles of the coarse tensor -> Et[xi][t][k] = (G5 e G5^T)[xi]
template <int MODE, typename TIN>
__global__ void __launch_bounds__(256) wino_wrw_tile5_kernel(const TIN* __restrict__ dy, int B, int K, int Ho, int Wo, int TY, int TX,
                                                             int Tp, int Kp, void* __restrict__ Et)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[op_smem_bytes<true, MODE>()];
    const OpIdx ix = op_index<true, MODE>(false);
    const int t = ix.t, k = ix.c, T = B * TY * TX;
    const bool live = t < T && k < K;
    int b = 0, ty = 0, tx = 0;
    if (live) { b = t / (TY * TX); const int rem = t - b * TY * TX; ty = rem / TX; tx = rem - ty * TX; }
    const TIN* dp = dy + ((size_t)b * K + (live ? k : 0)) * Ho * Wo;
    float e[5][5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int yy = 5 * ty + i, xx = 5 * tx + j;
            e[i][j] = (live && yy < Ho && xx < Wo) ? ld1(dp, (size_t)yy * Wo + xx) : 0.0f;
        }
    float w[6][5], v[6][6];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const float col[5] = {e[0][j], e[1][j], e[2][j], e[3][j], e[4][j]};
        float o[6];
        wino_g5(col, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) w[i][j] = o[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) wino_g5(w[i], v[i]);
    op_emit<true, MODE>(smem, v, ix, Et, Kp, Tp);
}

// mode 2: dW[kc][cf][2a+ey][2b+ex] = (A2^T Mw[:][kc][(ph, cf)] A2)[a][b]; one phase per blockIdx.z
__global__ void __launch_bounds__(256) wino_wrw_output2_kernel(const float* __restrict__ Mw, WinoSplit split, int Kc, int Cf, int Kp, int Cp,
                                                               float* __restrict__ dW)
{
    const int cf = blockIdx.x * 256 + threadIdx.x, kc = blockIdx.y, ph = blockIdx.z;
    if (cf >= Cf) return;
    float m[6][6], w[2][6], o[2][2];
    wino_load_sum(Mw, split, (size_t)Kp * Cp, (size_t)kc * Cp + (size_t)ph * Cf + cf, m);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const float col[6] = {m[0][j], m[1][j], m[2][j], m[3][j], m[4][j], m[5][j]};
        float r2[2];
        wino_at2(col, r2);
        w[0][j] = r2[0]; w[1][j] = r2[1];
    }
    wino_at2(w[0], o[0]);
    wino_at2(w[1], o[1]);
    const int ey = ph >> 1, ex = ph & 1;
    float* dst = dW + ((size_t)kc * Cf + cf) * 16;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) dst[(2 * a + ey) * 4 + 2 * b + ex] = o[a][b];
}

struct S2Plan { int TY, TX, T, Tp, Kp, Cp, red; WinoSplit sp; size_t a_floats, b_floats, m_floats, total_bytes; };

static int s2_plan(int mode, int B, int Kc, int Cf, int nh, int nw, S2Plan* p, int math)
{
    if (mode < 0 || mode > 2) return fail(IPSR_ERR_INVALID, "4x4 stride-2 winograd: mode %d", mode);
    p->TY = mode == 1 ? (nh + 5) / 5 : (nh + 4) / 5;
    p->TX = mode == 1 ? (nw + 5) / 5 : (nw + 4) / 5;
    p->T = B * p->TY * p->TX;
    p->Tp = (p->T + WG_BN - 1) / WG_BN * WG_BN;
    size_t m1;
    if (mode == 2) {
        p->Kp = wino_rows_padded(Kc, math);
        p->Cp = (4 * Cf + WG_BN - 1) / WG_BN * WG_BN;
        p->red = p->Tp;
        p->a_floats = (size_t)36 * p->Tp * p->Kp * 3 / 2;
        p->b_floats = (size_t)36 * p->Tp * p->Cp * 3 / 2;
        m1 = (size_t)36 * p->Kp * p->Cp;
        p->sp = wino_choose_split(36 * wino_row_tiles(p->Kp) * (p->Cp / WG_BN), p->Tp / WG_BK, m1 * 4);
    } else {
        p->red = mode == 0 ? 4 * Cf : Kc;
        if (p->red % WG_BK != 0)
            return fail(IPSR_ERR_UNSUPPORTED, "4x4 stride-2 winograd: reduction length %d is not a multiple of %d", p->red, WG_BK);
        const int prod = mode == 0 ? Kc : 4 * Cf;
        p->Kp = wino_rows_padded(prod, math);
        p->Cp = p->red;
        p->a_floats = (size_t)36 * p->red * p->Kp * 3 / 2;
        p->b_floats = (size_t)36 * p->red * p->Tp * 3 / 2;
        m1 = (size_t)36 * p->Kp * p->Tp;
        p->sp = wino_choose_split(36 * wino_row_tiles(p->Kp) * (p->Tp / WG_BN), p->red / WG_BK, m1 * 4);
    }
    p->m_floats = m1 * p->sp.slabs();
    p->total_bytes = align_up(p->a_floats * 4, 256) + align_up(p->b_floats * 4, 256) + align_up(p->m_floats * 4, 256) + 256;
    return IPSR_OK;
}

size_t winograd_s2_ws_bytes(int mode, int B, int Kc, int Cf, int nh, int nw)
{
    S2Plan p, q;
    if (s2_plan(mode, B, Kc, Cf, nh, nw, &p, -1) != IPSR_OK || s2_plan(mode, B, Kc, Cf, nh, nw, &q, 0) != IPSR_OK) return 0;
    return p.total_bytes > q.total_bytes ? p.total_bytes : q.total_bytes;
}

// fine [B,Cf,2nh,2nw], coarse [B,Kc,nh,nw], w / dW [Kc][Cf][4][4].
// mode 0: a = fine, b2 = w, out = coarse.   mode 1: a = coarse, b2 = w, out = fine.   mode 2: a = fine, b2 = coarse, out = dW.
int launch_winograd_s2(int mode, const void* a, const void* b2, void* out, int B, int Kc, int Cf, int nh, int nw,
                       void* ws, size_t ws_bytes, hipStream_t st, ConvArith ar = ConvArith{0, false, false})
{
    S2Plan p;
    if (int rc = s2_plan(mode, B, Kc, Cf, nh, nw, &p, ar.math)) return rc;
    if (ws_bytes < p.total_bytes) return fail(IPSR_ERR_WORKSPACE, "4x4 stride-2 winograd: workspace %zu < %zu", ws_bytes, p.total_bytes);
    if (!arith_ok(ar)) return fail(IPSR_ERR_INVALID, "4x4 stride-2 winograd: arithmetic %d", ar.math);
    // the split filter transform stores whole 8-channel blocks per phase (form 0) / whole 32-column blocks per phase (form 1):
    // shapes it cannot express run the fp32 arithmetic instead (never less accurate)
    if (ar.math && ((mode == 0 && Cf % 8 != 0) || (mode == 1 && (Cf % SPLIT_ROWS != 0 || p.Kp != 4 * Cf || Kc % 8 != 0)))) ar.math = 0;
    Carver cv(ws, ws_bytes);
    float* A = cv.take<float>(p.a_floats);
    float* Bv = cv.take<float>(p.b_floats);
    float* Mo = cv.take<float>(p.m_floats);
    const int remap = !debug_option(1);          // XCD-contiguous (channel, tile block) ranges; debug option 1 = off
    if (mode == 0 || mode == 1) {
        const float* w = static_cast<const float*>(b2);
        if (ar.math == 0) {
            if (mode == 0) wino52_filter_kernel<<<dim3(cdiv(p.Kp, 256), Cf), 256, 0, st>>>(w, Kc, Cf, p.Kp, 0, A);
            else wino52_filter_kernel<<<dim3(cdiv(Cf + (p.Kp - 4 * Cf), 256), Kc), 256, 0, st>>>(w, Kc, Cf, p.Kp, 1, A);
        } else {
            const dim3 fg = mode == 0 ? dim3(p.Kp / SPLIT_ROWS, Cf / 8) : dim3(Cf / SPLIT_ROWS, Kc / 8);
            if (ar.math == 2) wino52_filter_split_kernel<2><<<fg, 256, 0, st>>>(w, Kc, Cf, p.Kp, mode, reinterpret_cast<unsigned short*>(A));
            else wino52_filter_split_kernel<3><<<fg, 256, 0, st>>>(w, Kc, Cf, p.Kp, mode, reinterpret_cast<unsigned short*>(A));
        }
        with_mode(ar.math, [&](auto M) {
            constexpr int MODE = decltype(M)::value;
            with_type(ar.in_bf16, [&](auto* tag) {
                using T = ELEM_T(tag);
                if (mode == 0) wino5_window_kernel<2, false, MODE, T><<<op_grid(false, MODE, p.Tp, p.red), 256, 0, st>>>(static_cast<const T*>(a), B, Cf, 2 * nh, 2 * nw, 4, p.TY, p.TX, p.Tp, 0, Bv, remap);
                else wino5_window_kernel<1, false, MODE, T><<<op_grid(false, MODE, p.Tp, p.red), 256, 0, st>>>(static_cast<const T*>(a), B, Kc, nh, nw, 1, p.TY, p.TX, p.Tp, 0, Bv, remap);
            });
        });
        if (int rc = check_launch("wino5_window_kernel")) return rc;
        launch_wino_gemm(ar.math, A, Bv, p.red, p.Kp, p.Tp, p.sp, Mo, st, 72.0 * p.red * (mode == 0 ? Kc : 4 * Cf) * p.T);
        if (int rc = check_launch("wino_gemm_kernel")) return rc;
        with_type(ar.out_bf16, [&](auto* tag) {
            using T = ELEM_T(tag);
            if (mode == 0) wino5_output_kernel<T><<<dim3(cdiv(p.T, 256), Kc), 256, 0, st>>>(Mo, p.sp, B, Kc, p.Kp, nh, nw, p.TY, p.TX, p.Tp, static_cast<T*>(out));
            else wino5_output_rows_kernel<T><<<dim3(cdiv(p.T, W5R_TILES), Cf), 256, 0, st>>>(Mo, p.sp, B, Cf, p.Kp, nh, nw, p.TY, p.TX, p.Tp, static_cast<T*>(out));
        });
        return check_launch("wino5_output_kernel");
    }
    // mode 2: a = fine, b2 = coarse, out = dW (fp32)
    with_mode(ar.math, [&](auto M) {
        constexpr int MODE = decltype(M)::value;
        with_type(ar.in_bf16, [&](auto* tag) {
            using T = ELEM_T(tag);
            wino_wrw_tile5_kernel<MODE, T><<<op_grid(true, MODE, p.Tp, p.Kp), 256, 0, st>>>(static_cast<const T*>(b2), B, Kc, nh, nw, p.TY, p.TX, p.Tp, p.Kp, A);
            wino5_window_kernel<2, true, MODE, T><<<op_grid(true, MODE, p.Tp, p.Cp), 256, 0, st>>>(static_cast<const T*>(a), B, Cf, 2 * nh, 2 * nw, 4, p.TY, p.TX, p.Tp, p.Cp, Bv, 0);
        });
    });
    if (int rc = check_launch("wino5_window_kernel")) return rc;
    launch_wino_gemm(ar.math, A, Bv, p.Tp, p.Kp, p.Cp, p.sp, Mo, st, 72.0 * p.T * Kc * 4 * Cf);
    if (int rc = check_launch("wino_gemm_kernel")) return rc;
    wino_wrw_output2_kernel<<<dim3(cdiv(Cf, 256), Kc, 4), 256, 0, st>>>(Mo, p.sp, Kc, Cf, p.Kp, p.Cp, static_cast<float*>(out));
    return check_launch("wino_wrw_output2_kernel");
}

// ===================================================================================================
// Small maps.  The inner levels of both U-Nets and netF (512-1024 channels on 8x8 ... 1x1: models/networks.py:220-259, 404-432,
// 510-515) are ~85 convolution calls per training step whose arithmetic is a skinny GEMM between a 9-33 MB weight tensor and
// a handful of activations: MIOpen spends 45-150 us on each (layout transposes, zero fills, tiles made for large maps).
// Here the weight tensor Wm = [R][Q] (R = its first channel dimension, Q = second channel dimension x taps, contiguous: Conv2d
// [Cout][(Cin,t)], ConvTranspose2d [Cin][(Cout,t)]) is STREAMED ONCE from where it lies, 16 bytes per lane straight into
// MFMA operands — no LDS, no packing — and the P = B*Ho*Wo <= 512 positions ride on the 32-wide N side of 32x32x2 MFMAs:
//   DATA (Conv2d backward-data, ConvTranspose2d forward)   Mcol[q][p] = sum_r Wm[r][q] * in[r][p],  then col2im over the taps
//   FWD  (Conv2d forward, ConvTranspose2d backward-data)    y[r][p]    = sum_q Wm[r][q] * col(fine)[q][p]
//   WRW  (weight gradient of either)                        dW[r][q]   = sum_p coarse[r][p] * col(fine)[q][p]   (native layout, one pass)
// p runs over the grid on R's side ("coarse": the conv's output side), "fine" is the grid on Q's side.  The row / column labels
// of an MFMA tile are free, so a lane's float4 of four consecutive q feeds four MFMAs whose tiles are q = q0 + 4m + j: every
// weight byte is fetched by exactly one coalesced 16-byte load.  The streams are bandwidth bound (16.8 MB in ~5 us); the
// reduction is cut over workgroups into slabs that the tiny post-pass (col2im / layout) adds in order (deterministic).
typedef float f32x4v __attribute__((ext_vector_type(4)));


__global__ void __launch_bounds__(256) sm_to_cn_kernel(const float* __restrict__ x, int B, int C, int HW, int Tp, float* __restrict__ out)
{
    const int n = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (n >= Tp) return;
    float v = 0.0f;
    if (n < B * HW) { const int b = n / HW, p = n - b * HW; v = x[((size_t)b * C + c) * HW + p]; }
    out[(size_t)c * Tp + n] = v;
}

__global__ void __launch_bounds__(256) sm_to_nc_kernel(const float* __restrict__ x, int B, int C, int HW, float* __restrict__ out)
{
    const int c = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;              // rows n >= B*HW are zero padding of the reduction
    if (c >= C) return;
    float v = 0.0f;
    if (n < B * HW) { const int b = n / HW, p = n - b * HW; v = x[((size_t)b * C + c) * HW + p]; }
    out[(size_t)n * C + c] = v;
}

__device__ __forceinline__ float sm_tap(const float* __restrict__ f, int B, int C, int Hf, int Wf, int Ho, int Wo, int k, int st, int pad, int dil,
                                        int n, int col)
{
    if (n >= B * Ho * Wo) return 0.0f;
    const int kk = k * k;
    const int c = col / kk, t = col - c * kk, r = t / k, q = t - r * k;
    const int b = n / (Ho * Wo), o = n - b * Ho * Wo, oy = o / Wo, ox = o - oy * Wo;
    const int fy = oy * st - pad + r * dil, fx = ox * st - pad + q * dil;
    return ((unsigned)fy < (unsigned)Hf && (unsigned)fx < (unsigned)Wf) ? f[(((size_t)b * C + c) * Hf + fy) * Wf + fx] : 0.0f;
}

// V[n][(c,t)] = fine[b][c][oy*s - pad + r*dil][ox*s - pad + q*dil]  (0 outside; rows n >= B*Ho*Wo zero)
__global__ void __launch_bounds__(256) sm_im2col_nt_kernel(const float* __restrict__ f, int B, int C, int Hf, int Wf, int Ho, int Wo,
                                                           int k, int st, int pad, int dil, float* __restrict__ out)
{
    const int col = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
    const int ncol = C * k * k;
    if (col >= ncol) return;
    out[(size_t)n * ncol + col] = sm_tap(f, B, C, Hf, Wf, Ho, Wo, k, st, pad, dil, n, col);
}

// the same as [(c,t)][n] with row length Tp (columns n >= B*Ho*Wo zero)
__global__ void __launch_bounds__(256) sm_im2col_cn_kernel(const float* __restrict__ f, int B, int C, int Hf, int Wf, int Ho, int Wo,
                                                           int k, int st, int pad, int dil, int Tp, float* __restrict__ out)
{
    const int n = blockIdx.x * 256 + threadIdx.x, col = blockIdx.y;
    if (n >= Tp) return;
    out[(size_t)col * Tp + n] = sm_tap(f, B, C, Hf, Wf, Ho, Wo, k, st, pad, dil, n, col);
}

// out[b][c][fy][fx] = sum_{slabs} sum_{(r,q): fy = oy*s - pad + r*dil, fx = ox*s - pad + q*dil} M[(c,t)][(b,oy,ox)]
__global__ void __launch_bounds__(256) sm_col2im_kernel(const float* __restrict__ M, int nslab, size_t slab_stride, int B, int C, int Hf, int Wf,
                                                        int Ho, int Wo, int k, int st, int pad, int dil, int Tp, float* __restrict__ out)
{
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)B * C * Hf * Wf;
    if (idx >= total) return;
    const int fx = (int)(idx % Wf), fy = (int)((idx / Wf) % Hf), c = (int)((idx / ((size_t)Wf * Hf)) % C), b = (int)(idx / ((size_t)Wf * Hf * C));
    float acc = 0.0f;
    for (int r = 0; r < k; ++r) {
        const int ny = fy + pad - r * dil;
        if (ny < 0 || ny % st) continue;
        const int oy = ny / st;
        if (oy >= Ho) continue;
        for (int q = 0; q < k; ++q) {
            const int nx = fx + pad - q * dil;
            if (nx < 0 || nx % st) continue;
            const int ox = nx / st;
            if (ox >= Wo) continue;
            const size_t off = (size_t)(c * k * k + r * k + q) * Tp + ((size_t)b * Ho + oy) * Wo + ox;
            for (int sl = 0; sl < nslab; ++sl) acc += M[(size_t)sl * slab_stride + off];
        }
    }
    out[idx] = acc;
}

// out[b][r][o] = sum_{slabs} Y[slab][r][(b,o)]
__global__ void __launch_bounds__(256) sm_sum_to_nchw_kernel(const float* __restrict__ Y, int nslab, size_t slab_stride, int B, int R, int HW, int Tp,
                                                             float* __restrict__ out)
{
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)B * R * HW) return;
    const int o = (int)(idx % HW), r = (int)((idx / HW) % R), b = (int)(idx / ((size_t)HW * R));
    const size_t off = (size_t)r * Tp + (size_t)b * HW + o;
    float acc = 0.0f;
    for (int sl = 0; sl < nslab; ++sl) acc += Y[(size_t)sl * slab_stride + off];
    out[idx] = acc;
}

// In-workgroup reduction of the four waves' partial tiles (each wave took a quarter of the slab's reduction range): waves 1..3
// park a tile in LDS, wave 0 adds them in order.  Deterministic; 4x fewer slabs for the same number of waves in flight.
template <int NB>
__device__ __forceinline__ void sm_reduce4(f32x16 (&acc)[NB], float* lds, int wave, int lane)
{
    if (wave > 0) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) lds[(((wave - 1) * NB + nb) * 16 + e) * 64 + lane] = acc[nb][e];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[nb][e] += lds[((w * NB + nb) * 16 + e) * 64 + lane];
    }
    __syncthreads();
}

// DATA: one workgroup (4 waves) = 128 columns q of Wm x the rows of its slab (a quarter per wave) x NB position blocks.
//   A (MFMA j, lane (m, h)) = Wm[r + h][q0 + 4m + j]   — component j of the lane's float4 at Wm[r + h][q0 + 4m]
//   B (lane (n, h))         = in[r + h][n0 + 32 nb + n]
// slab s = blockIdx.y: Mcol_s[q0 + 4*row + j][n] for the 32x32 tile rows `row` of MFMA j.
template <int NB>
__global__ void __launch_bounds__(256) sm_data_kernel(const float* __restrict__ Wm, const float* __restrict__ In, int R, int Q, int Tp, int rows_per_wave,
                                                      float* __restrict__ Mcol)
{
    __shared__ float red[3 * NB * 16 * 64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), m = lane & 31, h = lane >> 5;
    const int q0 = blockIdx.x * 128, n0 = blockIdx.z * (32 * NB);
    const int ra = min(R, (blockIdx.y * 4 + wave) * rows_per_wave), rb = min(R, ra + rows_per_wave);
    f32x16 acc[4][NB];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][nb][e] = 0.0f;
    const float* wp = Wm + (size_t)(ra + h) * Q + q0 + 4 * m;
    const float* ip = In + (size_t)(ra + h) * Tp + n0 + m;
    constexpr int U = NB == 1 ? 8 : (NB == 2 ? 4 : 2);     // row pairs in flight (8 KB of the weight stream per wave at NB = 1); bounded by the 4*NB accumulator tiles
    for (int r = ra; r < rb; r += 2 * U) {
        f32x4v a[U];
        float bv[U][NB];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = r + 2 * u < rb;               // R and rows_per_wave are even: a pair is in or out as a whole
            a[u] = ok ? *reinterpret_cast<const f32x4v*>(wp + (size_t)2 * u * Q) : f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bv[u][nb] = ok ? ip[(size_t)2 * u * Tp + 32 * nb] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[j][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][j], bv[u][nb], acc[j][nb], 0, 0, 0);
        wp += (size_t)2 * U * Q;
        ip += (size_t)2 * U * Tp;
    }
    float* out = Mcol + (size_t)blockIdx.y * Q * Tp;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        sm_reduce4<NB>(acc[j], red, wave, lane);
        if (wave == 0) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                    out[(size_t)(q0 + 4 * row + j) * Tp + n0 + 32 * nb + m] = acc[j][nb][e];
                }
        }
    }
}

// FWD: one workgroup (4 waves) = 32 rows r of Wm x the columns of its slab (a quarter per wave) x NB position blocks.
//   A (MFMA j, lane (m, h)) = Wm[r0 + m][q + 4h + j]   — component j of the lane's float4 at Wm[r0 + m][q + 4h]
//   B (MFMA j, lane (n, h)) = col[q + 4h + j][n0 + 32 nb + n]
template <int NB>
__global__ void __launch_bounds__(256) sm_fwd_kernel(const float* __restrict__ Wm, const float* __restrict__ Xc, int R, int Q, int Tp, int cols_per_wave,
                                                     float* __restrict__ Y)
{
    __shared__ float red[3 * NB * 16 * 64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), m = lane & 31, h = lane >> 5;
    const int r0 = blockIdx.x * 32, n0 = blockIdx.z * (32 * NB);
    const int qa = min(Q, (blockIdx.y * 4 + wave) * cols_per_wave), qb = min(Q, qa + cols_per_wave);
    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nb][e] = 0.0f;
    const float* wp = Wm + (size_t)(r0 + m) * Q + qa + 4 * h;
    const float* xp = Xc + (size_t)(qa + 4 * h) * Tp + n0 + m;
    constexpr int U = 4;                                  // groups of 8 columns in flight
    for (int q = qa; q < qb; q += 8 * U) {
        f32x4v a[U];
        float bv[U][4][NB];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = q + 8 * u < qb;               // Q and cols_per_wave are multiples of 8
            a[u] = ok ? *reinterpret_cast<const f32x4v*>(wp + 8 * u) : f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) bv[u][j][nb] = ok ? xp[(size_t)(8 * u + j) * Tp + 32 * nb] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][j], bv[u][j][nb], acc[nb], 0, 0, 0);
        wp += 8 * U;
        xp += (size_t)8 * U * Tp;
    }
    sm_reduce4<NB>(acc, red, wave, lane);
    if (wave == 0) {
        float* out = Y + (size_t)blockIdx.y * R * Tp;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                out[(size_t)(r0 + row) * Tp + n0 + 32 * nb + m] = acc[nb][e];
            }
    }
}

// WRW: one wave = the 32 x 128 block dW[r0 .. r0+31][q0 .. q0+127], reduction over all Pp (even, zero padded) positions.
//   A (lane (m, h)) = Ct[p + h][r0 + m]                      coarse tensor as [p][r]
//   B (MFMA j)      = component j of the float4 Vt[p + h][q0 + 4n]   im2col of the fine tensor as [p][(c,t)]  -> tile column n is q0 + 4n + j
__global__ void __launch_bounds__(64) sm_wrw_kernel(const float* __restrict__ Ct, const float* __restrict__ Vt, int R, int Q, int Pp, float* __restrict__ dW)
{
    const int lane = threadIdx.x, m = lane & 31, h = lane >> 5;
    const int q0 = blockIdx.x * 128, r0 = blockIdx.y * 32;
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;
    const float* cp = Ct + (size_t)h * R + r0 + m;
    const float* vp = Vt + (size_t)h * Q + q0 + 4 * m;
    constexpr int U = 4;
    for (int p = 0; p < Pp; p += 2 * U) {
        float av[U];
        f32x4v b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = p + 2 * u < Pp;
            av[u] = ok ? cp[(size_t)2 * u * R] : 0.0f;
            b[u] = ok ? *reinterpret_cast<const f32x4v*>(vp + (size_t)2 * u * Q) : f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], b[u][j], acc[j], 0, 0, 0);
        cp += (size_t)2 * U * R;
        vp += (size_t)2 * U * Q;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
        *reinterpret_cast<f32x4v*>(dW + (size_t)(r0 + row) * Q + q0 + 4 * m) = f32x4v{acc[0][e], acc[1][e], acc[2][e], acc[3][e]};
    }
}

struct SmPlan { int P, Tp, Pp, Q, nb, ngroups, nslab, per_slab; size_t a_floats, b_floats, m_floats, total_bytes; };

// op 0 (DATA): in [B][R][Ho][Wo], W [R][Cq][k][k] -> out [B][Cq][Hf][Wf].   op 1 (WRW): coarse [B][R][Ho][Wo], fine [B][Cq][Hf][Wf]
// -> dW [R][Cq][k][k].   op 2 (FWD): fine [B][Cq][Hf][Wf], W -> out [B][R][Ho][Wo].   (R = the weight's first channel dimension.)
static int sm_plan(int op, int B, int R, int Cq, int Ho, int Wo, int Hf, int Wf, int k, int st, int pad, int dil, SmPlan* p)
{
    if (op < 0 || op > 2) return fail(IPSR_ERR_INVALID, "small-map convolution: op %d", op);
    if (k < 1 || k > 4 || st < 1 || st > 2 || dil < 1 || pad < 0) return fail(IPSR_ERR_UNSUPPORTED, "small-map convolution: k%d s%d p%d d%d", k, st, pad, dil);
    if (Ho != (Hf + 2 * pad - dil * (k - 1) - 1) / st + 1 || Wo != (Wf + 2 * pad - dil * (k - 1) - 1) / st + 1)
        return fail(IPSR_ERR_INVALID, "small-map convolution: %dx%d is not the output grid of %dx%d under k%d s%d p%d d%d", Ho, Wo, Hf, Wf, k, st, pad, dil);
    p->P = B * Ho * Wo;
    p->Q = Cq * k * k;
    if (p->Q % 128 != 0) return fail(IPSR_ERR_UNSUPPORTED, "small-map convolution: %d x %d taps is not a multiple of 128", Cq, k * k);
    if (R % 32 != 0) return fail(IPSR_ERR_UNSUPPORTED, "small-map convolution: %d weight rows are not a multiple of 32", R);
    if (p->P > 1024) return fail(IPSR_ERR_UNSUPPORTED, "small-map convolution: %d positions (made for <= 1024)", p->P);
    const int blocks = (p->P + 31) / 32;                           // 32-wide position blocks
    p->nb = blocks >= 4 ? 4 : (blocks == 3 ? 4 : blocks);          // 1, 2 or 4 per wave
    if (op == 0 && p->nb == 4) p->nb = 2;                          // DATA keeps 4 accumulator tiles per position block: 2 blocks fill the registers
    p->ngroups = (blocks + p->nb - 1) / p->nb;
    p->Tp = p->ngroups * p->nb * 32;
    p->Pp = (p->P + 1) & ~1;
    p->a_floats = p->b_floats = p->m_floats = 0;
    p->nslab = 1; p->per_slab = 0;
    // waves in flight ~ 1024 (one per SIMD) when the slab count allows it: a slab costs its write + its read in the post-pass, so it is
    // capped where that traffic would pass half the weight stream
    const size_t w_bytes = (size_t)R * p->Q * 4;
    if (op == 0) {
        const int qb = p->Q / 128;
        const size_t slab_bytes = (size_t)p->Q * p->Tp * 4;
        const int cap = (int)std::max<size_t>(1, w_bytes / slab_bytes);
        int ns = std::max(1, std::min({R / 64, cap, (256 + qb * p->ngroups - 1) / (qb * p->ngroups)}));
        p->per_slab = (((R + 4 * ns - 1) / (4 * ns)) + 1) & ~1;                      // rows per WAVE (even)
        p->nslab = (R + 4 * p->per_slab - 1) / (4 * p->per_slab);
        p->b_floats = (size_t)R * p->Tp;
        p->m_floats = (size_t)p->nslab * p->Q * p->Tp;
    } else if (op == 2) {
        const int rb = R / 32;
        const size_t slab_bytes = (size_t)R * p->Tp * 4;
        const int cap = (int)std::max<size_t>(1, w_bytes / slab_bytes);
        int ns = std::max(1, std::min({p->Q / 256, cap, (256 + rb * p->ngroups - 1) / (rb * p->ngroups)}));
        p->per_slab = (((p->Q + 4 * ns - 1) / (4 * ns)) + 7) & ~7;                   // columns per WAVE (multiple of 8)
        p->nslab = (p->Q + 4 * p->per_slab - 1) / (4 * p->per_slab);
        p->b_floats = (size_t)p->Q * p->Tp;
        p->m_floats = (size_t)p->nslab * R * p->Tp;
    } else {
        p->a_floats = (size_t)p->Pp * R;
        p->b_floats = (size_t)p->Pp * p->Q;
    }
    p->total_bytes = align_up(p->a_floats * 4, 256) + align_up(p->b_floats * 4, 256) + align_up(p->m_floats * 4, 256) + 256;
    return IPSR_OK;
}

size_t smallmap_ws_bytes(int op, int B, int R, int Cq, int Ho, int Wo, int Hf, int Wf, int k, int st, int pad, int dil)
{
    SmPlan p;
    if (sm_plan(op, B, R, Cq, Ho, Wo, Hf, Wf, k, st, pad, dil, &p) != IPSR_OK) return 0;
    return p.total_bytes;
}

int launch_smallmap(int op, const float* a, const float* b2, float* out, int B, int R, int Cq, int Ho, int Wo, int Hf, int Wf,
                    int k, int st_, int pad, int dil, void* ws, size_t ws_bytes, hipStream_t st)
{
    SmPlan p;
    if (int rc = sm_plan(op, B, R, Cq, Ho, Wo, Hf, Wf, k, st_, pad, dil, &p)) return rc;
    if (ws_bytes < p.total_bytes) return fail(IPSR_ERR_WORKSPACE, "small-map convolution: workspace %zu < %zu", ws_bytes, p.total_bytes);
    Carver cv(ws, ws_bytes);
    float* A = cv.take<float>(p.a_floats);
    float* Bv = cv.take<float>(p.b_floats);
    float* Mo = cv.take<float>(p.m_floats);
    if (op == 0) {          // a = in, b2 = W
        sm_to_cn_kernel<<<dim3(cdiv(p.Tp, 256), R), 256, 0, st>>>(a, B, R, Ho * Wo, p.Tp, Bv);
        const dim3 grid(p.Q / 128, p.nslab, p.ngroups);
        if (p.nb == 1) sm_data_kernel<1><<<grid, 256, 0, st>>>(b2, Bv, R, p.Q, p.Tp, p.per_slab, Mo);
        else sm_data_kernel<2><<<grid, 256, 0, st>>>(b2, Bv, R, p.Q, p.Tp, p.per_slab, Mo);
        if (int rc = check_launch("sm_data_kernel")) return rc;
        const size_t total = (size_t)B * Cq * Hf * Wf;
        sm_col2im_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(Mo, p.nslab, (size_t)p.Q * p.Tp, B, Cq, Hf, Wf, Ho, Wo, k, st_, pad, dil, p.Tp, out);
        return check_launch("sm_col2im_kernel");
    }
    if (op == 2) {          // a = fine, b2 = W
        sm_im2col_cn_kernel<<<dim3(cdiv(p.Tp, 256), p.Q), 256, 0, st>>>(a, B, Cq, Hf, Wf, Ho, Wo, k, st_, pad, dil, p.Tp, Bv);
        const dim3 grid(R / 32, p.nslab, p.ngroups);
        if (p.nb == 1) sm_fwd_kernel<1><<<grid, 256, 0, st>>>(b2, Bv, R, p.Q, p.Tp, p.per_slab, Mo);
        else if (p.nb == 2) sm_fwd_kernel<2><<<grid, 256, 0, st>>>(b2, Bv, R, p.Q, p.Tp, p.per_slab, Mo);
        else sm_fwd_kernel<4><<<grid, 256, 0, st>>>(b2, Bv, R, p.Q, p.Tp, p.per_slab, Mo);
        if (int rc = check_launch("sm_fwd_kernel")) return rc;
        const size_t total = (size_t)B * R * Ho * Wo;
        sm_sum_to_nchw_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(Mo, p.nslab, (size_t)R * p.Tp, B, R, Ho * Wo, p.Tp, out);
        return check_launch("sm_sum_to_nchw_kernel");
    }
    // op 1: a = coarse, b2 = fine
    sm_to_nc_kernel<<<dim3(cdiv(R, 256), p.Pp), 256, 0, st>>>(a, B, R, Ho * Wo, A);
    sm_im2col_nt_kernel<<<dim3(cdiv(p.Q, 256), p.Pp), 256, 0, st>>>(b2, B, Cq, Hf, Wf, Ho, Wo, k, st_, pad, dil, Bv);
    sm_wrw_kernel<<<dim3(p.Q / 128, R / 32), 64, 0, st>>>(A, Bv, R, p.Q, p.Pp, out);
    return check_launch("sm_wrw_kernel");
}

}  // namespace ipsr

using namespace ipsr;

extern "C" {

int ipsr_debug_force_wino_split(int nsplit, int xi_split, int nsplit_t)
{
    if (nsplit == 0 && xi_split == 0 && nsplit_t == 0) { g_force_split[0] = g_force_split[1] = g_force_split[2] = 0; return IPSR_OK; }
    if (nsplit < 1 || nsplit_t < 1 || xi_split < 1 || xi_split > 36) return fail(IPSR_ERR_INVALID, "ipsr_debug_force_wino_split: bad split %d,%d,%d", nsplit, xi_split, nsplit_t);
    g_force_split[0] = nsplit; g_force_split[1] = xi_split; g_force_split[2] = nsplit_t;
    return IPSR_OK;
}

int ipsr_wino_gemm_split(int rows, int cols, int reduction, int* out5)
{
    if (!out5 || rows < 1 || cols < 1 || reduction < 1 || rows % WG_BM || cols % WG_BN || reduction % WG_BK)
        return fail(IPSR_ERR_INVALID, "ipsr_wino_gemm_split: rows / cols must be multiples of 128 and the reduction of 16");
    const WinoSplit sp = wino_choose_split(36 * (rows / WG_BM) * (cols / WG_BN), reduction / WG_BK, (size_t)36 * rows * cols * 4);
    out5[0] = sp.nsplit; out5[1] = sp.sps; out5[2] = sp.xi_split; out5[3] = sp.nsplit_t; out5[4] = sp.sps_t;
    return IPSR_OK;
}

size_t ipsr_conv3x3_winograd_workspace_bytes(int op, int B, int Cin, int H, int W, int Cout)
{
    if (op < 0 || op > 3 || B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1) return 0;
    const bool fwd = op == 0 || op == 2;
    return winograd_ws_bytes(B, fwd ? Cin : Cout, fwd ? Cout : Cin, H, W);
}

static int arith_from(const char* who, int math, int io, ConvArith* ar)
{
    if ((math != 0 && math != 2 && math != 3) || io < 0 || io > 3)
        return fail(IPSR_ERR_INVALID, "%s: math %d (0 = fp32 MFMA, 2 = split bf16 x3, 3 = split bf16 x6) / io %d (bit 0: bf16 activations in, bit 1: out)", who, math, io);
    ar->math = math; ar->in_bf16 = (io & 1) != 0; ar->out_bf16 = (io & 2) != 0;
    return IPSR_OK;
}

int ipsr_conv3x3_winograd_mp(int op, const void* in, const float* weight, const float* bias, int epilogue, float* filter_cache,
                             int filter_cache_valid, void* out, int B, int Cin, int H, int W, int Cout, int math, int io,
                             void* ws, size_t ws_bytes, void* stream)
{
    if (!in || !weight || !out || !ws) return fail(IPSR_ERR_INVALID, "ipsr_conv3x3_winograd: null pointer");
    if (op < 0 || op > 3 || B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1) return fail(IPSR_ERR_INVALID, "ipsr_conv3x3_winograd: bad argument");
    if ((reinterpret_cast<uintptr_t>(ws) & 15u) || (reinterpret_cast<uintptr_t>(out) & 15u) || (reinterpret_cast<uintptr_t>(filter_cache) & 15u))
        return fail(IPSR_ERR_INVALID, "ipsr_conv3x3_winograd: out / workspace / filter_cache must be 16-byte aligned");
    ConvArith ar;
    if (int rc = arith_from("ipsr_conv3x3_winograd_mp", math, io, &ar)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // k3 s1 p1: input and output have the same extent.  C = reduction channels, K = produced channels.
    switch (op) {
        case 0:      // Conv2d forward: weight [Cout][Cin][3][3]: (c, k) at c*9 + k*Cin*9
            return launch_winograd(in, weight, out, B, Cin, Cout, H, W, 9, (long)Cin * 9, 0, ws, ws_bytes, st, bias, epilogue, filter_cache, filter_cache_valid, ar);
        case 1:      // Conv2d backward-data: dx = conv(dy, flipped w), reduction over Cout: (c=co, k=ci) at co*Cin*9 + ci*9
            return launch_winograd(in, weight, out, B, Cout, Cin, H, W, (long)Cin * 9, 9, 1, ws, ws_bytes, st, bias, epilogue, filter_cache, filter_cache_valid, ar);
        case 2:      // ConvTranspose2d forward: weight [Cin][Cout][3][3], flipped: (c=ci, k=co) at ci*Cout*9 + co*9
            return launch_winograd(in, weight, out, B, Cin, Cout, H, W, (long)Cout * 9, 9, 1, ws, ws_bytes, st, bias, epilogue, filter_cache, filter_cache_valid, ar);
        default:     // ConvTranspose2d backward-data: dx = conv(dy, w as [ci][co]), reduction over Cout: (c=co, k=ci) at co*9 + ci*Cout*9
            return launch_winograd(in, weight, out, B, Cout, Cin, H, W, 9, (long)Cout * 9, 0, ws, ws_bytes, st, bias, epilogue, filter_cache, filter_cache_valid, ar);
    }
}

int ipsr_conv3x3_winograd(int op, const float* in, const float* weight, float* out, int B, int Cin, int H, int W, int Cout,
                          void* ws, size_t ws_bytes, void* stream)
{
    return ipsr_conv3x3_winograd_mp(op, in, weight, nullptr, 0, nullptr, 0, out, B, Cin, H, W, Cout, 0, 0, ws, ws_bytes, stream);
}

size_t ipsr_conv3x3_winograd_filter_floats(int op, int Cin, int Cout)
{
    if (op < 0 || op > 3 || Cin < 1 || Cout < 1) return 0;
    const bool fwd = op == 0 || op == 2;
    return winograd_filter_floats(fwd ? Cin : Cout, fwd ? Cout : Cin);
}

int ipsr_conv3x3_winograd_ex(int op, const float* in, const float* weight, const float* bias, int epilogue, float* filter_cache,
                             int filter_cache_valid, float* out, int B, int Cin, int H, int W, int Cout,
                             void* ws, size_t ws_bytes, void* stream)
{
    return ipsr_conv3x3_winograd_mp(op, in, weight, bias, epilogue, filter_cache, filter_cache_valid, out, B, Cin, H, W, Cout, 0, 0, ws, ws_bytes, stream);
}

size_t ipsr_conv4x4_winograd_workspace_bytes(int geom, int mode, int B, int Cin, int H, int W, int Cout)
{
    if (geom < 0 || geom > 1 || mode < 0 || mode > 2 || B < 1 || Cin < 1 || Cout < 1 || H < 2 || W < 2) return 0;
    return winograd_dil_ws_bytes(geom, mode, B, Cin, H, W, Cout);
}

int ipsr_conv4x4_winograd_mp(int geom, int mode, const void* a, const void* b, void* out, int B, int Cin, int H, int W, int Cout,
                             int math, int io, void* ws, size_t ws_bytes, void* stream)
{
    if (!a || !b || !out || !ws) return fail(IPSR_ERR_INVALID, "ipsr_conv4x4_winograd: null pointer");
    if (geom < 0 || geom > 1 || mode < 0 || mode > 2 || B < 1 || Cin < 1 || Cout < 1 || H < 2 || W < 2)
        return fail(IPSR_ERR_INVALID, "ipsr_conv4x4_winograd: bad argument");
    if ((reinterpret_cast<uintptr_t>(ws) & 15u) || (reinterpret_cast<uintptr_t>(out) & 15u))
        return fail(IPSR_ERR_INVALID, "ipsr_conv4x4_winograd: out / workspace must be 16-byte aligned");
    ConvArith ar;
    if (int rc = arith_from("ipsr_conv4x4_winograd_mp", math, io, &ar)) return rc;
    return launch_winograd_dil(geom, mode, a, b, out, B, Cin, H, W, Cout, ws, ws_bytes, static_cast<hipStream_t>(stream), ar);
}

int ipsr_conv4x4_winograd(int geom, int mode, const float* a, const float* b, float* out, int B, int Cin, int H, int W, int Cout,
                          void* ws, size_t ws_bytes, void* stream)
{
    return ipsr_conv4x4_winograd_mp(geom, mode, a, b, out, B, Cin, H, W, Cout, 0, 0, ws, ws_bytes, stream);
}

size_t ipsr_conv4x4_dilated_winograd_workspace_bytes(int mode, int B, int Cin, int H, int W, int Cout)
{
    return ipsr_conv4x4_winograd_workspace_bytes(0, mode, B, Cin, H, W, Cout);
}

int ipsr_conv4x4_dilated_winograd(int mode, const float* a, const float* b, float* out, int B, int Cin, int H, int W, int Cout,
                                  void* ws, size_t ws_bytes, void* stream)
{
    return ipsr_conv4x4_winograd(0, mode, a, b, out, B, Cin, H, W, Cout, ws, ws_bytes, stream);
}

size_t ipsr_conv4x4s2_winograd_workspace_bytes(int mode, int B, int Kc, int Cf, int nh, int nw)
{
    if (mode < 0 || mode > 2 || B < 1 || Kc < 1 || Cf < 1 || nh < 1 || nw < 1) return 0;
    return winograd_s2_ws_bytes(mode, B, Kc, Cf, nh, nw);
}

int ipsr_conv4x4s2_winograd_mp(int mode, const void* a, const void* b, void* out, int B, int Kc, int Cf, int nh, int nw,
                               int math, int io, void* ws, size_t ws_bytes, void* stream)
{
    if (!a || !b || !out || !ws) return fail(IPSR_ERR_INVALID, "ipsr_conv4x4s2_winograd: null pointer");
    if (mode < 0 || mode > 2 || B < 1 || Kc < 1 || Cf < 1 || nh < 1 || nw < 1) return fail(IPSR_ERR_INVALID, "ipsr_conv4x4s2_winograd: bad argument");
    if ((reinterpret_cast<uintptr_t>(ws) & 15u) || (reinterpret_cast<uintptr_t>(out) & 15u))
        return fail(IPSR_ERR_INVALID, "ipsr_conv4x4s2_winograd: out / workspace must be 16-byte aligned");
    ConvArith ar;
    if (int rc = arith_from("ipsr_conv4x4s2_winograd_mp", math, io, &ar)) return rc;
    return launch_winograd_s2(mode, a, b, out, B, Kc, Cf, nh, nw, ws, ws_bytes, static_cast<hipStream_t>(stream), ar);
}

int ipsr_conv4x4s2_winograd(int mode, const float* a, const float* b, float* out, int B, int Kc, int Cf, int nh, int nw,
                            void* ws, size_t ws_bytes, void* stream)
{
    return ipsr_conv4x4s2_winograd_mp(mode, a, b, out, B, Kc, Cf, nh, nw, 0, 0, ws, ws_bytes, stream);
}

size_t ipsr_conv_smallmap_workspace_bytes(int op, int B, int R, int Cq, int Ho, int Wo, int Hf, int Wf, int k, int stride, int pad, int dil)
{
    if (B < 1 || R < 1 || Cq < 1 || Ho < 1 || Wo < 1 || Hf < 1 || Wf < 1) return 0;
    return smallmap_ws_bytes(op, B, R, Cq, Ho, Wo, Hf, Wf, k, stride, pad, dil);
}

int ipsr_conv_smallmap(int op, const float* a, const float* b, float* out, int B, int R, int Cq, int Ho, int Wo, int Hf, int Wf,
                       int k, int stride, int pad, int dil, void* ws, size_t ws_bytes, void* stream)
{
    if (!a || !b || !out || !ws) return fail(IPSR_ERR_INVALID, "ipsr_conv_smallmap: null pointer");
    if (B < 1 || R < 1 || Cq < 1 || Ho < 1 || Wo < 1 || Hf < 1 || Wf < 1) return fail(IPSR_ERR_INVALID, "ipsr_conv_smallmap: bad argument");
    if ((reinterpret_cast<uintptr_t>(ws) & 15u) || (reinterpret_cast<uintptr_t>(out) & 15u) || (reinterpret_cast<uintptr_t>(a) & 15u) ||
        (reinterpret_cast<uintptr_t>(b) & 15u))
        return fail(IPSR_ERR_INVALID, "ipsr_conv_smallmap: operands / workspace must be 16-byte aligned");
    return launch_smallmap(op, a, b, out, B, R, Cq, Ho, Wo, Hf, Wf, k, stride, pad, dil, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

size_t ipsr_conv3x3_winograd_wrw_workspace_bytes(int transposed, int B, int Cin, int H, int W, int Cout)
{
    if (B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1) return 0;
    return transposed ? winograd_wrw_ws_bytes(B, Cin, Cout, H, W) : winograd_wrw_ws_bytes(B, Cout, Cin, H, W);
}

int ipsr_conv3x3_winograd_wrw_mp(int transposed, const void* x, const void* dy, float* dw, int B, int Cin, int H, int W, int Cout,
                                 int math, int io, void* ws, size_t ws_bytes, void* stream)
{
    if (!x || !dy || !dw || !ws) return fail(IPSR_ERR_INVALID, "ipsr_conv3x3_winograd_wrw: null pointer");
    if (B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1) return fail(IPSR_ERR_INVALID, "ipsr_conv3x3_winograd_wrw: bad argument");
    if (reinterpret_cast<uintptr_t>(ws) & 15u) return fail(IPSR_ERR_INVALID, "ipsr_conv3x3_winograd_wrw: workspace must be 16-byte aligned");
    ConvArith ar;
    if (int rc = arith_from("ipsr_conv3x3_winograd_wrw_mp", math, io, &ar)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // Conv2d:          dW[co][ci][r][s] = sum dy[co][o] x[ci][o + r - 1]       tile operand dy, window operand x
    // ConvTranspose2d: dW[ci][co][r][s] = sum x[ci][i] dy[co][i + r - 1]       tile operand x,  window operand dy
    if (transposed) return launch_winograd_wrw(x, dy, dw, B, Cin, Cout, H, W, ws, ws_bytes, st, ar);
    return launch_winograd_wrw(dy, x, dw, B, Cout, Cin, H, W, ws, ws_bytes, st, ar);
}

int ipsr_conv3x3_winograd_wrw(int transposed, const float* x, const float* dy, float* dw, int B, int Cin, int H, int W, int Cout,
                              void* ws, size_t ws_bytes, void* stream)
{
    return ipsr_conv3x3_winograd_wrw_mp(transposed, x, dy, dw, B, Cin, H, W, Cout, 0, 0, ws, ws_bytes, stream);
}

}  // extern "C"
