// conv_gemm.hip — SURVEY §8 (f)1: the convolutions / transposed convolutions of the U-Nets, the PatchGAN discriminators and
// the VGG16 feature net as ONE-launch implicit GEMMs on the fp32 matrix cores, NCHW in and out (no layout shuffles).
//
// Reference geometry (models/networks.py:220-259, 404-432, 470-495, 510-515; models/vgg16.py:9-21): k3 s1 p1, k4 s2 p1,
// k4 s2 p3 d2 (dilated), k4 s1 p1, ConvTranspose k3 s1 p1, ConvTranspose k4 s2 p1 — forward and both gradients.
//
// Every one of those is the same contraction
//      D[m][n] = sum_{ci, t}  Wp[(ci, t)][m] * X[b][ci][ o_y'*in_mul + dy[t] ][ o_x'*in_mul + dx[t] ],      n = (b, o_y', o_x')
// with a per-launch TAP TABLE (dy[t], dx[t]) and an affine map from the tile grid (o_y', o_x') to the output tensor:
//   conv forward / convT backward-data : in = out*stride - pad + r*dil                     (one launch, NT = k*k taps)
//   conv backward-data / convT forward : out' = (in + pad - r*dil) / stride where that divides.  stride 1: one launch with
//       the offsets pad - r*dil.  stride 2: the output splits into 4 parity classes (y&1, x&1); each class sees a fixed
//       subset of the taps (2x2 of the 4x4 for k4 p1; all 16 for the odd/odd class of the dilated conv, none elsewhere),
//       i.e. a dense stride-1 gather again — one launch per non-empty class, written with stride 2 into the output.
// So ONE kernel template (NT = 4 | 9 | 16 taps per reduction channel) serves all of them.  The weight-layout differences
// (Conv2d [Cout][Cin][k][k] vs ConvTranspose2d [Cin][Cout][k][k], forward vs transposed use, tap subsets) are absorbed by a
// small re-pack kernel that writes Wp[(ci, t)][m] — reduction-major with m contiguous, which is exactly the MFMA A-operand
// LDS image — so the weight tiles stream HBM/L2 -> LDS by direct DMA like the correlation kernel's operands.
//
// Pipeline (same skeleton as corr_argmax_fast_kernel, measured there at 73-78 % of the fp32 MFMA peak):
//   workgroup 256 threads = 4 waves, tile 128 (m) x 128 (pixels), wave owns 32x32 sub-tiles {wm*32, +64} x {wn*32, +64};
//   4-slot LDS ring of stages of BK reduction rows (BK = 16: one channel x 16 taps / four channels x 4 taps; 18: two
//   channels x 9 taps); A rows by global_load_lds_dwordx4 three stages ahead; B rows (the im2col gather) through
//   registers one stage ahead: thread = one pixel column, BK/2 rows, validity of every tap of its pixel precomputed as a
//   bit mask, so a gathered element costs one predicated dword load; counted vmcnt + raw s_barrier per stage.
//   Split-K over the reduction (small feature maps have too few tiles to fill 256 CUs): partial tiles to the workspace,
//   summed in a fixed order by conv_reduce_kernel (deterministic, no atomics).
#include "ipsr_common.h"

namespace ipsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int CG_BM = 128, CG_BN = 128, CG_NBUF = 4, CG_THREADS = 256;

struct ConvLaunch {
    const float* x;       // activations read by the gather  [B][Cred][Hin][Win]
    const float* wp;      // packed weights                  [nstage*BK][Mp]
    float* y;             // output tensor                   [B][M][Hout][Wout]
    float* part;          // split-K partials [ksplit][M][Ntot] (ksplit > 1) else unused
    int B, Cred, Hin, Win, M, Mp;
    int Ho, Wo;           // tile grid per sample
    int Hout, Wout;
    int oy_mul, oy_add, ox_mul, ox_add;
    int in_mul;
    int nstage;           // total reduction stages
    int ksplit, stages_per_split;
    int m_tiles, n_tiles;
    // tap t = a*NSX + b (NSX = sqrt(NT) taps per kernel row):  dy = dy0 + a*dys,  dx = dx0 + b*dxs.  Scalars, not tables:
    // an indexed array inside the kernel-argument struct made the compiler fetch it per lane with waterfall loops.
    int dy0, dys, dx0, dxs;
    int minoff;           // min_t(dy*Win + dx)  (<= 0): the gather's buffer base is x + minoff
};

template <int NT> struct CgTap {
    static constexpr int NSX = NT == 4 ? 2 : (NT == 9 ? 3 : 4);
    static __device__ __forceinline__ int dy(const ConvLaunch& p, int t) { return p.dy0 + (t / NSX) * p.dys; }
    static __device__ __forceinline__ int dx(const ConvLaunch& p, int t) { return p.dx0 + (t % NSX) * p.dxs; }
};

template <int NT> struct CgCfg {
    static constexpr int BK = (NT == 9) ? 18 : 16;
    static constexpr int CPS = BK / NT;                  // reduction channels per stage
    static constexpr int NB = BK / 2;                    // gathered elements per thread per stage
    static constexpr int PAIRS = BK / 2;                 // 1-KiB DMA pieces (2 rows of 128 floats) per stage
    static constexpr int SLOT = 2 * BK * CG_BM;          // floats per ring slot (A then B)
};

// One stage of the im2col gather for the thread's pixel column: the NB rows  jj + NB*gpar  (gpar = 0 | 1, wave-uniform)
// -> registers.  Row -> (channel within the stage, tap):
//      NT = 16 (BK 16, 1 channel):  cc = 0,             t = jj + 8*gpar
//      NT =  9 (BK 18, 2 channels): cc = gpar,          t = jj
//      NT =  4 (BK 16, 4 channels): cc = jj/4 + 2*gpar, t = jj % 4
// Branch-free: raw buffer loads whose per-lane offset is the pixel's byte offset, or an offset with bit 31 set where the
// tap falls outside the input (hardware range check against num_records = 2^31 -> the load returns 0).  The channel /
// tap part of the address is wave-uniform and rides in the scalar offset, so an element costs two VALU operations and
// straight-line code (a predicated load per element, or one code path per gpar, made the compiler drain vmcnt — and with
// it the weight DMA in flight — in the middle of every stage).
template <int NT>
__device__ __forceinline__ void cg_gather(float (&reg)[CgCfg<NT>::NB], __amdgpu_buffer_rsrc_t rsrc, const ConvLaunch& p,
                                          unsigned pixoff, unsigned invm, int s, int hw_in, int gpar)
{
    constexpr int NB = CgCfg<NT>::NB, CPS = CgCfg<NT>::CPS, NSX = CgTap<NT>::NSX;
#pragma unroll
    for (int jj = 0; jj < NB; ++jj) {
        int cc, a, b, tl;                                // a, b: tap row / column; tl: bit of the (pre-shifted) mask
        if (NT == 16) { cc = 0; a = jj / NSX + 2 * gpar; b = jj % NSX; tl = jj; }
        else if (NT == 9) { cc = gpar; a = jj / NSX; b = jj % NSX; tl = jj; }
        else { cc = jj / 4 + 2 * gpar; a = (jj % 4) / NSX; b = jj % NSX; tl = jj % 4; }
        const int ci = s * CPS + cc;                     // uniform
        const unsigned voff = ((invm << (31 - tl)) & 0x80000000u) | pixoff;
        const int soff = (ci * hw_in + (p.dy0 + a * p.dys) * p.Win + p.dx0 + b * p.dxs - p.minoff) * 4;   // uniform, >= 0
        reg[jj] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0));
    }
}

template <int NT>
__global__ void __launch_bounds__(CG_THREADS, 2) conv_gemm_kernel(const ConvLaunch p)
{
    using Cfg = CgCfg<NT>;
    constexpr int BK = Cfg::BK, NB = Cfg::NB, PAIRS = Cfg::PAIRS, SLOT = Cfg::SLOT;
    __shared__ __attribute__((aligned(16))) float lds[CG_NBUF * SLOT];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // in an SGPR: branches on it are scalar
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    // logical tile: m-tiles fastest (they share the gathered activations), then the k-splits, then the pixel tiles
    const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = L % p.m_tiles;
    const int ks = (L / p.m_tiles) % p.ksplit;
    const int nt = L / (p.m_tiles * p.ksplit);
    const int m0 = mt * CG_BM, n0 = nt * CG_BN;
    const int hw_o = p.Ho * p.Wo, hw_in = p.Hin * p.Win;
    const int ntot = p.B * hw_o;

    // ---- gather role: this thread's pixel column and the validity of each tap there
    const int gpar = wave >> 1;                          // this half of the block gathers rows jj + NB*gpar (wave-uniform)
    const int gn = n0 + (tid & 127);
    unsigned inv_mask = 0;                               // bit t set = tap t of this pixel reads outside the input
    unsigned pixoff = 0;                                 // byte offset of (b, channel 0, iy0, ix0) from the tensor start
    {
        const bool nv = gn < ntot;
        const int gnc = nv ? gn : 0;
        const int b = gnc / hw_o, rem = gnc - b * hw_o;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        const int iy0 = oy * p.in_mul, ix0 = ox * p.in_mul;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int iy = iy0 + CgTap<NT>::dy(p, t), ix = ix0 + CgTap<NT>::dx(p, t);
            if (!(nv && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win)) inv_mask |= 1u << t;
        }
        pixoff = (unsigned)((b * p.Cred * hw_in + iy0 * p.Win + ix0) * 4);      // host checks the tensor is < 2 GiB
    }
    const unsigned invm = NT == 16 ? inv_mask >> (8 * gpar) : inv_mask;      // NT = 16: this half's taps are 8*gpar .. 8*gpar+7
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + p.minoff), 0, 0x80000000, 0x00020000);

    const int s_lo = ks * p.stages_per_split;
    const int ns = min(p.nstage, s_lo + p.stages_per_split) - s_lo;             // stages of this workgroup (>= 1)

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // ---- A operand: DMA pieces.  Stage rows = 2*PAIRS; pair q (rows 2q, 2q+1) belongs to wave q % 4.
    const int dma_row = lane >> 5, dma_col = (lane & 31) * 4;
    const float* wbase = p.wp + (size_t)s_lo * BK * p.Mp + m0 + dma_col;
    auto dma_piece = [&](int s, int q) {                 // s: stage local to this workgroup, q: pair index (uniform)
        const int slot = (s & (CG_NBUF - 1)) * SLOT + q * 2 * CG_BM;
        const float* g = wbase + ((size_t)s * BK + 2 * q + dma_row) * p.Mp;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)&lds[slot], 16, 0, 0);
    };
    constexpr int NPW = (PAIRS + 3) / 4;                 // pieces per wave per stage (the last one only for some waves)
    auto dma_stage = [&](int s) {
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int q = wave + 4 * i;
            if (q < PAIRS) dma_piece(s, q);
        }
    };

    float breg[NB];
    auto gather = [&](int s) {
        cg_gather<NT>(breg, rsrc, p, pixoff, invm, s_lo + s, hw_in, gpar);
    };
    auto store_b = [&](int s) {
        float* bt = lds + (s & (CG_NBUF - 1)) * SLOT + BK * CG_BM + (tid & 127);
#pragma unroll
        for (int jj = 0; jj < NB; ++jj) bt[(jj + NB * gpar) * CG_BM] = breg[jj];
    };

    // ---- prologue: A stages 0..2 in flight, B stage 0 in LDS
    dma_stage(0);
    if (ns > 1) dma_stage(1);
    if (ns > 2) dma_stage(2);
    gather(0);
    store_b(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    for (int s = 0; s < ns; ++s) {
        const int cur = s & (CG_NBUF - 1);
        const bool more = s + 1 < ns;
        if (more) gather(s + 1);                          // lands under this stage's MFMAs
        const bool prefetch = s + 3 < ns;
        const float* ta = lds + cur * SLOT + h * CG_BM + wm * 32 + r;
        const float* tb = lds + cur * SLOT + BK * CG_BM + h * CG_BM + wn * 32 + r;
        float fa0[3], fa1[3], fb0[3], fb1[3];
        fa0[0] = ta[0]; fa1[0] = ta[64]; fb0[0] = tb[0]; fb1[0] = tb[64];
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int cs = kk % 3, nx = (kk + 1) % 3;
            if (kk + 1 < BK / 2) {
                const int ro = (kk + 1) * 2 * CG_BM;
                fa0[nx] = ta[ro]; fa1[nx] = ta[ro + 64]; fb0[nx] = tb[ro]; fb1[nx] = tb[ro + 64];
            }
            // this wave's DMA pieces of stage s+3, spread over the k-steps (slot (s+3)%4 was last read in stage s-1)
            if (prefetch && kk < NPW) {
                const int q = wave + 4 * kk;
                if (q < PAIRS) dma_piece(s + 3, q);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[cs], fb0[cs], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[cs], fb1[cs], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[cs], fb0[cs], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[cs], fb1[cs], acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // Stage s+1 must be complete in LDS before anyone crosses the barrier: this thread's gathered registers (issued at the
        // top of this stage) and the A pieces in flight.  The pieces of stage s+3 were issued in the first k-steps above, a
        // whole stage of MFMAs ago, so draining them here costs nothing measurable and keeps the wait a plain vmcnt(0).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (more) store_b(s + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue: 2x2 sub-tiles of 32x32; lane = pixel column, 16 m rows per sub-tile in registers
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
        const int n = n0 + wn * 32 + jn * 64 + r;
        if (n >= ntot) continue;
        if (p.ksplit > 1) {
            float* dst = p.part + (size_t)ks * p.M * ntot + n;
#pragma unroll
            for (int im = 0; im < 2; ++im)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + wm * 32 + im * 64 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (m < p.M) dst[(size_t)m * ntot] = acc[im][jn][e];
                }
        } else {
            const int b = n / hw_o, rem = n - b * hw_o;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            const size_t plane = (size_t)p.Hout * p.Wout;
            float* dst = p.y + (size_t)b * p.M * plane + (size_t)(oy * p.oy_mul + p.oy_add) * p.Wout + ox * p.ox_mul + p.ox_add;
#pragma unroll
            for (int im = 0; im < 2; ++im)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + wm * 32 + im * 64 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (m < p.M) dst[(size_t)m * plane] = acc[im][jn][e];
                }
        }
    }
}

// split-K: y = sum over the k-splits (ascending), scattered to the output tensor
__global__ void __launch_bounds__(256) conv_reduce_kernel(const float* __restrict__ part, int ksplit, int M, int ntot, int Ho, int Wo,
                                                          int Hout, int Wout, int oy_mul, int oy_add, int ox_mul, int ox_add,
                                                          float* __restrict__ y)
{
    const int n = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y;
    if (n >= ntot) return;
    float acc = part[(size_t)m * ntot + n];
    for (int k = 1; k < ksplit; ++k) acc += part[((size_t)k * M + m) * ntot + n];
    const int hw_o = Ho * Wo;
    const int b = n / hw_o, rem = n - b * hw_o;
    const int oy = rem / Wo, ox = rem - oy * Wo;
    y[(((size_t)b * M + m) * Hout + oy * oy_mul + oy_add) * Wout + ox * ox_mul + ox_add] = acc;
}

// Wp[(ci, t)][m] = W[ci*sc + m*sm + src[t]]   (zero for ci >= Cred, m >= M); 32x32 tiles transposed through LDS so that
// both the reads (along the source's contiguous taps) and the writes (along m) are coalesced.
struct TapSrc { int nsx, r0, rs, s0, ss, k; };       // source tap of t = a*nsx + b:  (r0 + a*rs)*k + (s0 + b*ss)

__global__ void __launch_bounds__(256) conv_pack_weights_kernel(const float* __restrict__ W, float* __restrict__ Wp, int M, int Mp,
                                                                int Cred, int red_rows, int NT, long sc, long sm, TapSrc taps)
{
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    {
        const int red = r0 + tx;
        const int ci = red / NT, t = red - ci * NT;
        const bool rok = red < red_rows && ci < Cred;
        const int ta = t / taps.nsx, tb = t - ta * taps.nsx;
        const long roff = (long)ci * sc + (taps.r0 + ta * taps.rs) * taps.k + (taps.s0 + tb * taps.ss);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + ty + 8 * i;
            tile[ty + 8 * i][tx] = (rok && m < M) ? W[roff + (long)m * sm] : 0.0f;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int red = r0 + ty + 8 * i, m = m0 + tx;
        if (red < red_rows && m < Mp) Wp[(size_t)red * Mp + m] = tile[tx][ty + 8 * i];
    }
}

// ---------------------------------------------------------------------------------------------------
// host side
struct ConvPlan {
    int nclass;                 // launches
    int NT[4];                  // taps per class (0 = empty class: that part of the output is zero)
    int BK[4], nstage[4];
    int Ho[4], Wo[4];           // class grids
    int py[4], px[4];
    int ksplit[4], sps[4];
    int Mp;
    bool need_zero;             // some output positions are covered by no class
    size_t wp_off[4], wp_floats[4];
    size_t part_floats;
    size_t total_bytes;
};

static int nt_supported(int nt) { return nt == 4 || nt == 9 || nt == 16; }

// taps of one axis that reach output parity `par` for the transposed form: (par + pad - r*dil) % stride == 0
static int axis_taps(int k, int stride, int pad, int dil, int par, int* rr, int* off)
{
    int n = 0;
    for (int r = 0; r < k; ++r) {
        const int v = par + pad - r * dil;
        if (((v % stride) + stride) % stride == 0) { rr[n] = r; off[n] = (v - (((v % stride) + stride) % stride)) / stride; ++n; }
    }
    return n;
}

static void choose_split(int tiles, int nstage, int* ksplit, int* sps)
{
    // fill ~2 workgroups per CU; at least 8 stages per split so that the pipeline prologue stays small
    int ks = 1;
    while (tiles * ks < 512 && nstage / (ks * 2) >= 8 && ks < 64) ks *= 2;
    *sps = cdiv(nstage, ks);
    *ksplit = cdiv(nstage, *sps);
}

// transposed: false = "direct" gather (conv forward, convT backward-data); true = conv backward-data / convT forward.
// Geometry is that of the ORIGINAL convolution: (Hin, Win) here are the extents of the tensor being READ, (Hout, Wout)
// of the tensor being written.
static int make_plan(bool transposed, int B, int Cred, int M, int Hin, int Win, int Hout, int Wout, int k, int stride, int pad, int dil,
                     ConvPlan* pl)
{
    ConvPlan& P = *pl;
    // the gather addresses the input with 32-bit byte offsets (buffer loads, bit 31 = "outside")
    if ((size_t)B * Cred * Hin * Win * 4 + (size_t)(k * dil + pad + 1) * (Win + 1) * 8 >= (1ull << 31))
        return fail(IPSR_ERR_UNSUPPORTED, "conv: input tensor of %zu bytes is too large for the 32-bit gather offsets", (size_t)B * Cred * Hin * Win * 4);
    P.Mp = (M + CG_BM - 1) / CG_BM * CG_BM;
    P.need_zero = false;
    P.part_floats = 0;
    size_t off = 0;
    if (!transposed || stride == 1) {
        P.nclass = 1;
        P.NT[0] = k * k;
        P.Ho[0] = Hout; P.Wo[0] = Wout; P.py[0] = P.px[0] = 0;
    } else {
        if (stride != 2) return fail(IPSR_ERR_UNSUPPORTED, "conv: transposed gather with stride %d", stride);
        P.nclass = 4;
        for (int c = 0; c < 4; ++c) {
            const int py = c >> 1, px = c & 1;
            int rr[4], oo[4];
            const int nr = axis_taps(k, stride, pad, dil, py, rr, oo), nsx = axis_taps(k, stride, pad, dil, px, rr, oo);
            P.NT[c] = nr * nsx;
            P.py[c] = py; P.px[c] = px;
            P.Ho[c] = (Hout - py + 1) / 2; P.Wo[c] = (Wout - px + 1) / 2;
            if (P.NT[c] == 0 && P.Ho[c] > 0 && P.Wo[c] > 0) P.need_zero = true;
        }
    }
    for (int c = 0; c < P.nclass; ++c) {
        P.wp_off[c] = off; P.wp_floats[c] = 0; P.nstage[c] = 0; P.ksplit[c] = 1; P.sps[c] = 0; P.BK[c] = 0;
        if (P.NT[c] == 0 || P.Ho[c] <= 0 || P.Wo[c] <= 0) continue;
        if (!nt_supported(P.NT[c])) return fail(IPSR_ERR_UNSUPPORTED, "conv: %d taps per channel not supported (k=%d stride=%d pad=%d dil=%d)", P.NT[c], k, stride, pad, dil);
        P.BK[c] = P.NT[c] == 9 ? 18 : 16;
        const int cps = P.BK[c] / P.NT[c];
        if (Cred % cps != 0) return fail(IPSR_ERR_UNSUPPORTED, "conv: %d reduction channels are not a multiple of %d", Cred, cps);
        P.nstage[c] = cdiv(Cred, cps);
        P.wp_floats[c] = (size_t)P.nstage[c] * P.BK[c] * P.Mp;
        off += align_up(P.wp_floats[c] * 4, 256) / 4;
        const int tiles = cdiv(B * P.Ho[c] * P.Wo[c], CG_BN) * (P.Mp / CG_BM);
        choose_split(tiles, P.nstage[c], &P.ksplit[c], &P.sps[c]);
        if (P.ksplit[c] > 1) {
            const size_t pf = (size_t)P.ksplit[c] * M * B * P.Ho[c] * P.Wo[c];
            if (pf > P.part_floats) P.part_floats = pf;
        }
    }
    P.total_bytes = off * 4 + align_up(P.part_floats * 4, 256) + 256;
    return IPSR_OK;
}

size_t conv_gemm_ws_bytes(int transposed, int B, int Cred, int M, int Hin, int Win, int Hout, int Wout, int k, int stride, int pad, int dil)
{
    ConvPlan P;
    if (make_plan(transposed != 0, B, Cred, M, Hin, Win, Hout, Wout, k, stride, pad, dil, &P) != IPSR_OK) return 0;
    return P.total_bytes;
}

// w_cred_major: the weight tensor is [Cred][M][k][k] (true) or [M][Cred][k][k] (false).
int launch_conv_gemm(bool transposed, bool w_cred_major, const float* x, const float* w, float* y, int B, int Cred, int M,
                     int Hin, int Win, int Hout, int Wout, int k, int stride, int pad, int dil, void* ws, size_t ws_bytes, hipStream_t st)
{
    ConvPlan P;
    if (int rc = make_plan(transposed, B, Cred, M, Hin, Win, Hout, Wout, k, stride, pad, dil, &P)) return rc;
    if (ws_bytes < P.total_bytes) return fail(IPSR_ERR_WORKSPACE, "conv: workspace %zu < %zu", ws_bytes, P.total_bytes);
    float* wsf = static_cast<float*>(ws);
    size_t wp_total = 0;
    for (int c = 0; c < P.nclass; ++c) wp_total = P.wp_off[c] + align_up(P.wp_floats[c] * 4, 256) / 4 > wp_total ? P.wp_off[c] + align_up(P.wp_floats[c] * 4, 256) / 4 : wp_total;
    float* part = wsf + wp_total;
    const long sc = w_cred_major ? (long)M * k * k : (long)k * k;
    const long sm = w_cred_major ? (long)k * k : (long)Cred * k * k;
    if (P.need_zero)
        if (hipMemsetAsync(y, 0, (size_t)B * M * Hout * Wout * sizeof(float), st) != hipSuccess) return fail(IPSR_ERR_LAUNCH, "conv: hipMemsetAsync failed");
    for (int c = 0; c < P.nclass; ++c) {
        if (P.nstage[c] == 0) continue;
        ConvLaunch L;
        TapSrc ts;
        ts.k = k;
        if (!transposed) {
            L.dy0 = L.dx0 = -pad; L.dys = L.dxs = dil;
            ts.nsx = k; ts.r0 = ts.s0 = 0; ts.rs = ts.ss = 1;
            L.in_mul = stride; L.oy_mul = L.ox_mul = 1; L.oy_add = L.ox_add = 0;
        } else if (stride == 1) {
            L.dy0 = L.dx0 = pad; L.dys = L.dxs = -dil;
            ts.nsx = k; ts.r0 = ts.s0 = 0; ts.rs = ts.ss = 1;
            L.in_mul = 1; L.oy_mul = L.ox_mul = 1; L.oy_add = L.ox_add = 0;
        } else {
            int rr[4], ro[4], sr[4], so[4];
            const int nr = axis_taps(k, stride, pad, dil, P.py[c], rr, ro), nsx = axis_taps(k, stride, pad, dil, P.px[c], sr, so);
            // the valid taps of an axis form an arithmetic progression in r, and so do their offsets
            for (int a = 2; a < nr; ++a)
                if (rr[a] - rr[a - 1] != rr[1] - rr[0] || ro[a] - ro[a - 1] != ro[1] - ro[0]) return fail(IPSR_ERR_UNSUPPORTED, "conv: irregular tap set");
            for (int a = 2; a < nsx; ++a)
                if (sr[a] - sr[a - 1] != sr[1] - sr[0] || so[a] - so[a - 1] != so[1] - so[0]) return fail(IPSR_ERR_UNSUPPORTED, "conv: irregular tap set");
            if (nr != nsx) return fail(IPSR_ERR_UNSUPPORTED, "conv: %d x %d taps in a parity class", nr, nsx);
            L.dy0 = ro[0]; L.dys = nr > 1 ? ro[1] - ro[0] : 0;
            L.dx0 = so[0]; L.dxs = nsx > 1 ? so[1] - so[0] : 0;
            ts.nsx = nsx; ts.r0 = rr[0]; ts.rs = nr > 1 ? rr[1] - rr[0] : 0; ts.s0 = sr[0]; ts.ss = nsx > 1 ? sr[1] - sr[0] : 0;
            L.in_mul = 1; L.oy_mul = L.ox_mul = 2; L.oy_add = P.py[c]; L.ox_add = P.px[c];
        }
        {
            const int nsx = ts.nsx, na = P.NT[c] / nsx;
            L.minoff = 0;
            for (int a = 0; a < na; ++a)
                for (int b2 = 0; b2 < nsx; ++b2) L.minoff = min(L.minoff, (L.dy0 + a * L.dys) * Win + L.dx0 + b2 * L.dxs);
        }
        float* wp = wsf + P.wp_off[c];
        const int red_rows = P.nstage[c] * P.BK[c];
        conv_pack_weights_kernel<<<dim3(cdiv(red_rows, 32), cdiv(P.Mp, 32)), 256, 0, st>>>(w, wp, M, P.Mp, Cred, red_rows, P.NT[c], sc, sm, ts);
        L.x = x; L.wp = wp; L.y = y; L.part = part;
        L.B = B; L.Cred = Cred; L.Hin = Hin; L.Win = Win; L.M = M; L.Mp = P.Mp;
        L.Ho = P.Ho[c]; L.Wo = P.Wo[c]; L.Hout = Hout; L.Wout = Wout;
        L.nstage = P.nstage[c]; L.ksplit = P.ksplit[c]; L.stages_per_split = P.sps[c];
        L.m_tiles = P.Mp / CG_BM;
        const int ntot = B * L.Ho * L.Wo;
        L.n_tiles = cdiv(ntot, CG_BN);
        const int grid = L.m_tiles * L.n_tiles * L.ksplit;
        switch (P.NT[c]) {
            case 4: conv_gemm_kernel<4><<<grid, CG_THREADS, 0, st>>>(L); break;
            case 9: conv_gemm_kernel<9><<<grid, CG_THREADS, 0, st>>>(L); break;
            default: conv_gemm_kernel<16><<<grid, CG_THREADS, 0, st>>>(L); break;
        }
        if (int rc = check_launch("conv_gemm_kernel")) return rc;
        if (L.ksplit > 1) {
            conv_reduce_kernel<<<dim3(cdiv(ntot, 256), M), 256, 0, st>>>(part, L.ksplit, M, ntot, L.Ho, L.Wo, Hout, Wout, L.oy_mul, L.oy_add,
                                                                          L.ox_mul, L.ox_add, y);
            if (int rc = check_launch("conv_reduce_kernel")) return rc;
        }
    }
    return IPSR_OK;
}

}  // namespace ipsr

using namespace ipsr;

extern "C" {

static int conv_out_dim(int op, int n, int k, int stride, int pad, int dil)
{
    if (op == 0 || op == 1) return (n + 2 * pad - dil * (k - 1) - 1) / stride + 1;     // Conv2d
    return (n - 1) * stride - 2 * pad + dil * (k - 1) + 1;                              // ConvTranspose2d, output_padding 0
}

static int conv_args_ok(const char* who, int op, int B, int Cin, int H, int W, int Cout, int k, int stride, int pad, int dil)
{
    if (op < 0 || op > 3) return fail(IPSR_ERR_INVALID, "%s: op %d", who, op);
    if (B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1 || k < 1 || k > 4 || stride < 1 || pad < 0 || dil < 1)
        return fail(IPSR_ERR_INVALID, "%s: bad geometry B=%d Cin=%d H=%d W=%d Cout=%d k=%d stride=%d pad=%d dil=%d", who, B, Cin, H, W, Cout, k, stride, pad, dil);
    if (conv_out_dim(op, H, k, stride, pad, dil) < 1 || conv_out_dim(op, W, k, stride, pad, dil) < 1)
        return fail(IPSR_ERR_INVALID, "%s: empty output", who);
    return IPSR_OK;
}

size_t ipsr_conv2d_workspace_bytes(int op, int B, int Cin, int H, int W, int Cout, int k, int stride, int pad, int dil)
{
    if (conv_args_ok("ipsr_conv2d_workspace_bytes", op, B, Cin, H, W, Cout, k, stride, pad, dil)) return 0;
    const int Ho = conv_out_dim(op, H, k, stride, pad, dil), Wo = conv_out_dim(op, W, k, stride, pad, dil);
    switch (op) {
        case 0: return conv_gemm_ws_bytes(0, B, Cin, Cout, H, W, Ho, Wo, k, stride, pad, dil);
        case 1: return conv_gemm_ws_bytes(1, B, Cout, Cin, Ho, Wo, H, W, k, stride, pad, dil);
        case 2: return conv_gemm_ws_bytes(1, B, Cin, Cout, H, W, Ho, Wo, k, stride, pad, dil);
        default: return conv_gemm_ws_bytes(0, B, Cout, Cin, Ho, Wo, H, W, k, stride, pad, dil);
    }
}

int ipsr_conv2d(int op, const float* in, const float* weight, float* out, int B, int Cin, int H, int W, int Cout,
                int k, int stride, int pad, int dil, void* ws, size_t ws_bytes, void* stream)
{
    if (!in || !weight || !out || !ws) return fail(IPSR_ERR_INVALID, "ipsr_conv2d: null pointer");
    if (int rc = conv_args_ok("ipsr_conv2d", op, B, Cin, H, W, Cout, k, stride, pad, dil)) return rc;
    if ((reinterpret_cast<uintptr_t>(ws) & 15u) != 0) return fail(IPSR_ERR_INVALID, "ipsr_conv2d: workspace must be 16-byte aligned");
    const int Ho = conv_out_dim(op, H, k, stride, pad, dil), Wo = conv_out_dim(op, W, k, stride, pad, dil);
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (op) {
        case 0:      // y[B,Cout,Ho,Wo] = conv(x[B,Cin,H,W], w[Cout,Cin,k,k])
            return launch_conv_gemm(false, false, in, weight, out, B, Cin, Cout, H, W, Ho, Wo, k, stride, pad, dil, ws, ws_bytes, st);
        case 1:      // dx[B,Cin,H,W] from dy[B,Cout,Ho,Wo], w[Cout,Cin,k,k]  (reduction over Cout: weight is [Cred][M])
            return launch_conv_gemm(true, true, in, weight, out, B, Cout, Cin, Ho, Wo, H, W, k, stride, pad, dil, ws, ws_bytes, st);
        case 2:      // y[B,Cout,Ho,Wo] = conv_transpose(x[B,Cin,H,W], w[Cin,Cout,k,k])  (weight is [Cred][M])
            return launch_conv_gemm(true, true, in, weight, out, B, Cin, Cout, H, W, Ho, Wo, k, stride, pad, dil, ws, ws_bytes, st);
        default:     // dx[B,Cin,H,W] from dy[B,Cout,Ho,Wo], w[Cin,Cout,k,k]  (a direct gather over dy; weight is [M][Cred])
            return launch_conv_gemm(false, false, in, weight, out, B, Cout, Cin, Ho, Wo, H, W, k, stride, pad, dil, ws, ws_bytes, st);
    }
}

}  // extern "C"
