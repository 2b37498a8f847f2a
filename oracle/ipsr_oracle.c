/*
 * ipsr_oracle.c — CPU restatement (ORACLE) of the IPSR patch-attention hot path of
 * Image-Processing-Systems-Laboratory/DeepInPainting.
 *
 * THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library, and only as the checker / the reported CPU baseline — never as a product
 * path.  The product (deepinpainting_amd) fails loudly when libipsr_hip.so is missing.
 *
 * Parity pinning: the reference holds no tests or golden vectors.  This restatement
 * is pinned against outputs of the reference ITSELF, run in the build container by
 * oracle/gen_golden.py (which imports /root/reference read-only) and committed as fixtures under
 * tests/golden/; tests/test_oracle_golden.py checks every function below against them.
 *
 * Entry points mirror include/ipsr_hip.h one-to-one with a `_cpu` suffix and HOST pointers, so the
 * same ctypes harness drives both.  Each function cites the reference code it restates
 * (file:line under /root/reference).
 *
 * Canonical arithmetic.  The reference leaves fp32 summation order to its BLAS/conv backend; the
 * restatement fixes one order per reduction (identical to what the HIP kernels do) so
 * that HIP-vs-oracle parity can be asserted bit-for-bit, incl. the arg-max indices:
 *   - correlation / reconstruction dot products: one fmaf chain in ascending reduction index
 *     (what v_mfma_f32_32x32x2_f32 computes when K is walked in order);
 *   - patch norm: 8 contiguous channel segments, fmaf chain inside each, partials added in order;
 *   - recurrence dot <u, o>: 64 "lanes", lane j owns the 8-channel chunks ch with ch % 64 == j
 *     (fmaf chain, ascending), then an xor-butterfly over the 64 partials (patches wider than 2048
 *     numbers: 256 lanes = four waves, see lane_dot);
 *   - everything else is element-wise with every product and sum rounded separately (no FMA
 *     contraction: build with -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define IPSR_OK 0
#define IPSR_ERR_INVALID -1
#define IPSR_ERR_UNSUPPORTED -2

#if defined(__x86_64__) && defined(__GNUC__) && !defined(IPSR_ORACLE_NO_CLONES)
#define HOT __attribute__((target_clones("default", "avx2,fma")))
#else
#define HOT
#endif

int ipsr_oracle_abi_version(void) { return 1; }

/* ---- K1: util.cal_feat_mask (util/util.py:68-84) -------------------------------------------
 * The reference convolves the 0/1 mask with all-1/16 4x4 kernels (stride 2, zero padding 1) `layers`
 * times in fp32 and thresholds once at the end.  Every intermediate is count/16^level with
 * count <= 16^level, exactly representable in fp32 for layers <= 5, so integer counting is exact. */
static int out_dim(int n) { return (n + 2 - 4) / 2 + 1; }

int ipsr_feat_mask_cpu(const uint8_t* mask, int H, int W, int layers, float threshold, uint8_t* feat)
{
    if (!mask || !feat || H < 2 || W < 2 || layers < 1 || layers > 5) return IPSR_ERR_INVALID;
    int h = H, w = W;
    uint32_t* cur = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)H * W);
    uint32_t* nxt = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)H * W);
    if (!cur || !nxt) { free(cur); free(nxt); return IPSR_ERR_INVALID; }
    for (size_t i = 0; i < (size_t)H * W; ++i) cur[i] = mask[i] ? 1u : 0u;
    for (int l = 0; l < layers; ++l) {
        int oh = out_dim(h), ow = out_dim(w);
        if (oh < 1 || ow < 1) { free(cur); free(nxt); return IPSR_ERR_INVALID; }
        for (int y = 0; y < oh; ++y)
            for (int x = 0; x < ow; ++x) {
                uint32_t s = 0;
                for (int dy = 0; dy < 4; ++dy) {
                    int yy = 2 * y - 1 + dy;
                    if (yy < 0 || yy >= h) continue;
                    for (int dx = 0; dx < 4; ++dx) {
                        int xx = 2 * x - 1 + dx;
                        if (xx < 0 || xx >= w) continue;
                        s += cur[(size_t)yy * w + xx];
                    }
                }
                nxt[(size_t)y * ow + x] = s;
            }
        uint32_t* t = cur; cur = nxt; nxt = t;
        h = oh; w = ow;
    }
    float scale = 1.0f;
    for (int l = 0; l < layers; ++l) scale *= (1.0f / 16.0f);
    for (size_t i = 0; i < (size_t)h * w; ++i) feat[i] = ((float)cur[i] * scale > threshold) ? 1 : 0;
    free(cur); free(nxt);
    return IPSR_OK;
}

/* ---- K2: util.cal_mask_given_mask_thred (util/util.py:88-161) ------------------------------ */
int ipsr_index_prep_cpu(const uint8_t* feat, int h, int w, int patch, int stride, int mask_thred,
                        int32_t* flag, int32_t* mask_point_idx, int32_t* count)
{
    if (!feat || !flag || !mask_point_idx || !count || patch < 1 || stride < 1 || h < patch || w < patch)
        return IPSR_ERR_INVALID;
    int nH = (h - patch) / stride + 1, nW = (w - patch) / stride + 1;  /* util.py:95-96 */
    int N = nH * nW, m = 0;
    for (int i = 0; i < N; ++i) {
        int py = i / nW, px = i % nW, s = 0;                            /* util.py:114-118 */
        for (int dy = 0; dy < patch; ++dy)
            for (int dx = 0; dx < patch; ++dx) s += feat[(size_t)(py * stride + dy) * w + px * stride + dx];
        flag[i] = (s >= mask_thred) ? 1 : 0;                            /* util.py:132-135 */
        if (flag[i]) mask_point_idx[m++] = i;
    }
    for (int i = m; i < N; ++i) mask_point_idx[i] = -1;
    *count = m;
    return IPSR_OK;
}

/* ---- K3: NonparametricShift._extract_patches/_build (util/NonparametricShift.py:36-73) ----- */
/* squared L2 norm over the channels of position k: 8 contiguous channel segments, one fmaf chain each, partials added in order */
static float chan_sumsq(const float* xb, int C, int N, int k)
{
    const int L = (C + 7) / 8;
    float tot = 0.0f;
    for (int s = 0; s < 8; ++s) {
        float part = 0.0f;
        int c1 = (s + 1) * L < C ? (s + 1) * L : C;
        for (int c = s * L; c < c1; ++c) part = fmaf(xb[(size_t)c * N + k], xb[(size_t)c * N + k], part);
        tot = (s == 0) ? part : tot + part;
    }
    return tot;
}

HOT int ipsr_patch_normalize_cpu(const float* x, int B, int C, int N, float* xn, float* inv)
{
    if (!x || !xn || !inv || B < 1 || C < 1 || N < 1) return IPSR_ERR_INVALID;
    for (int b = 0; b < B; ++b) {
        const float* xb = x + (size_t)b * C * N;
        float* xnb = xn + (size_t)b * C * N;
        for (int k = 0; k < N; ++k) {
            const float tot = chan_sumsq(xb, C, N, k);
            /* enc_patches[i]*(1/(enc_patches[i].norm(2)+1e-8))   NonparametricShift.py:40 */
            float iv = 1.0f / (sqrtf(tot) + 1e-8f);
            inv[(size_t)b * N + k] = iv;
        }
        for (int c = 0; c < C; ++c)
            for (int k = 0; k < N; ++k) xnb[(size_t)c * N + k] = xb[(size_t)c * N + k] * inv[(size_t)b * N + k];
    }
    return IPSR_OK;
}

/* ---- K4+K5: conv_enc(ref) + MaxCoord (IPSRFunction.py:59-65, MaxCoord.py:16-28) ------------ */
HOT int ipsr_corr_argmax_cpu(const float* xn, const float* ref, int B, int C, int N,
                             int32_t* ind, float* vmax, float* S_out)
{
    if (!xn || !ref || !ind || !vmax || B < 1 || C < 1 || N < 1) return IPSR_ERR_INVALID;
    /* every (b, q-block) is independent, so threading changes nothing in the results */
    enum { QB = 256 };
    const int nqb = (N + QB - 1) / QB;
#pragma omp parallel for schedule(static)
    for (int job = 0; job < B * nqb; ++job) {
        const int b = job / nqb, q0 = (job % nqb) * QB;
        const int nq = N - q0 < QB ? N - q0 : QB;
        const float* A = xn + (size_t)b * C * N;
        const float* R = ref + (size_t)b * C * N + q0;
        float acc[QB];
        float* best = vmax + (size_t)b * N + q0;
        int32_t* bi = ind + (size_t)b * N + q0;
        for (int k = 0; k < N; ++k) {
            for (int q = 0; q < nq; ++q) acc[q] = 0.0f;
            for (int c = 0; c < C; ++c) {
                const float a = A[(size_t)c * N + k];
                const float* r = R + (size_t)c * N;
                for (int q = 0; q < nq; ++q) acc[q] = fmaf(a, r[q], acc[q]);
            }
            if (S_out) memcpy(S_out + ((size_t)b * N + k) * N + q0, acc, sizeof(float) * nq);
            if (k == 0) for (int q = 0; q < nq; ++q) { best[q] = acc[q]; bi[q] = 0; }
            else for (int q = 0; q < nq; ++q) {
                /* torch.max semantics (MaxCoord.py:23): first max wins; a NaN counts as the maximum and the FIRST NaN wins */
                const float v = acc[q], cur = best[q];
                if (v > cur || (v != v && cur == cur)) { best[q] = v; bi[q] = k; }
            }
        }
    }
    return IPSR_OK;
}

/* Dot used by the coherent-attention recurrence (IPSRFunction.py:109-118), in the order the HIP wave(s) reduce it:
 * "lanes" own 8-element chunks round-robin (chunk ch -> lane ch % L), one fmaf chain per lane in ascending ch; the 64
 * partials of a wave are summed by an xor-butterfly.  L = 64 (one wave) for patches up to 2048 numbers; wider patches
 * (shift_sz > 1: C*p*p numbers) are spread over four waves, L = 256, and the four wave sums are added as (s0+s1)+(s2+s3). */
HOT static float lane_dot(const float* u, const float* o, int C)
{
    const int nwave = ((C + 7) & ~7) > 2048 ? 4 : 1;
    const int L = 64 * nwave;
    float p[256];
    for (int j = 0; j < L; ++j) p[j] = 0.0f;
    const int nch = (C + 7) / 8;
    for (int ch = 0; ch < nch; ++ch) {
        int j = ch % L;
        int c1 = ch * 8 + 8 < C ? ch * 8 + 8 : C;
        float a = p[j];
        for (int c = ch * 8; c < c1; ++c) a = fmaf(u[c], o[c], a);
        p[j] = a;
    }
    float sw[4];
    for (int wv = 0; wv < nwave; ++wv) {
        float* pw = p + 64 * wv;
        for (int s = 1; s < 64; s <<= 1) {
            float t[64];
            for (int j = 0; j < 64; ++j) t[j] = pw[j] + pw[j ^ s];
            memcpy(pw, t, sizeof(t));
        }
        sw[wv] = pw[0];
    }
    return nwave == 1 ? sw[0] : (sw[0] + sw[1]) + (sw[2] + sw[3]);
}

/* Sparse form of trunc(kbar) kept for the backward, per sample (int32 words), two CSRs over the patch index k:
 *   offA[N+1] | entA_q[N]                   the non-masked q with ind[q] == k (weight 1), ascending q (N-M entries)
 *   offB[N+1] | entB_q[capB] | entB_w[capB] the masked rows l with trunc(a_l[k]) != 0: q = mask_point_idx[l],
 *                                           weight trunc(a_l[k]) (fp32 bits), ascending l;  capB = M(M+1)/2
 *                                           (row l of the attention has at most l+1 non-zeros). */
static size_t bwd_capB(int M) { return (size_t)M * (M + 1) / 2; }
size_t ipsr_bwd_index_ints_cpu(int N, int M) { return 2 * ((size_t)N + 1) + (size_t)N + 2 * bwd_capB(M); }

/* ---- whole layer forward: IPSRFunction.forward (models/IPSRFunction.py:13-140) -------------- */
/* ---- shift_sz > 1: patch unfold / overlap-add fold around the same algorithm ------------------
 * NonparametricShift._extract_patches (util/NonparametricShift.py:59-73) unfolds p x p windows (stride 1, no padding)
 * into patches [N', C, p, p], N' = (h-p+1)(w-p+1); every later step of IPSRFunction.forward (:46-133) treats a patch as
 * one flat vector of K = C*p*p numbers (norm over the whole patch :40, correlation conv with p x p kernels :59, the
 * recurrence's full-patch dot :109-116), and the final ConvTranspose2d (:130) overlap-adds the p x p patches back.
 * So shift_sz = p is: unfold -> the p = 1 algorithm on a [K, N'] "feature" -> fold.
 * Row order of the unfolded matrix: k = (c*p + dy)*p + dx (the flattening of a [C,p,p] patch); the fold adds the up
 * to p*p contributions of a pixel in ascending (dy, dx), each add rounded (canonical order). */
int ipsr_unfold_cpu(const float* x, int B, int C, int h, int w, int patch, float* xu)
{
    if (!x || !xu || B < 1 || C < 1 || patch < 1 || h < patch || w < patch) return IPSR_ERR_INVALID;
    const int nH = h - patch + 1, nW = w - patch + 1, Np = nH * nW, K = C * patch * patch;
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int dy = 0; dy < patch; ++dy)
                for (int dx = 0; dx < patch; ++dx) {
                    float* row = xu + ((size_t)b * K + ((size_t)c * patch + dy) * patch + dx) * Np;
                    const float* src = x + ((size_t)b * C + c) * h * w;
                    for (int i = 0; i < nH; ++i)
                        for (int j = 0; j < nW; ++j) row[i * nW + j] = src[(size_t)(i + dy) * w + j + dx];
                }
    return IPSR_OK;
}

int ipsr_fold_cpu(const float* yu, int B, int C, int h, int w, int patch, float* out)
{
    if (!yu || !out || B < 1 || C < 1 || patch < 1 || h < patch || w < patch) return IPSR_ERR_INVALID;
    const int nH = h - patch + 1, nW = w - patch + 1, Np = nH * nW, K = C * patch * patch;
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int y = 0; y < h; ++y)
                for (int xx = 0; xx < w; ++xx) {
                    float acc = 0.0f;
                    for (int dy = 0; dy < patch; ++dy)
                        for (int dx = 0; dx < patch; ++dx) {
                            const int i = y - dy, j = xx - dx;
                            if (i < 0 || i >= nH || j < 0 || j >= nW) continue;
                            acc = acc + yu[((size_t)b * K + ((size_t)c * patch + dy) * patch + dx) * Np + (size_t)i * nW + j];
                        }
                    out[(((size_t)b * C + c) * h + y) * w + xx] = acc;
                }
    return IPSR_OK;
}

/* shift_sz > 1: the correlation of p x p windows (the reference's conv2d with p x p kernels, IPSRFunction.py:59) and the window
 * norm (NonparametricShift.py:40), written as what they are — sums over the p*p taps of the 1x1 quantities:
 *     nsq[k'] = sum_{dy,dx} n1[(ky+dy)*w + kx+dx],                 n1[a] = channel sum of squares at position a (as for p = 1)
 *     inv[k'] = 1 / (sqrt(nsq[k']) + 1e-8)
 *     S[k'][q'] = inv[k'] * sum_{dy,dx} R[(ky+dy)*w + kx+dx][(qy+dy)*w + qx+dx],     R[a][b] = <x[:,a], ref[:,b]> (one fmaf chain over c)
 * (taps in ascending (dy, dx), plain adds from the first term; one multiply at the end).  Mathematically identical to
 * normalising the unfolded patches and contracting over C*p*p; canonical order chosen so that the GPU computes R ONCE
 * (2*N*N*C flop instead of 2*N'*N'*C*p*p: 8x fewer at p = 3) and sums shifted diagonals of it. */
HOT static int window_corr_argmax(const float* x, const float* ref, int B, int C, int h, int w, int patch,
                                  float* inv, int32_t* ind, float* vmax)
{
    const int N = h * w, nH = h - patch + 1, nW = w - patch + 1, Np = nH * nW;
    float* n1 = (float*)malloc(sizeof(float) * N);
    float* R = (float*)malloc(sizeof(float) * (size_t)N * N);
    if (!n1 || !R) { free(n1); free(R); return IPSR_ERR_INVALID; }
    for (int b = 0; b < B; ++b) {
        const float* xb = x + (size_t)b * C * N;
        const float* rb = ref + (size_t)b * C * N;
        for (int a = 0; a < N; ++a) n1[a] = chan_sumsq(xb, C, N, a);
        for (int k = 0; k < Np; ++k) {
            const int base = (k / nW) * w + k % nW;
            float tot = 0.0f;
            int first = 1;
            for (int dy = 0; dy < patch; ++dy)
                for (int dx = 0; dx < patch; ++dx) { const float v = n1[base + dy * w + dx]; tot = first ? v : tot + v; first = 0; }
            inv[(size_t)b * Np + k] = 1.0f / (sqrtf(tot) + 1e-8f);
        }
#pragma omp parallel for schedule(static)
        for (int a = 0; a < N; ++a) {
            float* Ra = R + (size_t)a * N;
            for (int q = 0; q < N; ++q) Ra[q] = 0.0f;
            for (int c = 0; c < C; ++c) {
                const float xa = xb[(size_t)c * N + a];
                const float* r = rb + (size_t)c * N;
                for (int q = 0; q < N; ++q) Ra[q] = fmaf(xa, r[q], Ra[q]);
            }
        }
#pragma omp parallel for schedule(static)
        for (int q = 0; q < Np; ++q) {
            const int qb = (q / nW) * w + q % nW;
            float best = 0.0f;
            int bi = 0;
            for (int k = 0; k < Np; ++k) {
                const int kb = (k / nW) * w + k % nW;
                float acc = 0.0f;
                int first = 1;
                for (int dy = 0; dy < patch; ++dy)
                    for (int dx = 0; dx < patch; ++dx) {
                        const float v = R[(size_t)(kb + dy * w + dx) * N + qb + dy * w + dx];
                        acc = first ? v : acc + v;
                        first = 0;
                    }
                const float sv = inv[(size_t)b * Np + k] * acc;
                if (k == 0 || sv > best || (sv != sv && best == best)) { best = sv; bi = k; }      /* torch.max semantics */
            }
            ind[(size_t)b * Np + q] = bi;
            vmax[(size_t)b * Np + q] = best;
        }
    }
    free(n1); free(R);
    return IPSR_OK;
}

/* the layer behind the correlation: recurrence, kbar, reconstruction and the sparse trunc(kbar), on patches x [B,C,N] with their
 * inverse norms `inv`, arg-max `ind` and maximum `vmax` already known */
static int attention_core(const float* x, const float* inv, const int32_t* ind, const float* vmax, const int32_t* mask_point_idx, int M,
                          int B, int C, int N, float* out, float* attn_rows, int32_t* bwd_index);

HOT int ipsr_forward_cpu(const float* x, const float* ref, const int32_t* mask_point_idx, int M,
                         int B, int C, int h, int w, int patch, int stride,
                         float* out, int32_t* ind, float* vmax, float* attn_rows, int32_t* bwd_index)
{
    if (!x || !ref || !out || !ind || !vmax || B < 1 || C < 1 || h < 1 || w < 1 || M < 0) return IPSR_ERR_INVALID;
    if (stride != 1 || patch < 1) return IPSR_ERR_UNSUPPORTED;
    if (M > 0 && (!mask_point_idx || !attn_rows)) return IPSR_ERR_INVALID;
    if (patch > 1) {                                  /* outputs ind/vmax/attn_rows/bwd_index live on the N' window grid */
        if (h < patch || w < patch) return IPSR_ERR_INVALID;
        const int nH = h - patch + 1, nW = w - patch + 1, K = C * patch * patch;
        const size_t un = (size_t)B * K * nH * nW;
        float* xu = (float*)malloc(sizeof(float) * un);
        float* ou = (float*)malloc(sizeof(float) * un);
        float* inv = (float*)malloc(sizeof(float) * (size_t)B * nH * nW);
        int rc = ipsr_unfold_cpu(x, B, C, h, w, patch, xu);
        if (rc == IPSR_OK) rc = window_corr_argmax(x, ref, B, C, h, w, patch, inv, ind, vmax);
        if (rc == IPSR_OK) rc = attention_core(xu, inv, ind, vmax, mask_point_idx, M, B, K, nH * nW, ou, attn_rows, bwd_index);
        if (rc == IPSR_OK) rc = ipsr_fold_cpu(ou, B, C, h, w, patch, out);
        free(xu); free(ou); free(inv);
        return rc;
    }
    const int N = h * w;
    float* xn = (float*)malloc(sizeof(float) * (size_t)B * C * N);
    float* inv = (float*)malloc(sizeof(float) * (size_t)B * N);
    int rc = ipsr_patch_normalize_cpu(x, B, C, N, xn, inv);
    if (rc == IPSR_OK) rc = ipsr_corr_argmax_cpu(xn, ref, B, C, N, ind, vmax, NULL);
    if (rc == IPSR_OK) rc = attention_core(x, inv, ind, vmax, mask_point_idx, M, B, C, N, out, attn_rows, bwd_index);
    free(xn); free(inv);
    return rc;
}

HOT static int attention_core(const float* x, const float* inv, const int32_t* ind, const float* vmax, const int32_t* mask_point_idx, int M,
                              int B, int C, int N, float* out, float* attn_rows, int32_t* bwd_index)
{
    int8_t* is_mask = (int8_t*)calloc(N, 1);
    for (int l = 0; l < M; ++l) is_mask[mask_point_idx[l]] = 1;
    float* o = (float*)malloc(sizeof(float) * C);
    float* u = (float*)malloc(sizeof(float) * C);
    float* kk = (float*)malloc(sizeof(float) * C);

    for (int b = 0; b < B; ++b) {
        const float* xb = x + (size_t)b * C * N;
        const float* invb = inv + (size_t)b * N;
        const int32_t* indb = ind + (size_t)b * N;
        const float* vb = vmax + (size_t)b * N;
        float* outb = out + (size_t)b * C * N;
        float* attn = attn_rows ? attn_rows + (size_t)b * M * N : NULL;

        /* coherent-attention recurrence over the masked positions in raster order (:82-129) */
        for (int l = 0; l < M; ++l) {
            const int q = mask_point_idx[l];
            const int kq = indb[q];
            for (int c = 0; c < C; ++c) kk[c] = xb[(size_t)c * N + kq];          /* known_region  :94 */
            float* a = attn + (size_t)l * N;
            if (l == 0) {                                                         /* :98-101 */
                memcpy(o, kk, sizeof(float) * C);
                for (int k = 0; k < N; ++k) a[k] = 0.0f;
                a[kq] = 1.0f;
            } else {
                for (int c = 0; c < C; ++c) u[c] = xb[(size_t)c * N + q] * invb[q];  /* value_2 :109 (= the normalised patch) */
                const float at = lane_dot(u, o, C);                               /* :116 */
                const float v = vb[q];                                            /* vamx_mask :70 */
                const float s = at + v;
                const float wn = at / s, wo = v / s;                              /* :120-121 */
                for (int c = 0; c < C; ++c) { float t0 = wn * o[c], t1 = wo * kk[c]; o[c] = t0 + t1; }   /* :122 */
                const float* ap = attn + (size_t)(l - 1) * N;
                for (int k = 0; k < N; ++k) a[k] = ap[k] * wn;                    /* :123 */
                a[kq] = a[kq] + wo;                                               /* :124 */
            }
        }
        /* reconstruction: conv_transpose2d(kbar, raw patches) (:130-133) */
        for (int q = 0; q < N; ++q)
            if (!is_mask[q]) { const int kq = indb[q]; for (int c = 0; c < C; ++c) outb[(size_t)c * N + q] = xb[(size_t)c * N + kq]; }
        for (int l = 0; l < M; ++l) {
            const int q = mask_point_idx[l];
            const float* a = attn + (size_t)l * N;
            for (int c = 0; c < C; ++c) {
                const float* xr = xb + (size_t)c * N;
                float acc = 0.0f;
                for (int k = 0; k < N; ++k) acc = fmaf(a[k], xr[k], acc);
                outb[(size_t)c * N + q] = acc;
            }
        }
        /* sparse form of trunc(kbar) for the backward (:36,134 — kbar is stored in a LongTensor) */
        if (bwd_index) {
            const size_t capB = bwd_capB(M);
            int32_t* offA = bwd_index + (size_t)b * ipsr_bwd_index_ints_cpu(N, M);
            int32_t* entA = offA + N + 1;
            int32_t* offB = entA + N;
            int32_t* entB_q = offB + N + 1;
            float* entB_w = (float*)(entB_q + capB);
            for (int k = 0; k <= N; ++k) { offA[k] = 0; offB[k] = 0; }
            for (int q = 0; q < N; ++q) if (!is_mask[q]) offA[indb[q] + 1]++;
            for (int l = 0; l < M; ++l) {
                const float* a = attn + (size_t)l * N;
                for (int k = 0; k < N; ++k) if (truncf(a[k]) != 0.0f) offB[k + 1]++;
            }
            for (int k = 0; k < N; ++k) { offA[k + 1] += offA[k]; offB[k + 1] += offB[k]; }
            int32_t* fillA = (int32_t*)calloc(N, sizeof(int32_t));
            int32_t* fillB = (int32_t*)calloc(N, sizeof(int32_t));
            for (int q = 0; q < N; ++q) if (!is_mask[q]) { const int k = indb[q]; entA[offA[k] + fillA[k]++] = q; }
            for (int l = 0; l < M; ++l) {
                const float* a = attn + (size_t)l * N;
                for (int k = 0; k < N; ++k) {
                    const float t = truncf(a[k]);
                    if (t != 0.0f) { const int e = offB[k] + fillB[k]++; entB_q[e] = mask_point_idx[l]; entB_w[e] = t; }
                }
            }
            free(fillA); free(fillB);
        }
    }
    free(is_mask); free(o); free(u); free(kk);
    return IPSR_OK;
}

/* ---- K8: IPSRFunction.backward (models/IPSRFunction.py:144-178) ------------------------------ */
static int backward_core(const float* g, int M, const int32_t* bwd_index, float triple_w, int B, int C, int N,
                         int identity, float* gin);

int ipsr_backward_cpu(const float* g, const int32_t* mask_point_idx, int M, const float* attn_rows,
                      const int32_t* bwd_index, float triple_w, int B, int C, int h, int w, float* gin)
{
    if (!g || !gin || !bwd_index || B < 1 || C < 1 || h < 1 || w < 1 || M < 0) return IPSR_ERR_INVALID;
    (void)mask_point_idx; (void)attn_rows;   /* everything the backward needs is in bwd_index */
    return backward_core(g, M, bwd_index, triple_w, B, C, h * w, 1, gin);
}

/* shift_sz > 1 — EXTENSION: the reference has no working backward for p > 1 (its forward already raises at :134 and
 * :158-170 index an N x N matrix by h*w).  This is the same rule (:144-178: kbar is a constant, truncated to integers;
 * grad_in = grad_out + triple_w * d out / d patches) carried through the unfold/fold pair, whose adjoints are each other:
 *     GU = unfold(g);   TU[:,k] = triple_w * sum_q trunc(kbar)[k][q] GU[:,q];   grad_in = g + fold(TU).           */
int ipsr_backward_patch_cpu(const float* g, int M, const int32_t* bwd_index, float triple_w,
                            int B, int C, int h, int w, int patch, float* gin)
{
    if (!g || !gin || !bwd_index || B < 1 || C < 1 || patch < 1 || h < patch || w < patch || M < 0) return IPSR_ERR_INVALID;
    if (patch == 1) return backward_core(g, M, bwd_index, triple_w, B, C, h * w, 1, gin);
    const int nH = h - patch + 1, nW = w - patch + 1, K = C * patch * patch;
    const size_t un = (size_t)B * K * nH * nW;
    float* gu = (float*)malloc(sizeof(float) * un);
    float* tu = (float*)malloc(sizeof(float) * un);
    int rc = ipsr_unfold_cpu(g, B, C, h, w, patch, gu);
    if (rc == IPSR_OK) rc = backward_core(gu, M, bwd_index, triple_w, B, K, nH * nW, 0, tu);
    if (rc == IPSR_OK) rc = ipsr_fold_cpu(tu, B, C, h, w, patch, gin);
    if (rc == IPSR_OK) {
        const size_t n = (size_t)B * C * h * w;
        for (size_t i = 0; i < n; ++i) gin[i] = g[i] + gin[i];
    }
    free(gu); free(tu);
    return rc;
}

static int backward_core(const float* g, int M, const int32_t* bwd_index, float triple_w, int B, int C, int N,
                         int identity, float* gin)
{
    const size_t capB = bwd_capB(M);
    for (int b = 0; b < B; ++b) {
        const int32_t* offA = bwd_index + (size_t)b * ipsr_bwd_index_ints_cpu(N, M);
        const int32_t* entA = offA + N + 1;
        const int32_t* offB = entA + N;
        const int32_t* entB_q = offB + N + 1;
        const float* entB_w = (const float*)(entB_q + capB);
        for (int c = 0; c < C; ++c) {
            const float* gr = g + ((size_t)b * C + c) * N;
            float* go = gin + ((size_t)b * C + c) * N;
            for (int k = 0; k < N; ++k) {
                float acc = 0.0f;
                for (int e = offA[k]; e < offA[k + 1]; ++e) acc = acc + gr[entA[e]];                      /* one-hot rows :129 */
                for (int e = offB[k]; e < offB[k + 1]; ++e) acc = fmaf(entB_w[e], gr[entB_q[e]], acc);   /* truncated masked rows */
                const float t = acc * triple_w;                                                 /* :173 */
                go[k] = identity ? gr[k] + t : t;
            }
        }
    }
    return IPSR_OK;
}

/* ---- K9: InnerCos.forward / InnerCos2.forward (models/InnerCos.py:30-41, InnerCos2.py:34-46) - */
int innercos_loss_cpu(const float* x, int B, int Cx, int Cuse, int N, const float* mask,
                      const float* target, float strength, float* loss)
{
    if (!x || !mask || !target || !loss || B < 1 || Cuse < 1 || Cx < Cuse || N < 1) return IPSR_ERR_INVALID;
    double acc = 0.0;
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < Cuse; ++c) {
            const float* xr = x + ((size_t)b * Cx + c) * N;
            const float* tr = target + ((size_t)b * Cuse + c) * N;
            for (int n = 0; n < N; ++n) {
                float m = xr[n] * mask[n];           /* InnerCos.py:34 */
                float d = m * strength - tr[n];      /* :36 */
                acc += (double)(d * d);
            }
        }
    *loss = (float)(acc / ((double)B * Cuse * N));
    return IPSR_OK;
}

int innercos_loss_backward_cpu(const float* x, int B, int Cx, int Cuse, int N, const float* mask,
                               const float* target, float strength, const float* grad_loss, float* grad_x)
{
    if (!x || !mask || !target || !grad_loss || !grad_x || B < 1 || Cuse < 1 || Cx < Cuse || N < 1) return IPSR_ERR_INVALID;
    const float scale = (*grad_loss) * (2.0f / (float)((double)B * Cuse * N));
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < Cx; ++c) {
            float* gr = grad_x + ((size_t)b * Cx + c) * N;
            if (c >= Cuse) { for (int n = 0; n < N; ++n) gr[n] = 0.0f; continue; }
            const float* xr = x + ((size_t)b * Cx + c) * N;
            const float* tr = target + ((size_t)b * Cuse + c) * N;
            for (int n = 0; n < N; ++n) {
                float m = xr[n] * mask[n];
                float d = m * strength - tr[n];
                float ms = mask[n] * strength;
                gr[n] = (d * scale) * ms;
            }
        }
    return IPSR_OK;
}
