/*
 * ipsr_hip.h — C-ABI of libipsr_hip.so, the MI355X (gfx950) implementation of the
 * IPSR / CSA patch-attention hot path of Image-Processing-Systems-Laboratory/DeepInPainting.
 *
 * The reference has NO native/FFI interface (it is pure Python on torch); its "plugin API" for this
 * path is the Python surface IPSRFunction.apply / IPSR_model / InnerCos / util.* .  Each entry point
 * below therefore cites the reference Python code it replaces (file:line under /root/reference).
 * The Python host side (deepinpainting_amd/models, deepinpainting_amd/util) binds these through
 * ctypes; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is a DEVICE pointer unless marked [host].
 *   - all tensors are dense, row-major, NCHW:  x[b][c][y][x]  ->  ((b*C + c)*h + y)*w + x.
 *     N = number of patches = nH*nW with nH = (h-patch)/stride+1 (reference util/util.py:95-98).
 *     patch (the reference's shift_sz) >= 1 with stride == 1 is implemented (the reference itself raises for
 *     shift_sz != 1 at models/IPSRFunction.py:134, see ipsr_forward below); stride != 1 returns IPSR_ERR_UNSUPPORTED.
 *   - `stream` is a hipStream_t passed as void*; NULL = the default stream.  No entry point
 *     allocates, frees or synchronises: the caller owns every buffer incl. the workspace whose size
 *     the matching *_workspace_bytes() query returns.  All entry points are re-entrant per stream.
 *   - return value: 0 = ok, <0 = error (codes below); ipsr_last_error() returns a thread-local
 *     message for the last failing call on this thread.
 *   - arithmetic is fp32 throughout with a FIXED summation order (documented in DESIGN.md §4 and
 *     restated in oracle/ipsr_oracle.c), so results are bit-reproducible run to run.
 */
#ifndef IPSR_HIP_H
#define IPSR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IPSR_OK                 0
#define IPSR_ERR_INVALID       -1   /* bad argument (null pointer, non-positive size ...)          */
#define IPSR_ERR_UNSUPPORTED   -2   /* patch/stride/size outside what this build implements       */
#define IPSR_ERR_WORKSPACE     -3   /* workspace too small                                        */
#define IPSR_ERR_LAUNCH        -4   /* hipLaunchKernel / hipMemsetAsync reported an error         */

/* ABI version: bumped whenever a signature changes. */
int ipsr_abi_version(void);
/* Thread-local message of the last failing call ("" if none). */
const char* ipsr_last_error(void);

/* ---- K1  feature-mask pyramid --------------------------------------------------------------
 * replaces util.cal_feat_mask (util/util.py:68-84): `layers` x Conv2d(1,1,4,2,1,weight=1/16) on the
 * 0/1 mask, then (> threshold) after the LAST conv only, returned as bytes.
 * mask  [H,W] u8 (0/1)  ->  feat [h,w] u8, h = H after `layers` halvings ((H-2)/2+1 each). */
size_t ipsr_feat_mask_workspace_bytes(int H, int W, int layers);
int ipsr_feat_mask(const uint8_t* mask, int H, int W, int layers, float threshold,
                   uint8_t* feat, void* ws, size_t ws_bytes, void* stream);

/* ---- K2  index prep -------------------------------------------------------------------------
 * replaces util.cal_mask_given_mask_thred (util/util.py:88-161): raster scan of the feature mask;
 * flag[i] = (sum of mask over patch i >= mask_thred); mask_point_idx = raster-ordered list of the
 * flagged patches (first *count entries valid; the rest is filled with -1).
 * (nonmask_point_idx == arange(N) and the dead flatten_offsets are produced by the Python host.) */
int ipsr_index_prep(const uint8_t* feat, int h, int w, int patch, int stride, int mask_thred,
                    int32_t* flag /*[N]*/, int32_t* mask_point_idx /*[N]*/, int32_t* count /*[1]*/,
                    void* stream);

/* ---- K3  patch unfold + L2 normalisation ----------------------------------------------------
 * replaces NonparametricShift._extract_patches/_build (util/NonparametricShift.py:36-73):
 * inv[b][k] = 1/(||x[b,:,k]||_2 + 1e-8);  xn[b][c][k] = x[b][c][k] * inv[b][k]   (patch == 1). */
int ipsr_patch_normalize(const float* x, int B, int C, int N, float* xn, float* inv, void* stream);

/* ---- K4+K5  cross-correlation + arg-max ------------------------------------------------------
 * replaces conv_enc(ref) + MaxCoord.update_output (models/IPSRFunction.py:59-65,
 * util/MaxCoord.py:16-28):  S[k][q] = <xn[:,k], ref[:,q]>;  ind[q] = argmax_k S[k][q] (lowest k on
 * ties), vmax[q] = max_k S[k][q].  S is never written unless S_out != NULL (tests only). */
size_t ipsr_corr_argmax_workspace_bytes(int B, int C, int N);
int ipsr_corr_argmax(const float* xn, const float* ref, int B, int C, int N,
                     int32_t* ind /*[B,N]*/, float* vmax /*[B,N]*/, float* S_out /*[B,N,N] or NULL*/,
                     void* ws, size_t ws_bytes, void* stream);

/* ---- whole layer forward (K3..K7) -------------------------------------------------------------
 * replaces IPSRFunction.forward (models/IPSRFunction.py:13-140).
 *   x, ref            [B,C,h,w] fp32           (ref = ref.relu4_3)
 *   mask_point_idx    [M] i32  raster-ordered masked patch positions in [0,N), shared by the whole batch; device
 *                     data, so NOT validated here — a stray value is clamped into range (wrong answer, never an
 *                     out-of-bounds access); the Python wrapper checks foreign tensors
 *                     (reference semantics: one mask per batch, models/IPSR.py:36)
 *   out               [B,C,h,w] fp32
 *   ind, vmax         [B,N]     arg-max / max of the correlation (kept for inspection + backward)
 *   attn_rows         [B,M,N]   the reference's `in_attention` rows (IPSRFunction.py:76,123-125)
 *   bwd_index         [B, ipsr_bwd_index_ints(N,M)] i32, optional (NULL when no backward will follow):
 *                     trunc(kbar) (the reference keeps kbar in a LongTensor, IPSRFunction.py:36,134) in
 *                     sparse form, per sample two CSRs over the patch index k:
 *                       offA[N+1] | entA_q[N]                    non-masked q with ind[q]==k (weight 1),
 *                                                                ascending q; N-M entries
 *                       offB[N+1] | entB_q[capB] | entB_w[capB]  masked rows with |a_l[k]| >= 1: q = mpi[l],
 *                                                                weight trunc(a_l[k]) (fp32 bits), ascending l;
 *                                                                capB = M(M+1)/2, offB[N] entries defined
 *   attn_rows         optional as well (NULL = the dense rows are not materialised; the layer itself works
 *                     on a compressed form).
 * patch (the reference's shift_sz) > 1, stride 1: p x p windows as in NonparametricShift._extract_patches
 * (util/NonparametricShift.py:59-73).  N above becomes the window grid N' = (h-p+1)(w-p+1): mask_point_idx indexes
 * windows (ipsr_index_prep with the same patch), ind/vmax are [B,N'], attn_rows [B,M,N'], bwd_index
 * [B, ipsr_bwd_index_ints(N',M)]; out stays [B,C,h,w] (the overlap-add of models/IPSRFunction.py:130).  The
 * reference's own forward computes exactly this up to :133 and then raises at :134 on a mis-sized buffer.
 * stride != 1 -> IPSR_ERR_UNSUPPORTED. */
size_t ipsr_bwd_index_ints(int N, int M);
size_t ipsr_forward_workspace_bytes(int B, int C, int h, int w, int M, int patch, int stride);
int ipsr_forward(const float* x, const float* ref, const int32_t* mask_point_idx, int M,
                 int B, int C, int h, int w, int patch, int stride,
                 float* out, int32_t* ind, float* vmax, float* attn_rows, int32_t* bwd_index,
                 void* ws, size_t ws_bytes, void* stream);

/* ---- per-sample masks / device-side counts (extension; BASELINE.json config 3 "irregular free-form masks") -----------------
 * The reference has ONE mask per batch and learns M on the host (util/util.py:132 syncs per position).  This entry point
 * takes the number of masked positions from DEVICE memory, one count per sample, so that
 *   (a) a batch whose samples have different holes runs as ONE launch sequence (mask_point_idx [B,Mcap], mpi_stride = Mcap,
 *       counts[b] <= Mcap valid entries in row b) — per-sample results equal a batch-of-one ipsr_forward call bit for bit;
 *   (b) the shared-mask case needs no host round trip at all (mask_point_idx [Mcap] straight from ipsr_index_prep with
 *       mpi_stride = 0 and counts[b] = its *count for every b).
 * Everything that ipsr_forward sizes by M is sized by the capacity Mcap here (workspace: query with M = Mcap; attn_rows
 * [B,Mcap,N] with the rows past counts[b] zero; bwd_index [B, ipsr_bwd_index_ints(N, Mcap)], to be handed to
 * ipsr_backward / ipsr_backward_patch with M = Mcap).  corr_bf16 != 0 selects the bf16 correlation (workspace: the bf16corr query). */
int ipsr_forward_masks(const float* x, const float* ref, const int32_t* mask_point_idx, int mpi_stride, const int32_t* counts /*[B] device*/,
                       int Mcap, int B, int C, int h, int w, int patch, int stride,
                       float* out, int32_t* ind, float* vmax, float* attn_rows, int32_t* bwd_index,
                       void* ws, size_t ws_bytes, void* stream, int corr_bf16);

/* ---- bf16 MFMA correlation (BASELINE.json config 5: "CDNA4 bf16 MFMA for patch-corr") — OPT-IN ----------------------
 * The same layer with the cross-correlation of models/IPSRFunction.py:59 computed on v_mfma_f32_32x32x16_bf16: both
 * operands (the normalised patches and ref.relu4_3) are rounded to bf16, products are exact, accumulation is fp32.
 * Everything else — normalisation, arg-max rule, recurrence, reconstruction, backward index — is the fp32 path above.
 * Not the reference's arithmetic: an arg-max moves wherever the two best patches are closer than bf16 rounding; the
 * agreement rate with ipsr_forward is measured by bench.py and tests/test_gpu_parity.py.  Same arguments as
 * ipsr_forward / ipsr_corr_argmax (fp32 tensors in and out; the bf16 copies live in the workspace).  Shapes: C*patch^2
 * a multiple of 64 and, for patch == 1, N a multiple of 128; anything else -> IPSR_ERR_UNSUPPORTED (no silent fp32 run). */
size_t ipsr_forward_bf16corr_workspace_bytes(int B, int C, int h, int w, int M, int patch, int stride);
int ipsr_forward_bf16corr(const float* x, const float* ref, const int32_t* mask_point_idx, int M,
                          int B, int C, int h, int w, int patch, int stride,
                          float* out, int32_t* ind, float* vmax, float* attn_rows, int32_t* bwd_index,
                          void* ws, size_t ws_bytes, void* stream);
size_t ipsr_corr_argmax_bf16_workspace_bytes(int B, int C, int N);
int ipsr_corr_argmax_bf16(const float* xn, const float* ref, int B, int C, int N,
                          int32_t* ind /*[B,N]*/, float* vmax /*[B,N]*/, void* ws, size_t ws_bytes, void* stream);

/* ---- K8  backward -----------------------------------------------------------------------------
 * replaces IPSRFunction.backward (models/IPSRFunction.py:144-178):
 * grad_in[b,:,k] = g[b,:,k] + triple_w * sum_q trunc(A[b])[q][k] * g[b,:,q].
 * Reads only grad_out and bwd_index; mask_point_idx / attn_rows are accepted for symmetry with
 * ipsr_forward (and the oracle twin) and may be NULL. */
int ipsr_backward(const float* grad_out, const int32_t* mask_point_idx, int M,
                  const float* attn_rows, const int32_t* bwd_index, float triple_w,
                  int B, int C, int h, int w, float* grad_in, void* stream);

/* Backward for any patch size.  patch == 1 is ipsr_backward (no workspace).  patch > 1 is an EXTENSION: the
 * reference has no working backward there (models/IPSRFunction.py:158-170 index an N x N matrix by h*w); this
 * carries the same rule (kbar constant and truncated, grad_in = grad_out + triple_w * d out / d patches) through
 * the unfold/fold pair:  grad_in = g + fold(triple_w * trunc(kbar) applied to unfold(g)).
 * bwd_index is the one ipsr_forward wrote for the same (h, w, patch, M). */
size_t ipsr_backward_workspace_bytes(int B, int C, int h, int w, int patch);
int ipsr_backward_patch(const float* grad_out, int M, const int32_t* bwd_index, float triple_w,
                        int B, int C, int h, int w, int patch, float* grad_in,
                        void* ws, size_t ws_bytes, void* stream);

/* ---- VGG16 feature-net glue ----------------------------------------------------------------------
 * replaces the Conv2d bias add + ReLU(inplace) [+ MaxPool2d(2,2)] passes of torchvision's vgg16.features as wrapped by
 * models/vgg16.py:6-37 (run three times per training step without gradients, models/IPSR.py:163,187,212-213):
 * the convolution is issued without bias and one pass does the rest, with the same arithmetic (bit-identical).
 *   ipsr_bias_act          in place  x[b,c,:] = act(x[b,c,:] + bias[c]);  act: 0 none, 1 ReLU, 2 LeakyReLU(slope);
 *                          bias may be NULL.  `tickets` (NULL, or C words of caller memory): zeroed here for the matching
 *                          ipsr_bias_act_backward's in-launch batch sum — see the norm entry points below
 *   ipsr_bias_relu_pool2   y[b,c,i,j] = max over the 2x2 window of relu(x + bias[c]);  y is [B,C,H/2,W/2]
 * io_bf16 (here and in the norm entry points below): 0 = the activation tensors are fp32, 1 = bf16 (BASELINE config 5:
 * convolutions under bf16 autocast); bias/gamma/beta, statistics and all arithmetic are fp32 either way. */
/* skip connection: torch.cat([y, x], 1) + the parent level's in-place ReLU (models/networks.py:270-278 with the `uprelu`
 * of :229 / :408) in one pass, and its backward (ReLU mask from `out`, the two channel slices as contiguous tensors).
 * y == NULL (forward) / dy == NULL (backward): only the x half is processed — the y half of `out` was written in place by
 * ipsr_instnorm_act_forward_slice, and its gradient is read in place by ipsr_instnorm_act_backward_slice. */
int ipsr_cat_relu_forward(const void* y, const void* x, int B, int C1, int C2, int HW, int io_bf16, void* out, void* stream);
int ipsr_cat_relu_backward(const void* grad_out, const void* out, int B, int C1, int C2, int HW, int io_bf16,
                           void* dy, void* dx, void* stream);
int ipsr_bias_act(void* x, const float* bias, int B, int C, int HW, int act, float slope, int io_bf16, unsigned* tickets, void* stream);
int ipsr_bias_relu_pool2(const void* x, const float* bias, int B, int C, int H, int W, int io_bf16, void* y, void* stream);
/* ipsr_bias_act with a second output y2 = relu(x + bias), a C-channel slice (pointer to its first element + batch stride in
 * elements) of the child level's concatenated tensor, and the matching backward dx = dy * act'(y) + dy2 * relu'(y): the level-1
 * blocks of the U-Nets, whose input comes from a convolution without a norm (see the y2 / dy2 forms of the norm entry points). */
int ipsr_bias_act_skip(void* x, const float* bias, int B, int C, int HW, int act, float slope, int io_bf16, void* y2, size_t y2_batch_stride,
                       unsigned* tickets, void* stream);
int ipsr_bias_act_backward_skip(const void* dy, const void* dy2, size_t dy2_batch_stride, const void* y, int act, float slope, int B, int C, int HW,
                                int io_bf16, void* dx, float* dbias_p, float* sums, unsigned* tickets, void* stream);

/* ---- conv-bias + InstanceNorm2d + activation ------------------------------------------------------
 * replaces the chain  Conv2d/ConvTranspose2d bias add -> nn.InstanceNorm2d(affine) -> LeakyReLU(0.2)/ReLU  that follows
 * every convolution of the U-Nets and discriminators (models/networks.py:220-259, 404-432, 470-497, 507-514) and its
 * autograd backward: one pass forward (1 read + 1 write), one pass backward (3 reads + 1 write), one (sample, channel)
 * plane of HW <= 16384 elements per workgroup.  act: 0 none, 1 ReLU, 2 LeakyReLU(slope); bias/gamma/beta may be NULL.
 *   forward : z = x + bias[c]; y = act((z - mean) * rstd * gamma[c] + beta[c]); mean/rstd [B*C] are kept for backward
 *   backward: dx [B,C,HW]; per-plane partials dgamma_p/dbeta_p/dbias_p [B*C] (any may be NULL).  `sums` [3,C] (NULL = the
 *             caller sums the partials itself): sums[k][c] = sum over b = 0..B-1, in that order, of the k-th partial array
 *             (rows of a NULL array are left untouched) — written by the last of channel c's B planes to finish, inside the
 *             same launch.  `y` is the forward OUTPUT (the activation's derivative is taken from its sign).
 *   tickets : the per-channel arrival counters of that in-launch sum — C 32-bit words of CALLER memory (the library keeps no
 *             device state).  Pass the same words to the forward entry point, which zeroes them, and to the backward, which
 *             needs them zero on entry and leaves them zero (a second backward over the same node finds them ready).  NULL in
 *             the forward = not zeroed; a backward with `sums` and no tickets -> IPSR_ERR_INVALID.
 *   ipsr_bias_act_backward: backward of ipsr_bias_act: dx = dy * act'(y), dbias_p[b*C+c] = sum of dx over the plane;
 *             `sums` [C] as above (needs dbias_p). */
int ipsr_instnorm_act_forward(const void* x, const float* bias, const float* gamma, const float* beta, float eps,
                              int act, float slope, int B, int C, int HW, int io_bf16,
                              void* y, float* mean, float* rstd, unsigned* tickets, void* stream);
int ipsr_instnorm_act_backward(const void* dy, const void* y, const void* x, const float* bias, const float* gamma,
                               const float* mean, const float* rstd, int act, float slope, int B, int C, int HW, int io_bf16,
                               void* dx, float* dgamma_p, float* dbeta_p, float* dbias_p, float* sums, unsigned* tickets, void* stream);
/* The `_slice` forms: y (forward) / dy and y (backward) are C channels of wider tensors — the output of a skip concatenation
 * torch.cat([y, x], 1) (models/networks.py:270-278) and its gradient — given by a pointer to their first element and their batch
 * strides in elements (>= C*HW); x, dx and the statistics stay dense.  With them a level's last normalisation writes straight into
 * the concatenated tensor and its backward reads the gradient's slice in place.
 * y2 / dy2 (optional, NULL = absent): a second output relu(normalised value) — the SKIP half of the child level's concatenated tensor,
 * written by the norm that produces the level's input — and its gradient: the two consumers of a level's input (its down path and
 * its skip connection) then meet inside this backward, dz = dy * act'(y) + dy2 * relu'(y), instead of in an add kernel. */
int ipsr_instnorm_act_forward_slice(const void* x, const float* bias, const float* gamma, const float* beta, float eps,
                                    int act, float slope, int B, int C, int HW, int io_bf16,
                                    void* y, size_t y_batch_stride, void* y2, size_t y2_batch_stride, float* mean, float* rstd, unsigned* tickets,
                                    void* stream);
int ipsr_instnorm_act_backward_slice(const void* dy, size_t dy_batch_stride, const void* dy2, size_t dy2_batch_stride,
                                     const void* y, size_t y_batch_stride, const void* x,
                                     const float* bias, const float* gamma, const float* mean, const float* rstd, int act, float slope,
                                     int B, int C, int HW, int io_bf16,
                                     void* dx, float* dgamma_p, float* dbeta_p, float* dbias_p, float* sums, unsigned* tickets, void* stream);
int ipsr_bias_act_backward(const void* dy, const void* y, int act, float slope, int B, int C, int HW, int io_bf16,
                           void* dx, float* dbias_p, float* sums, unsigned* tickets, void* stream);

/* ---- convolutions of the surrounding nets ---------------------------------------------------------------
 * replaces nn.Conv2d / nn.ConvTranspose2d (bias-free part; the bias rides in the fused epilogues above) of the U-Nets,
 * the PatchGAN discriminators and the VGG16 feature net — models/networks.py:220-259, 404-432, 470-495, 510-515,
 * models/vgg16.py:9-21 — and their input gradients, as one implicit GEMM per call on the fp32 matrix cores, NCHW in and out.
 *   op 0  Conv2d forward           in = x  [B,Cin,H,W]      weight [Cout,Cin,k,k]   out = y  [B,Cout,Ho,Wo]
 *   op 1  Conv2d backward-data     in = dy [B,Cout,Ho,Wo]   weight [Cout,Cin,k,k]   out = dx [B,Cin,H,W]
 *   op 2  ConvTranspose2d forward  in = x  [B,Cin,H,W]      weight [Cin,Cout,k,k]   out = y  [B,Cout,Ho,Wo]
 *   op 3  ConvTranspose2d bwd-data in = dy [B,Cout,Ho,Wo]   weight [Cin,Cout,k,k]   out = dx [B,Cin,H,W]
 * (Cin, H, W) always describe the module's INPUT; Ho = (H + 2*pad - dil*(k-1) - 1)/stride + 1 for Conv2d and
 * (H-1)*stride - 2*pad + dil*(k-1) + 1 for ConvTranspose2d (output_padding 0, groups 1, square kernels k <= 4).
 * Supported tap sets: 9 or 16 taps (k = 3, 4) with any stride for the direct forms (op 0, 3) and stride 1 or 2 for the
 * transposed forms (op 1, 2); the reduction channel count must be even for k = 3.  Anything else -> IPSR_ERR_UNSUPPORTED
 * (the Python wrapper then leaves that layer on MIOpen).  fp32, summation order fixed (deterministic, no atomics). */
size_t ipsr_conv2d_workspace_bytes(int op, int B, int Cin, int H, int W, int Cout, int k, int stride, int pad, int dil);
int ipsr_conv2d(int op, const float* in, const float* weight, float* out, int B, int Cin, int H, int W, int Cout,
                int k, int stride, int pad, int dil, void* ws, size_t ws_bytes, void* stream);

/* The 3x3 / stride 1 / pad 1 members of that family (every VGG16 convolution, `downconv_3` / `upconv_3` of every netG level)
 * by Winograd F(4x4,3x3): 4x fewer matrix-core flops than the direct form at 2.25x the activation traffic; pays from ~128
 * channels up.  Same `op` codes and tensor roles as ipsr_conv2d (k = 3, stride = 1, pad = 1, dil = 1 implied).  The
 * reduction channel count (Cin for op 0/2, Cout for op 1/3) must be a multiple of 16.  fp32; the transforms use the
 * standard points (0, +-1, +-2, inf): error ~4e-6 of the output scale at 512 channels. */
size_t ipsr_conv3x3_winograd_workspace_bytes(int op, int B, int Cin, int H, int W, int Cout);
int ipsr_conv3x3_winograd(int op, const float* in, const float* weight, float* out, int B, int Cin, int H, int W, int Cout,
                          void* ws, size_t ws_bytes, void* stream);

/* The same with an epilogue and a reusable filter transform — the VGG16 chain Conv2d(bias) -> ReLU(inplace) [-> MaxPool2d(2,2)]
 * (models/vgg16.py:9-21, run three times per step on FROZEN weights) in one pass over the activations:
 *   epilogue 0 none | 1 = out = relu(conv + bias[k]) | 2 = out = maxpool2x2(relu(conv + bias[k])), out [.,.,H/2,W/2] (H, W even)
 *   filter_cache (optional, ipsr_conv3x3_winograd_filter_floats(op, Cin, Cout) floats owned by the caller): the transformed
 *   filter is written there when filter_cache_valid == 0 and reused unchanged when != 0.  bias may be NULL. */
size_t ipsr_conv3x3_winograd_filter_floats(int op, int Cin, int Cout);
int ipsr_conv3x3_winograd_ex(int op, const float* in, const float* weight, const float* bias, int epilogue, float* filter_cache,
                             int filter_cache_valid, float* out, int B, int Cin, int H, int W, int Cout,
                             void* ws, size_t ws_bytes, void* stream);

/* Weight gradient of the same 3x3 / stride 1 / pad 1 layers by Winograd F(3x3,4x4) (the transposed algorithm: 4x4 tiles of one
 * operand against 6x6 windows of the other, reduced over all tiles of the batch on the matrix cores):
 *   transposed = 0  Conv2d           dw [Cout,Cin,3,3] = sum dy[.,co,o] * x[.,ci,o+r-1]
 *   transposed = 1  ConvTranspose2d  dw [Cin,Cout,3,3] = sum x[.,ci,i] * dy[.,co,i+r-1]
 * x [B,Cin,H,W], dy [B,Cout,H,W]; any channel counts (tiles are zero padded to 128). */
size_t ipsr_conv3x3_winograd_wrw_workspace_bytes(int transposed, int B, int Cin, int H, int W, int Cout);
int ipsr_conv3x3_winograd_wrw(int transposed, const float* x, const float* dy, float* dw, int B, int Cin, int H, int W, int Cout,
                              void* ws, size_t ws_bytes, void* stream);

/* The dilated down convolution of every netG level — Conv2d(k4, stride 2, pad 3, dilation 2), models/networks.py:226 — by
 * Winograd F(3x3,4x4): with dilation 2 and stride 2 the layer is a plain 4-tap stride-1 correlation on the odd-subsampled
 * input, so it runs through the same 36 GEMMs as the 3x3 layers (4x fewer matrix-core flops than the direct form).
 *   mode 0  forward        a = x  [B,Cin,H,W]       b = weight [Cout,Cin,4,4]   out = y  [B,Cout,H/2,W/2]
 *   mode 1  backward-data  a = dy [B,Cout,H/2,W/2]  b = weight                  out = dx [B,Cin,H,W] (even rows / columns are zero)
 *   mode 2  weight grad    a = x                    b = dy                      out = dw [Cout,Cin,4,4]
 * H, W even; the reduction channels (Cin for mode 0, Cout for mode 1) a multiple of 16. */
size_t ipsr_conv4x4_dilated_winograd_workspace_bytes(int mode, int B, int Cin, int H, int W, int Cout);
int ipsr_conv4x4_dilated_winograd(int mode, const float* a, const float* b, float* out, int B, int Cin, int H, int W, int Cout,
                                  void* ws, size_t ws_bytes, void* stream);

/* The same F(3x3,4x4) pipeline for either 4x4 geometry of the nets that is a 4-tap stride-1 correlation:
 *   geom 0  Conv2d(k4, stride 2, pad 3, dilation 2)  (== the two entry points above)
 *   geom 1  Conv2d(k4, stride 1, pad 1) — netD's fourth convolution 256 -> 512 on 32x32, models/networks.py:483-489 —
 *           y / dy are [B,Cout,H-1,W-1]; every position of dx is written.
 * Modes and operand roles as above. */
size_t ipsr_conv4x4_winograd_workspace_bytes(int geom, int mode, int B, int Cin, int H, int W, int Cout);
int ipsr_conv4x4_winograd(int geom, int mode, const float* a, const float* b, float* out, int B, int Cin, int H, int W, int Cout,
                          void* ws, size_t ws_bytes, void* stream);

/* The 4x4 stride-2 pad-1 layers — the down convolutions of netP / netD / netF (Conv2d, models/networks.py:404-432, 470-495,
 * 510-515) and the up convolutions of netP / netG (ConvTranspose2d, models/networks.py:235-243, 420-428) — by Winograd
 * F(5x5,2x2) on the polyphase components of the fine grid (2.8x fewer matrix-core flops than the direct form; same 36 GEMMs).
 * One geometry for both modules: fine = the 2n-grid tensor [B,Cf,2nh,2nw] (x of Conv2d, y of ConvTranspose2d), coarse = the
 * n-grid tensor [B,Kc,nh,nw]; weight = [Kc][Cf][4][4] (Conv2d [Cout][Cin], ConvTranspose2d [Cin][Cout]).
 *   mode 0  fine -> coarse   a = fine    b = weight   out = coarse   Conv2d forward / ConvTranspose2d backward-data; Cf % 4 == 0
 *   mode 1  coarse -> fine   a = coarse  b = weight   out = fine     ConvTranspose2d forward / Conv2d backward-data; Kc % 16 == 0
 *   mode 2  weight gradient  a = fine    b = coarse   out = dW [Kc][Cf][4][4]
 * Every element of `out` is written. */
size_t ipsr_conv4x4s2_winograd_workspace_bytes(int mode, int B, int Kc, int Cf, int nh, int nw);
int ipsr_conv4x4s2_winograd(int mode, const float* a, const float* b, float* out, int B, int Kc, int Cf, int nh, int nw,
                            void* ws, size_t ws_bytes, void* stream);

/* Small maps: the inner levels of the U-Nets and netF (512-1024 channels on 8x8 ... 1x1; models/networks.py:220-259, 404-432,
 * 510-515).  The weight tensor [R][Cq][k][k] (Conv2d: [Cout][Cin], ConvTranspose2d: [Cin][Cout]) is the GEMM operand as it lies
 * in memory; (k, stride, pad, dil) are the module's own parameters, Hf x Wf the grid on the weight's SECOND-channel side
 * (Conv2d input / ConvTranspose2d output), Ho x Wo the grid on its first-channel side.
 *   op 0  a = in [B,R,Ho,Wo]       b = weight        out = [B,Cq,Hf,Wf]    Conv2d backward-data / ConvTranspose2d forward
 *   op 1  a = coarse [B,R,Ho,Wo]   b = fine [B,Cq,Hf,Wf]   out = dW [R,Cq,k,k]   weight gradient of either module
 *   op 2  a = fine [B,Cq,Hf,Wf]    b = weight        out = [B,R,Ho,Wo]     Conv2d forward / ConvTranspose2d backward-data
 * R % 32 == 0, Cq*k*k % 128 == 0, B*Ho*Wo <= 1024. */
size_t ipsr_conv_smallmap_workspace_bytes(int op, int B, int R, int Cq, int Ho, int Wo, int Hf, int Wf, int k, int stride, int pad, int dil);
int ipsr_conv_smallmap(int op, const float* a, const float* b, float* out, int B, int R, int Cq, int Ho, int Wo, int Hf, int Wf,
                       int k, int stride, int pad, int dil, void* ws, size_t ws_bytes, void* stream);

/* The 3x3 stride-1 pad-1 layers with 3 or 6 channels on one side, at full resolution — VGG16 conv1_1 (models/vgg16.py:9), netG's
 * first Conv2d 6 -> 64 (models/networks.py:300-312) and its last ConvTranspose2d 128 -> 3 (:255-259): one pass over the wide
 * tensor on the vector ALUs (these are HBM streams, not matrix-core work).
 *   out[b][o][y][x] = sum_i sum_t W[o*so + i*si + (flip ? 8-t : t)] * in[b][i][y+r-1][x+s-1]      (t = 3r+s)
 *   op 0  few -> many: I in {3,6}, O % 16 == 0; optional bias [O] and ReLU (VGG)
 *   op 1  many -> few: O in {3,6}, W % 4 == 0, I*O*36 bytes <= 48 KB; bias must be NULL, relu 0
 * Conv2d forward: so = Cin*9, si = 9, flip 0; Conv2d backward-data: so = 9, si = Cin*9, flip 1 (o = ci, i = co);
 * ConvTranspose2d forward: so = 9, si = Cout*9, flip 1; ConvTranspose2d backward-data: so = Cout*9, si = 9, flip 0.
 * ipsr_conv3x3_thin_wrw: g[cb][cs][u][v] = sum_{b,y,x} big[b][cb][y][x] * small[b][cs][y+u-1][x+v-1] — the weight gradient of
 * either module with big = the wide tensor of (x, dy) and small = the other (Cs in {3,6}, Cb even, W % 4 == 0). */
int ipsr_conv3x3_thin(int op, const float* in, const float* w, const float* bias, int relu, float* out, int B, int I, int O, int H, int W,
                      long so, long si, int flip, void* stream);
size_t ipsr_conv3x3_thin_wrw_workspace_bytes(int B, int Cb, int Cs, int H, int W);
/* The same kernels on bf16 activation tensors (BASELINE config 5, the modules under torch.autocast(bfloat16)).  `io`: bit 0 = the first
 * tensor is bf16, bit 1 = the second (_io: in / out; _wrw_io: big / small); 0 = the fp32 entry points above.  With a bf16 side the
 * arithmetic is autocast's: weights and an fp32 operand are rounded to bf16 on the way in, products accumulate in fp32, bias / ReLU in
 * fp32, one rounding on the way out; the weight gradient stays fp32.  Alignment: four elements (few -> many output: two). */
int ipsr_conv3x3_thin_io(int op, const void* in, const float* w, const float* bias, int relu, void* out, int B, int I, int O, int H, int W,
                         long so, long si, int flip, int io, void* stream);
int ipsr_conv3x3_thin_wrw_io(const void* big, const void* small, float* g, int B, int Cb, int Cs, int H, int W, int io,
                             void* ws, size_t ws_bytes, void* stream);
/* few -> many on the bf16 matrix cores (BASELINE config 5): out[b][o][y][x] = sum_{cs,t} W[o*so + cs*si + (flip ? k*k-1-t : t)] *
 * in[b][cs][y*stride + r - 1][x*stride + s - 1] (+ bias, ReLU) — VGG16 conv1_1 (models/vgg16.py:9), netG's first Conv2d and the input
 * gradient of its last ConvTranspose2d (models/networks.py:255-259,300-312), the first Conv2d of netP / netD (:404-410,470-476).
 * Cs in {3, 6}; (k, stride) = (3, 1) or (4, 2), padding 1; Wo % 32 == 0, O % 8 == 0.  `io`: bit 0 = `in` bf16 (else fp32), bit 1 = `out`
 * bf16 (else fp32); operands are rounded to bf16, accumulation / bias / ReLU in fp32.  so / si / flip as in ipsr_conv3x3_thin. */
int ipsr_conv_thin_f2m_mfma_supported(int B, int Cs, int O, int Ho, int Wo, int k, int stride);
int ipsr_conv_thin_f2m_mfma(const void* in, const float* w, const float* bias, int relu, void* out, int B, int Cs, int O, int Ho, int Wo, int k, int stride,
                            long so, long si, int flip, int io, void* stream);
/* Weight gradient of the thin layers on the matrix cores:
 *   g[kb][cs][r][s] = sum_{b,y,x} big[b][kb][y][x] * small[b][cs][y*stride + r - 1][x*stride + s - 1]
 * big [B,Kb,Hb,Wb] = the wide tensor of the pair (x, dy) — dy of a Conv2d, x of a ConvTranspose2d — small [B,Cs,Hb*stride,Wb*stride],
 * Cs in {3, 6}; (k, stride) = (3, 1) or (4, 2), padding 1; Wb % 16 == 0.  `io`: bit 0 = big is bf16, bit 1 = small is bf16.  bf16 big:
 * v_mfma_f32_32x32x16_bf16, an fp32 small is rounded to bf16 (BASELINE config 5); fp32 big + fp32 small: v_mfma_f32_32x32x2_f32, the
 * reference's arithmetic (config 2).  g fp32 in the module's own weight layout: Conv2d [Cout=Kb][Cin=Cs][k][k]
 * (models/networks.py:300-312,404-410,470-476), ConvTranspose2d [Cin=Kb][Cout=Cs][k][k] (:255-259,424-432).  Fixed summation order.
 * 0 workspace bytes = not implemented. */
size_t ipsr_conv_thin_wrw_mfma_workspace_bytes(int B, int Kb, int Cs, int Hb, int Wb, int k, int stride);
int ipsr_conv_thin_wrw_mfma(const void* big, const void* small, float* g, int B, int Kb, int Cs, int Hb, int Wb, int k, int stride, int io,
                            void* ws, size_t ws_bytes, void* stream);
/* ipsr_conv_to_one: nn.Conv2d(C, 1, K, stride 1, padding pad) — netD's last layer (models/networks.py:489-495, 512 -> 1, k4 p1 on
 * 31x31) — as one pass over the input (252 MFLOP against 31.5 MB: a stream).  x [B,C,H,W] fp32, K in {3, 4}.
 *   op 0 forward:          other = w [1,C,K,K],    out = y [B,1,Ho,Wo],  Ho = H + 2 pad - K + 1
 *   op 2 weight gradient:  other = dy [B,1,Ho,Wo], out = dw [1,C,K,K]
 * Fixed summation orders (deterministic).  The input gradient (1 -> C) is left to the caller's library.  Planes of at most 256
 * groups of 4 output pixels (Ho * ceil(Wo / 4) <= 256); ipsr_conv_to_one_workspace_bytes returns 0 for anything else.  The forward
 * needs that workspace (per-chunk partial sums), the weight gradient none. */
size_t ipsr_conv_to_one_workspace_bytes(int B, int C, int H, int W, int K, int pad);
int ipsr_conv_to_one(int op, const float* x, const float* other, float* out, int B, int C, int H, int W, int K, int pad,
                     void* ws, size_t ws_bytes, void* stream);
int ipsr_conv3x3_thin_wrw(const float* big, const float* small, float* g, int B, int Cb, int Cs, int H, int W, void* ws, size_t ws_bytes, void* stream);

/* ---- K9  InnerCos / InnerCos2 feature-consistency loss ----------------------------------------
 * replaces InnerCos.forward (models/InnerCos.py:30-41) and InnerCos2.forward
 * (models/InnerCos2.py:34-46):  loss = mean_{b,c<Cuse,n} ((x[b,c,n]*mask[n])*strength - target)^2.
 * x has Cx >= Cuse channels per sample (InnerCos2 reads the first 512 of 1024, InnerCos2.py:38);
 * target is [B,Cuse,N].  loss is one fp32 on the device.  The backward entry point returns
 * d loss / d x (only for the first Cuse channels; the rest is zero-filled). */
size_t innercos_workspace_bytes(int B, int Cuse, int N);
int innercos_loss(const float* x, int B, int Cx, int Cuse, int N, const float* mask /*[N] fp32*/,
                  const float* target, float strength, float* loss /*[1]*/,
                  void* ws, size_t ws_bytes, void* stream);
/* the same in ONE launch: the last workgroup to finish adds the block partials (ascending order) and writes the loss.  `ticket`: one
 * 32-bit word of caller memory that is zero on entry and left zero (the library keeps no device state); calls sharing a word must
 * be ordered by their stream. */
int innercos_loss_fused(const float* x, int B, int Cx, int Cuse, int N, const float* mask /*[N] fp32*/,
                        const float* target, float strength, float* loss /*[1]*/,
                        void* ws, size_t ws_bytes, unsigned* ticket, void* stream);
int innercos_loss_backward(const float* x, int B, int Cx, int Cuse, int N, const float* mask,
                           const float* target, float strength, const float* grad_loss /*[1]*/,
                           float* grad_x /*[B,Cx,N]*/, void* stream);

/* ---- measurement hook (bench.py) -----------------------------------------------------------------
 * Opt-in: when enabled, three kinds of regions are bracketed by a pair of HIP events recorded on the SAME stream the
 * work is launched on:
 *   region 0  every launch of the correlation + arg-max kernel (the layer's dominant kernel, inside ipsr_forward /
 *             ipsr_corr_argmax)
 *   region 1  every whole ipsr_forward call (all its kernels)
 *   region 2  every whole ipsr_backward / ipsr_backward_patch call
 *   region 3  every launch of the Winograd GEMM kernel (see ipsr_profile_read_region_work)
 *   region 4  every launch of the direct bf16 convolution kernels (ipsr_conv3x3_bf16 / ipsr_conv4x4s2_bf16 / their weight gradient)
 *   region 5  every launch of the InnerCos loss kernel (innercos_loss_fused; work = algorithmic bytes)
 * so that a training step can report the layer's time ON ITS REAL INPUTS.  ipsr_profile_read_region synchronises the
 * recorded pairs of one region, writes their elapsed times (ms) to the HOST array `ms` and resets that ring; it returns
 * the number written.  ipsr_profile_read(ms, n) == ipsr_profile_read_region(0, ms, n).  ipsr_profile_enable(capacity)
 * sizes every ring for `capacity` regions; ipsr_profile_enable(0) disables and frees.  This is the only global state in
 * the library and it is off by default; enable/read are not re-entrant (call them from one thread, outside a timed step). */
int ipsr_profile_enable(int capacity);                              /* regions 0..2 */
int ipsr_profile_enable_mask(int capacity, unsigned region_mask);   /* bit r = region r (region 3 is off unless asked for) */
int ipsr_profile_read(float* ms /*[host]*/, int max_n);
int ipsr_profile_read_region(int region, float* ms /*[host]*/, int max_n);
/* region 3 = every launch of the Winograd GEMM kernel (csrc/winograd.hip, the convolutions' matrix-core kernel; its ring holds
 * 256 x capacity launches); `work` receives the flop count (2 x 36 x rows x columns x reduction, padded sizes) of each launch. */
int ipsr_profile_read_region_work(int region, float* ms /*[host]*/, double* work /*[host]*/, int max_n);
/* the same with `useful` = the flops of the UNPADDED problem of each launch (produced channels / tiles / reduction before they are
 * rounded up to the 128 x 128 x 16 tile): work - useful is arithmetic on zero padding */
int ipsr_profile_read_region_work2(int region, float* ms /*[host]*/, double* work /*[host]*/, double* useful /*[host]*/, int max_n);

/* ---- mixed-precision forms of the Winograd convolutions (BASELINE config 5: "bf16 mixed precision (CDNA4 bf16 MFMA for patch-corr
 * + convs)"; also an opt-in arithmetic for the fp32 nets) -------------------------------------------------------------------------
 * Same operations, arguments and workspace queries as ipsr_conv3x3_winograd_ex / ipsr_conv3x3_winograd_wrw / ipsr_conv4x4_winograd /
 * ipsr_conv4x4s2_winograd above (models/networks.py:220-259, 404-432, 470-495, 510-515; models/vgg16.py:9-21), plus:
 *   math  0  fp32 operands on v_mfma_f32_32x32x2_f32 — the reference's arithmetic, what the entry points above run;
 *         2  transformed operands SPLIT into two bf16 numbers (hi + lo, sum exact to 2^-16) and multiplied as hi*hi + hi*lo + lo*hi on
 *            v_mfma_f32_32x32x16_bf16 with fp32 accumulation: 3 MFMAs of 32 cycles per 16 channels instead of 8 of 64; error
 *            ~1e-4 of the output scale (fp32 path: ~1e-5; a plain bf16 convolution: ~2e-3);
 *         3  split into three bf16 numbers (exact to 2^-24), six products: the fp32 path's accuracy (measured equal) at ~0.6x its time.
 *         Operands merely ROUNDED to bf16 are not offered: F(4x4,3x3) amplifies rounding ~100x (3-6 % error).
 *   io    bit 0: the activation tensors READ (x / dy; both operands of a weight gradient) are bf16; bit 1: the activation tensor
 *         WRITTEN is bf16.  Weights, biases, weight gradients, transforms and accumulation are always fp32. */
int ipsr_conv3x3_winograd_mp(int op, const void* in, const float* weight, const float* bias, int epilogue, float* filter_cache,
                             int filter_cache_valid, void* out, int B, int Cin, int H, int W, int Cout, int math, int io,
                             void* ws, size_t ws_bytes, void* stream);
int ipsr_conv3x3_winograd_wrw_mp(int transposed, const void* x, const void* dy, float* dw, int B, int Cin, int H, int W, int Cout,
                                 int math, int io, void* ws, size_t ws_bytes, void* stream);
int ipsr_conv4x4_winograd_mp(int geom, int mode, const void* a, const void* b, void* out, int B, int Cin, int H, int W, int Cout,
                             int math, int io, void* ws, size_t ws_bytes, void* stream);
int ipsr_conv4x4s2_winograd_mp(int mode, const void* a, const void* b, void* out, int B, int Kc, int Cf, int nh, int nw,
                               int math, int io, void* ws, size_t ws_bytes, void* stream);

/* ---- direct bf16 convolutions (BASELINE config 5: "CDNA4 bf16 MFMA for ... convs") --------------------------------------------
 * replaces nn.Conv2d / nn.ConvTranspose2d(k3 s1 p1) under bf16 autocast — models/networks.py:220-243 (`downconv_3` / `upconv_3` of
 * every netG level), models/vgg16.py:9-21 — and their input gradients: ONE implicit-GEMM launch on v_mfma_f32_32x32x16_bf16 (bf16
 * operands, fp32 accumulation), NCHW bf16 activations in, NCHW bf16 (out_bf16 = 1) or fp32 (0) out, fp32 weights cast inside.
 * op as in ipsr_conv2d (0 Conv2d forward, 1 Conv2d backward-data, 2 ConvTranspose2d forward, 3 ConvTranspose2d backward-data);
 * (Cin, H, W) describe the module's input.  Supported: W in {16, 32, 64, 128, 256}, H a multiple of 256 / W, reduction channels a
 * multiple of 16; anything else -> IPSR_ERR_UNSUPPORTED.  Not bit-comparable with anything: operands are rounded to bf16 (tests
 * compare with an fp64 convolution of the bf16-rounded operands). */
size_t ipsr_conv3x3_bf16_workspace_bytes(int op, int B, int Cin, int H, int W, int Cout);
int ipsr_conv3x3_bf16(int op, const void* in, const float* weight, void* out, int B, int Cin, int H, int W, int Cout, int out_bf16,
                      void* ws, size_t ws_bytes, void* stream);
/* _packed: `ws` doubles as the caller-owned cache of the re-packed bf16 weights (same size as the workspace query): pack_valid = 0
 * packs into it, pack_valid = 1 reuses what an earlier call on the SAME weights, op and channel counts left there (frozen VGG16). */
int ipsr_conv3x3_bf16_packed(int op, const void* in, const float* weight, void* out, int B, int Cin, int H, int W, int Cout, int out_bf16,
                             int pack_valid, void* ws, size_t ws_bytes, void* stream);
/* the 4x4 stride-2 pad-1 layers (every down convolution of netP / netD / netF, every up convolution of netP / netG: models/networks.py:
 * 235-243, 404-432, 470-495, 510-515) in the coarse / fine terms of ipsr_conv4x4s2_winograd: fine = the 2n-grid tensor [B,Cf,2nh,2nw],
 * coarse = the n-grid one [B,Kc,nh,nw], weight [Kc][Cf][4][4] for both modules.  mode 0: fine -> coarse (Conv2d forward,
 * ConvTranspose2d input gradient); mode 1: coarse -> fine (ConvTranspose2d forward, Conv2d input gradient).  Supported: nw in
 * {16, 32, 64} (mode 1 also 128), nh a multiple of 256 / nw, reduction channels a multiple of 16. */
size_t ipsr_conv4x4s2_bf16_workspace_bytes(int mode, int B, int Kc, int Cf, int nh, int nw);
int ipsr_conv4x4s2_bf16(int mode, const void* in, const float* weight, void* out, int B, int Kc, int Cf, int nh, int nw, int out_bf16,
                        void* ws, size_t ws_bytes, void* stream);
/* weight gradient of the 4x4 stride-2 layers: fine [B,Cf,2nh,2nw], coarse [B,Kc,nh,nw] bf16 -> dw [Kc][Cf][4][4] fp32 (both modules' layout);
 * nw in {16, 32, 64}, nh a multiple of 64 / nw.  Partial sums added in a fixed order by a second launch. */
size_t ipsr_conv4x4s2_bf16_wrw_workspace_bytes(int B, int Kc, int Cf, int nh, int nw);
int ipsr_conv4x4s2_bf16_wrw(const void* fine, const void* coarse, float* dw, int B, int Kc, int Cf, int nh, int nw,
                            void* ws, size_t ws_bytes, void* stream);
/* their weight gradient: x [B,Cin,H,W], dy [B,Cout,H,W] bf16 -> dw fp32 in the module's layout (transposed = 0: Conv2d [Cout][Cin][3][3];
 * 1: ConvTranspose2d [Cin][Cout][3][3]).  The reduction over pixels is cut over workgroups; the partial results are added in a fixed
 * order by a second launch (deterministic).  Same shape limits as above. */
size_t ipsr_conv3x3_bf16_wrw_workspace_bytes(int transposed, int B, int Cin, int H, int W, int Cout);
int ipsr_conv3x3_bf16_wrw(int transposed, const void* x, const void* dy, float* dw, int B, int Cin, int H, int W, int Cout,
                          void* ws, size_t ws_bytes, void* stream);

/* How the reduction of the 36 Winograd GEMMs of a layer is cut over workgroups (csrc/winograd.hip, wino_choose_split): for a GEMM
 * of `rows` x `cols` (multiples of 128: produced channels x tiles, padded) with `reduction` (multiple of 16) terms,
 * out5 = {nsplit, stages per range, xi_split, nsplit_tail, stages per tail range}: the GEMMs of points xi < xi_split run in
 * `nsplit` ranges, the others in `nsplit_tail`.  Pure function; the parity tests use it to prove that both the uniform and the
 * head / tail cut are exercised.  ipsr_debug_force_wino_split(nsplit, xi_split, nsplit_tail) overrides the rule for every
 * following call in this process (tuning aid of tools/sweep_wino_split.py; (0,0,0) restores the rule) — with the measurement
 * hook above the only global state of the library, and like it off by default. */
int ipsr_wino_gemm_split(int rows, int cols, int reduction, int* out5 /*[host]*/);
int ipsr_debug_force_wino_split(int nsplit, int xi_split, int nsplit_tail);
/* A/B switches for kernel variants (tools/exp_*.py): key in [0,16), value 0 = the shipped behaviour.  Same status as the
 * override above: process-global, off by default, never set by the product path. */
int ipsr_debug_set_option(int key, int value);

#ifdef __cplusplus
}
#endif
#endif /* IPSR_HIP_H */
