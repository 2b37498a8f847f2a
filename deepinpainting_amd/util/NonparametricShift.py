"""Host-side mirror of the reference's util/NonparametricShift.py for callers that use it directly.

The layer itself (IPSR_model / IPSRFunction -> ipsr_forward) never goes through this class: it keeps the patches in
HBM once and runs the fused correlation + arg-max / reconstruction kernels.  A foreign caller, though, does what
models/IPSRFunction.py:54-131 does —

    _, conv_enc, conv_new_dec, _, known_patch, unknown_patch = NonparametricShift().buildAutoencoder(x[0], False, False, ...)
    tmp1 = conv_enc(ref)                  # [1,N,h',w'] correlation map
    ...
    out = conv_new_dec(kbar)              # [1,C,h,w]

— so `buildAutoencoder` returns what the reference returns (:43-55): `nn.Conv2d` / `nn.ConvTranspose2d` modules without
bias whose weights are the L2-normalised / raw patches.  The normalisation is the HIP kernel of the layer
(ops.patch_normalize, bit-identical to the layer's own `xn`); the encoder's forward runs the layer's MFMA correlation
kernel (S materialised) when the patches are 1x1 on a CUDA tensor, anything else is the module's ordinary forward.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


class PatchEncoder(nn.Conv2d):
    """`conv_enc` (NonparametricShift.py:43-45): Conv2d(C, npatches, patch_size, stride, bias=False) with the normalised
    patches as its weight.  For 1x1 patches on the GPU the correlation map comes from the layer's own kernel."""

    def __init__(self, enc_patches, stride):
        n, c, p, _ = enc_patches.shape
        super(PatchEncoder, self).__init__(c, n, kernel_size=p, stride=stride, bias=False)
        self.weight.data = enc_patches
        # a constant of the forward pass, as in the reference (which only ever reads conv_enc(...).data, IPSRFunction.py:66)
        self.weight.requires_grad_(False)

    def forward(self, ref_1chw):
        w = self.weight
        if ref_1chw.is_cuda and w.is_cuda and w.size(2) == 1 and self.stride == (1, 1) and ref_1chw.size(0) == 1 \
                and ref_1chw.dtype == torch.float32 and not (torch.is_grad_enabled() and (ref_1chw.requires_grad or w.requires_grad)) \
                and ref_1chw.size(2) * ref_1chw.size(3) == w.size(0):
            _, C, h, w_ = ref_1chw.shape
            xn = w.detach().reshape(w.size(0), C).t().contiguous()[None]                       # [1,C,N] channel-major
            _, _, S = ops.corr_argmax(xn, ref_1chw.reshape(1, C, h * w_), want_S=True)
            return S.reshape(1, h * w_, h, w_)
        return super(PatchEncoder, self).forward(ref_1chw)


class NonparametricShift(object):
    def buildAutoencoder(self, target_img, normalize, interpolate, nonmask_point_idx, mask_point_idx,
                         patch_size=1, stride=1):
        """reference :10-33 -> (conv_enc_all, conv_enc_non_mask, conv_dec_all, conv_dec_non_mask, patches_part, patches_mask)."""
        assert target_img.dim() == 3, 'target image must be of dimension 3.'
        C = target_img.size(0)
        patches_all, patches_part, patches_mask = self._extract_patches(target_img, patch_size, stride,
                                                                       nonmask_point_idx, mask_point_idx)
        conv_enc_non_mask, conv_dec_non_mask = self._build(patch_size, stride, C, patches_part, patches_part.size(0),
                                                           normalize, interpolate)
        if patches_part.size(0) == patches_all.size(0) and torch.equal(nonmask_point_idx.to(patches_all.device).long().view(-1),
                                                                        torch.arange(patches_all.size(0), device=patches_all.device)):
            # the reference's nonmask_point_idx is arange(N) (util/util.py:134-138): both builds are the same patches
            conv_enc_all, conv_dec_all = conv_enc_non_mask, conv_dec_non_mask
        else:
            conv_enc_all, conv_dec_all = self._build(patch_size, stride, C, patches_all, patches_all.size(0), normalize, interpolate)
        return conv_enc_all, conv_enc_non_mask, conv_dec_all, conv_dec_non_mask, patches_part, patches_mask

    def _build(self, patch_size, stride, C, target_patches, npatches, normalize, interpolate):
        """reference :36-57: (Conv2d with the patches / (||patch||_2 + 1e-8), ConvTranspose2d with the raw patches)."""
        if normalize or interpolate:
            raise NotImplementedError
        n, c, p, _ = target_patches.shape
        if target_patches.is_cuda and target_patches.dtype == torch.float32:
            flat = target_patches.reshape(n, c * p * p).t().contiguous()[None]                   # [1, C*p*p, n]
            xn, _ = ops.patch_normalize(flat)                                                   # K3, the layer's kernel
            enc_patches = xn[0].t().contiguous().view(n, c, p, p)
        else:
            nrm = target_patches.reshape(n, -1).norm(2, dim=1).view(n, 1, 1, 1)
            enc_patches = target_patches * (1 / (nrm + 1e-8))
        conv_enc = PatchEncoder(enc_patches, stride)
        conv_dec = nn.ConvTranspose2d(npatches, C, kernel_size=patch_size, stride=stride, bias=False)
        conv_dec.weight.data = target_patches
        return conv_enc, conv_dec

    def _extract_patches(self, img, patch_size, stride, nonmask_point_idx, mask_point_idx):
        """reference :59-73: all p x p windows in raster order as [N',C,p,p], and the two index_selects of them."""
        assert img.dim() == 3, 'image must be of dimension 3.'
        C = img.size(0)
        cols = F.unfold(img[None], kernel_size=patch_size, stride=stride)[0]                     # [C*p*p, N']
        patches_all = cols.t().contiguous().view(-1, C, patch_size, patch_size)
        return (patches_all, patches_all.index_select(0, nonmask_point_idx.to(img.device)),
                patches_all.index_select(0, mask_point_idx.to(img.device)))

    def _extract_patches_mask(self, img, patch_size, stride, nonmask_point_idx, mask_point_idx):
        """reference :75-86: only the masked windows."""
        return self._extract_patches(img, patch_size, stride, nonmask_point_idx, mask_point_idx)[2]
