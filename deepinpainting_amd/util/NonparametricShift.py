"""Host-side mirror of the reference's util/NonparametricShift.py on the HIP kernels.

The reference wraps the patches into throw-away nn.Conv2d / nn.ConvTranspose2d modules per sample per
step (:43-55).  Here the "encoder" is the fused correlation+arg-max kernel and the "decoder" is the
reconstruction kernel; this class keeps the reference's entry points for callers that use it directly.
This stand-alone helper covers patch_size == 1, stride == 1 (what the reference can actually run,
models/IPSRFunction.py:134); the layer itself (IPSR_model / IPSRFunction -> ipsr_forward) also takes shift_sz > 1.
"""
import torch

from .. import ops


class PatchEncoder(object):
    """Stands in for `conv_enc` (NonparametricShift.py:43-45): call it on the reference feature
    [1,C,h,w] to get the correlation map [1,N,h,w] (materialised — test/inspection use only; the layer
    itself never writes S)."""

    def __init__(self, xn_1cn, h, w):
        self.xn, self.h, self.w = xn_1cn, h, w

    def __call__(self, ref_1chw):
        _, C, h, w = ref_1chw.shape
        _, _, S = ops.corr_argmax(self.xn, ref_1chw.reshape(1, C, h * w), want_S=True)
        return S.reshape(1, h * w, h, w)

    def argmax(self, ref_1chw):
        _, C, h, w = ref_1chw.shape
        ind, vmax, _ = ops.corr_argmax(self.xn, ref_1chw.reshape(1, C, h * w))
        return ind[0].to(torch.int64), vmax[0]


class NonparametricShift(object):
    def buildAutoencoder(self, target_img, normalize, interpolate, nonmask_point_idx, mask_point_idx,
                         patch_size=1, stride=1):
        """reference :10-33.  Returns (conv_enc_all, conv_enc_non_mask, conv_dec_all, conv_dec_non_mask,
        patches_part, patches_mask); the two decoders are returned as the raw patch tensors they would
        have wrapped."""
        assert target_img.dim() == 3, 'target image must be of dimension 3.'
        if normalize or interpolate:
            raise NotImplementedError
        if patch_size != 1 or stride != 1:
            raise NotImplementedError("this helper implements patch_size=1, stride=1 only; use IPSR_model for shift_sz > 1")
        C, h, w = target_img.shape
        patches_all, patches_part, patches_mask = self._extract_patches(target_img, patch_size, stride,
                                                                       nonmask_point_idx, mask_point_idx)
        xn, _ = ops.patch_normalize(target_img.reshape(1, C, h * w))
        enc = PatchEncoder(xn, h, w)
        return enc, enc, patches_all, patches_part, patches_part, patches_mask

    def _extract_patches(self, img, patch_size, stride, nonmask_point_idx, mask_point_idx):
        """reference :59-73: [N,C,1,1] patches (all, non-masked == all, masked)."""
        assert img.dim() == 3, 'image must be of dimension 3.'
        C, h, w = img.shape
        patches_all = img.reshape(C, h * w).t().contiguous().view(h * w, C, 1, 1)
        return (patches_all, patches_all.index_select(0, nonmask_point_idx.to(img.device)),
                patches_all.index_select(0, mask_point_idx.to(img.device)))
