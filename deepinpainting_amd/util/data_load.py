"""Training / test dataset of the reference's drivers — mirror of util/data_load.py (host-side I/O).

    Data_load(img_root, mask_root, ref_root, img_transform, mask_transform, ref_transform)[i] -> (image, mask, ref)

Same constructor, same triple (train.ipynb cell 1, test.ipynb cell 1, app.py:64).  Like the reference it pairs image i
with reference image i by their positions in the two `*.jpg` listings (:15-17,22-30) — the listings are sorted here so
that the pairing does not depend on the file system's enumeration order — and draws a random `*.png` mask per item (:25).
"""
import random
from glob import glob

import torch
from PIL import Image


def _listing(root, pattern):
    return sorted(glob('%s/%s' % (root, pattern), recursive=False))


class Data_load(torch.utils.data.Dataset):
    def __init__(self, img_root, mask_root, ref_root, img_transform, mask_transform, ref_transform):
        super(Data_load, self).__init__()
        self.img_transform = img_transform
        self.mask_transform = mask_transform
        self.ref_transform = ref_transform
        self.paths = _listing(img_root, '*.jpg')
        self.ref_paths = _listing(ref_root, '*.jpg')
        self.mask_paths = _listing(mask_root, '*.png')
        self.N_mask = len(self.mask_paths)

    def _rgb(self, path, transform):
        with Image.open(path) as im:
            return transform(im.convert('RGB'))

    def __getitem__(self, index):
        gt_img = self._rgb(self.paths[index], self.img_transform)
        mask = self._rgb(self.mask_paths[random.randint(0, self.N_mask - 1)], self.mask_transform)
        ref = self._rgb(self.ref_paths[index], self.ref_transform)
        return gt_img, mask, ref

    def __len__(self):
        return len(self.paths)
