"""Host-side mirrors the reference's drivers import next to the hot path (train.ipynb / test.ipynb / app.py): datasets,
EarlyStopping, image helpers, input staging.  CPU only."""
import os
import random

import numpy as np
import pytest
import torch
from PIL import Image


def _write_images(root, n, ext, size=(40, 30), seed=0):
    os.makedirs(root, exist_ok=True)
    rs = np.random.RandomState(seed)
    for i in range(n):
        arr = rs.randint(0, 255, (size[1], size[0], 3), dtype=np.uint8)
        Image.fromarray(arr).save(os.path.join(root, "im_%02d.%s" % (i, ext)))


def _to_tensor(im):
    a = np.asarray(im.resize((16, 16)), dtype=np.float32) / 255.0
    return torch.from_numpy(a.transpose(2, 0, 1))


def test_data_load_triples(tmp_path):
    from deepinpainting_amd.util.data_load import Data_load
    from deepinpainting_amd.util.ref_data_load import Ref_Data_load
    _write_images(tmp_path / "img", 5, "jpg", seed=1)
    _write_images(tmp_path / "ref", 5, "jpg", seed=2)
    _write_images(tmp_path / "mask", 3, "png", seed=3)
    for cls in (Data_load, Ref_Data_load):
        ds = cls(str(tmp_path / "img"), str(tmp_path / "mask"), str(tmp_path / "ref"), _to_tensor, _to_tensor, _to_tensor)
        assert len(ds) == 5 and ds.N_mask == 3
        random.seed(0)
        image, mask, ref = ds[2]
        assert image.shape == mask.shape == ref.shape == (3, 16, 16)
        # image i is paired with reference image i of the (sorted) listings
        assert os.path.basename(ds.paths[2]) == os.path.basename(ds.ref_paths[2]) == "im_02.jpg"
        batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False)))
        assert batch[0].shape == (2, 3, 16, 16)


def test_early_stopping_semantics():
    from deepinpainting_amd.models.Early import EarlyStopping
    e = EarlyStopping(3)
    for v in (5.0, 4.0, 4.0):          # an equal loss counts as an improvement (reference models/Early.py:15-21)
        e(v)
    assert e.counter == 0 and not e.early_stop and e.best_score == -4.0
    e(4.5); e(4.2)
    assert e.counter == 2 and not e.early_stop
    e(3.9)
    assert e.counter == 0
    for v in (4.0, 4.1, 4.2):
        e(v)
    assert e.early_stop


def test_util_image_helpers(tmp_path):
    from deepinpainting_amd.util import util
    t = torch.linspace(-1, 1, 3 * 4 * 5).view(1, 3, 4, 5)
    im = util.tensor2im(t)
    assert im.shape == (4, 5, 3) and im.dtype == np.uint8 and im.min() == 0 and im.max() == 255
    assert util.tensor2im(t[:, :1]).shape == (4, 5, 3)
    util.save_image(im, str(tmp_path / "x.png"))
    assert np.array_equal(np.asarray(Image.open(tmp_path / "x.png")), im)
    pattern = torch.zeros(64, 64)
    pattern[:, :20] = 1                 # any 32x32 window starting left of column ~13 is 20-40 % masked
    random.seed(1)
    m = util.create_gMask({"pattern": pattern, "mask_global": torch.zeros(1, 1, 32, 32), "MAX_SIZE": 64, "fineSize": 32,
                           "maxPartition": 45})
    assert m.shape == (1, 1, 32, 32) and 20 < float(m.sum()) * 100 / 1024 < 45
    assert util.binary_mask(torch.tensor([[0.2, 0.8]]), 0.5).tolist() == [[0.0, 1.0]]
    net = torch.nn.Linear(3, 2)
    net(torch.ones(1, 3)).sum().backward()
    assert util.diagnose_network(net) > 0


def test_staging_cpu_and_stroke_masks():
    from deepinpainting_amd.util.staging import DeviceStager, prepare_mask, random_stroke_mask
    batches = [(torch.full((2, 3, 8, 8), float(i)), (torch.rand(2, 3, 8, 8) > 0.5).float(), torch.zeros(2, 3, 8, 8)) for i in range(3)]
    got = list(DeviceStager(batches, "cpu"))
    assert len(got) == 3
    for i, (image, mask, ref) in enumerate(got):
        assert float(image[0, 0, 0, 0]) == float(i)
        assert mask.shape == (1, 1, 8, 8) and mask.dtype == torch.bool
        assert torch.equal(mask[0, 0], batches[i][1][0][0] != 0)           # train.ipynb cell 2: mask[0][0]
    g1, g2 = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
    a, b = random_stroke_mask(128, g1), random_stroke_mask(128, g2)
    assert torch.equal(a, b) and a.shape == (1, 1, 128, 128) and 0.2 < float(a.float().mean()) < 0.4
    assert prepare_mask(torch.ones(4, 4)).shape == (1, 1, 4, 4)


@pytest.mark.gpu
def test_staging_on_gpu_feeds_the_trainer_inputs():
    from deepinpainting_amd.util.staging import DeviceStager, random_stroke_mask
    batches = [(torch.rand(2, 3, 32, 32) * 2 - 1, random_stroke_mask(32, torch.Generator().manual_seed(i), width=(3, 8)).float().expand(2, 3, 32, 32),
                torch.rand(2, 3, 32, 32)) for i in range(4)]
    for i, (image, mask, ref) in enumerate(DeviceStager(batches, "cuda:0")):
        assert image.is_cuda and mask.is_cuda and ref.is_cuda and mask.dtype == torch.bool
        assert torch.equal(image.cpu(), batches[i][0]) and torch.equal(mask.cpu()[0, 0], batches[i][1][0][0] != 0)
    m = random_stroke_mask(256, torch.Generator().manual_seed(1), device="cuda")
    ref = random_stroke_mask(256, torch.Generator().manual_seed(1))
    assert m.is_cuda and float((m.cpu() != ref).float().mean()) < 1e-3       # same strokes (edge pixels may round differently)
