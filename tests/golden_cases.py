"""Seeded regeneration of the inputs of the large golden case (kept out of the fixture to keep it small).

Shared by oracle/gen_golden.py (which produced the expected outputs by running the reference) and by the
tests.  np.random.RandomState is the legacy, bit-stable generator.
"""
import zlib

import numpy as np


def grad_seed(name):
    return zlib.crc32(name.encode()) % (2 ** 31)


def big_case_inputs():
    """BASELINE.json config 2, one sample: x = |N(0,1)|, ref ~ U(0,1), [1,512,32,32]."""
    rs = np.random.RandomState(2024)
    x = np.abs(rs.standard_normal((1, 512, 32, 32))).astype(np.float32)
    ref = rs.rand(1, 512, 32, 32).astype(np.float32)
    return x, ref


def big_case_grad_out(name="layer_c512_32x32_cfg2"):
    return np.random.RandomState(grad_seed(name)).standard_normal((1, 512, 32, 32)).astype(np.float32)


def big_case_ic_target():
    return np.random.RandomState(7).rand(1, 512, 32, 32).astype(np.float32)


def reinit_deterministic(module, seed):
    """Overwrite every parameter of `module`, in registration order, from one seeded CPU generator.
    Used on BOTH sides (the reference's nets in oracle/gen_golden.py, ours in the tests) so the two start
    from identical weights without depending on how many random numbers their constructors consumed."""
    import torch
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for _, p in module.named_parameters():
            r = torch.randn(p.shape, generator=gen)
            if p.dim() > 1:
                p.copy_((r * 0.02).to(p.device))
            else:
                p.copy_((0.5 + 0.02 * r).to(p.device))      # norm scales / biases: away from 0 so they matter
    return module


def trainer_inputs(B=1, size=256):
    """Seeded (image, mask, ref) for the trainer-level fixture: image/ref ~ U(-1,1), 128x128 centre hole."""
    import torch
    gen = torch.Generator().manual_seed(4321)
    img = torch.rand(B, 3, size, size, generator=gen) * 2 - 1
    ref = torch.rand(B, 3, size, size, generator=gen) * 2 - 1
    mask = torch.zeros(1, 1, size, size, dtype=torch.bool)
    mask[:, :, size // 4:size * 3 // 4, size // 4:size * 3 // 4] = 1
    return img, mask, ref


def net_input(shape, seed):
    import torch
    return torch.rand(*shape, generator=torch.Generator().manual_seed(seed)) * 2 - 1
