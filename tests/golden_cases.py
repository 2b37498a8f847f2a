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
