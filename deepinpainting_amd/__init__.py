"""deepinpainting_amd — MI355X (gfx950) implementation of the IPSR patch-attention training hot path of
Image-Processing-Systems-Laboratory/DeepInPainting behind the reference's own Python surface.

    from deepinpainting_amd.models.models import create_model      # reference: models/models.py:2-12
    from deepinpainting_amd.models.IPSRFunction import IPSRFunction # reference: models/IPSRFunction.py

The compute path is libipsr_hip.so (hand-written HIP, C-ABI in include/ipsr_hip.h).  There is no CPU or
eager-PyTorch fallback for the layer: if the library is missing the ops raise.
"""
__version__ = "0.1.0"

import os as _os

# Conv solver selections measured once on an MI355X for the step's shapes (see miopen_db/README.md).  MIOpen reads the
# variable when its first handle is created, i.e. at the first convolution — so setting it at import time is early enough.
if _os.environ.get("IPSR_NO_MIOPEN_DB", "0") != "1":
    _os.environ.setdefault("MIOPEN_USER_DB_PATH", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "miopen_db"))
