"""deepinpainting_amd — MI355X (gfx950) implementation of the IPSR patch-attention training hot path of
Image-Processing-Systems-Laboratory/DeepInPainting behind the reference's own Python surface.

    from deepinpainting_amd.models.models import create_model      # reference: models/models.py:2-12
    from deepinpainting_amd.models.IPSRFunction import IPSRFunction # reference: models/IPSRFunction.py

The compute path is libipsr_hip.so (hand-written HIP, C-ABI in include/ipsr_hip.h).  There is no CPU or
eager-PyTorch fallback for the layer: if the library is missing the ops raise.
"""
__version__ = "0.1.0"

import os as _os


def use_shipped_miopen_db(local_rank=None):
    """Point MIOpen at a PRIVATE, writable copy of the conv solver selections shipped in `miopen_db/` (measured once on an
    MI355X for the step's shapes, see miopen_db/README.md).  MIOpen takes file locks on its user db and appends to it, so
    the shipped files are treated as read-only seeds: they are copied to a per-user, per-rank temp directory (8 ranks sharing
    one directory would serialise on the locks; a site-packages directory may not be writable at all).  Must run before the
    first convolution (MIOpen reads the variable when its first handle is created).  Importing this package does NOT touch
    the environment; `create_model` and `bench.py` call this.  No-op when MIOPEN_USER_DB_PATH is already set or
    IPSR_NO_MIOPEN_DB=1.  Returns the directory in use (or None)."""
    if _os.environ.get("IPSR_NO_MIOPEN_DB", "0") == "1":
        return None
    if "MIOPEN_USER_DB_PATH" in _os.environ:
        return _os.environ["MIOPEN_USER_DB_PATH"]
    import shutil
    import tempfile
    src = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "miopen_db")
    rank = _os.environ.get("LOCAL_RANK", "0") if local_rank is None else str(local_rank)
    dst = _os.path.join(tempfile.gettempdir(), "ipsr_miopen_db_%d_%s" % (_os.getuid(), rank))
    try:
        _os.makedirs(dst, exist_ok=True)
        for f in _os.listdir(src):
            if f.endswith(".txt") and not _os.path.exists(_os.path.join(dst, f)):
                shutil.copy(_os.path.join(src, f), _os.path.join(dst, f))
        _os.environ["MIOPEN_USER_DB_PATH"] = dst
        return dst
    except OSError:
        return None          # MIOpen then uses its default user db: slower solver picks, same results
