"""InnerCos2 — the decoder-side feature-consistency tap (reference models/InnerCos2.py): same loss as
InnerCos on the first 512 channels of its [B,1024,h,w] input (:38)."""
from .InnerCos import InnerCos


class InnerCos2(InnerCos):
    _narrow = 512

    def __init__(self, crit='MSE', strength=1, skip=0, infe=None):
        super(InnerCos2, self).__init__(crit=crit, strength=strength, skip=skip)
        self.inin = None
        self.infe = infe
