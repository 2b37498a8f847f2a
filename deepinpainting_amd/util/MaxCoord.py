"""Host-side mirror of the reference's util/MaxCoord.py: channel arg-max of a MATERIALISED correlation map, for callers
that hold one (the layer itself never does: its correlation kernel folds the arg-max into the GEMM epilogue)."""
import torch


class MaxCoord():
    def __init__(self):
        pass

    def update_output(self, input, sp_x, sp_y):
        """reference :16-28.  input [1,N,h,w] -> (zeros_like(input), ind [h*w] int64, vmax [h*w]).
        The first value is dead in the reference too (:21-24 never fills it) but callers read its size
        (models/IPSRFunction.py:67,72), so it is returned as the reference returns it.  sp_x / sp_y are accepted and unused
        ("just for Advanced Indexing", :8-11)."""
        assert input.dim() == 4, "Input must be 3D or 4D(batch)."
        assert input.size(0) == 1, "The first dimension of input has to be 1!"
        v_max, c_max = torch.max(input, 1)
        return torch.zeros_like(input), c_max.view(-1), v_max.view(-1)
