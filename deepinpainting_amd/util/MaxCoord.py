"""Host-side mirror of the reference's util/MaxCoord.py: channel arg-max of a materialised correlation
map.  (The layer itself uses the fused kernel and never materialises the map.)"""
import torch


class MaxCoord():
    def update_output(self, input, sp_x, sp_y):
        """reference :16-28.  input [1,N,h,w] -> (None, ind [h*w], vmax [h*w]); the reference's first
        return value is a dead all-zero tensor of the input's size (:21), returned here as None."""
        assert input.dim() == 4, "Input must be 3D or 4D(batch)."
        assert input.size(0) == 1, "The first dimension of input has to be 1!"
        v_max, c_max = torch.max(input, 1)
        return None, c_max.view(-1), v_max.view(-1)
