"""Fused conv-bias + InstanceNorm2d + activation for the U-Nets and discriminators (HIP kernels of csrc/instnorm.hip).

The reference builds every level as  Conv2d/ConvTranspose2d(bias) -> InstanceNorm2d -> LeakyReLU(0.2, True) / ReLU(True)
(models/networks.py:220-259, 404-432, 470-497, 507-514), the activation often being the FIRST module of the next
(child) level, applied in place.  On MIOpen that chain is a bias add, a 2-3 pass norm and an activation pass (and as
many again backward, plus a bias-gradient reduction).  `FusedSequential` keeps the module tree — children, names and
therefore state_dict keys are exactly those of nn.Sequential — and only changes HOW a run of such modules is executed:

    conv (called without bias) -> ONE kernel: bias + instance norm + affine + activation      (_InstNormAct)
    conv (called without bias) -> ONE kernel: bias + activation, in place                     (_BiasAct)

The in-place activation at the head of a child level is absorbed into the parent's kernel (the child is then told to skip
it), which is value-identical: the reference's in-place LeakyReLU rewrites the very tensor the parent's norm produced,
so nobody ever sees the un-activated values (that is also why the skip connection carries the activated tensor).

Activations may be fp32 or bf16 (BASELINE config 5 runs the convolutions under bf16 autocast; the kernels then read and
write bf16 and compute in fp32).  CPU tensors, other dtypes, planes above 128x128 and `FusedSequential.enabled = False`
take the plain module-by-module path.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


def _ticket_words(x, bias):
    """The per-channel arrival counters of the backward's in-launch bias-gradient sum: C words owned by THIS autograd node, zeroed by
    the node's forward launch (csrc/instnorm.hip: the library keeps no device state)."""
    if bias is None or not bias.requires_grad:
        return None
    return torch.empty(x.shape[1], dtype=torch.int32, device=x.device)


class _InstNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, gamma, beta, eps, act, slope):
        y, mean, rstd = ops.instnorm_act_forward(x, bias, gamma, beta, eps, act, slope)
        ctx.save_for_backward(x, bias, gamma, y, mean, rstd)
        ctx.act, ctx.slope = act, slope
        ctx.has_affine = gamma is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, bias, gamma, y, mean, rstd = ctx.saved_tensors
        need_bias = bias is not None and ctx.needs_input_grad[1]
        need_affine = ctx.has_affine and (ctx.needs_input_grad[2] or ctx.needs_input_grad[3])
        dx, dg, db, dbias = ops.instnorm_act_backward(dy, y, x, bias, gamma, mean, rstd, ctx.act, ctx.slope, need_affine, need_bias)
        return dx, dbias, dg, db, None, None, None


class _BiasAct(torch.autograd.Function):
    """act(x + bias[c]) in place on x (a fresh convolution output nobody else holds)."""

    @staticmethod
    def forward(ctx, x, bias, act, slope):
        tickets = _ticket_words(x, bias)
        ops.bias_act_(x, bias, act, slope, tickets=tickets)
        ctx.mark_dirty(x)
        ctx.save_for_backward(x, tickets)
        ctx.act, ctx.slope = act, slope
        ctx.has_bias = bias is not None
        return x

    @staticmethod
    def backward(ctx, dy):
        y, tickets = ctx.saved_tensors
        dx, dbias = ops.bias_act_backward(dy, y, ctx.act, ctx.slope, ctx.has_bias and ctx.needs_input_grad[1], tickets=tickets)
        return dx, dbias, None, None


class _BiasActSkip(torch.autograd.Function):
    """_BiasAct with a second output: relu(x + bias) written into the skip half of the child level's concatenated tensor (allocated
    here); backward dx = dy * act'(y) + dbuf[:, c1:] * relu'(y).  The level-1 blocks, whose input comes from a convolution alone."""

    @staticmethod
    def forward(ctx, x, bias, act, slope, c1):
        B, C2 = x.shape[0], x.shape[1]
        buf = torch.empty((B, c1 + C2) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device)
        tickets = _ticket_words(x, bias)
        ops.bias_act_(x, bias, act, slope, relu_into=buf, relu_at=c1, tickets=tickets)
        ctx.mark_dirty(x)
        ctx.save_for_backward(x, tickets)
        ctx.act, ctx.slope, ctx.c1 = act, slope, c1
        ctx.has_bias = bias is not None
        return x, buf

    @staticmethod
    def backward(ctx, dy, dbuf):
        y, tickets = ctx.saved_tensors
        if dy is None:
            dy = torch.zeros_like(y)
        dx, dbias = ops.bias_act_backward(dy.contiguous(), y, ctx.act, ctx.slope, ctx.has_bias and ctx.needs_input_grad[1],
                                          dy2=None if dbuf is None else dbuf.contiguous(), dy2_at=ctx.c1, tickets=tickets)
        return dx, dbias, None, None, None


class _CatReLU(torch.autograd.Function):
    """relu(cat([y, x], 1)): the skip concatenation of a level followed by the parent's in-place ReLU."""

    @staticmethod
    def forward(ctx, y, x):
        out = ops.cat_relu_forward(y, x)
        ctx.save_for_backward(out)
        ctx.c1 = y.size(1)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        return ops.cat_relu_backward(g, out, ctx.c1)


class _InstNormReLUCat(torch.autograd.Function):
    """relu(cat([InstanceNorm(y + bias) * gamma + beta, x], 1)) — a level's last norm, its skip concatenation and the parent's in-place
    ReLU as ONE node: the norm kernel writes its half straight into the concatenated tensor (and its backward reads the gradient's
    slice in place); only the skip half is left to the concatenation kernels."""

    @staticmethod
    def forward(ctx, y, bias, gamma, beta, eps, x):
        B, C1, C2 = y.shape[0], y.shape[1], x.shape[1]
        out = torch.empty((B, C1 + C2) + tuple(y.shape[2:]), dtype=y.dtype, device=y.device)
        _, mean, rstd = ops.instnorm_act_forward(y, bias, gamma, beta, eps, "relu", 0.0, into=out)
        ops.cat_relu_skip_half_(out, x)
        ctx.save_for_backward(y, bias, gamma, out, mean, rstd)
        ctx.c1 = C1
        ctx.has_affine = gamma is not None
        return out

    @staticmethod
    def backward(ctx, g):
        y, bias, gamma, out, mean, rstd = ctx.saved_tensors
        need_bias = bias is not None and ctx.needs_input_grad[1]
        need_affine = ctx.has_affine and (ctx.needs_input_grad[2] or ctx.needs_input_grad[3])
        g = g.contiguous()
        dyn, dg, db, dbias = ops.instnorm_act_backward(g, out, y, bias, gamma, mean, rstd, "relu", 0.0, need_affine, need_bias)
        _, dx = ops.cat_relu_backward(g, out, ctx.c1, skip_half_only=True)
        return dyn, dbias, dg, db, None, dx


class _InstNormActSkip(torch.autograd.Function):
    """The norm that produces a level's input, with TWO outputs: act(norm) for the level's down path and relu(norm) written straight
    into the skip half of the level's concatenated tensor (allocated here, completed by _InstNormReLUCatInto at the level's end).
    Backward: the two consumers' gradients meet inside the norm's backward kernel (no add kernel, no slice copy)."""

    @staticmethod
    def forward(ctx, y, bias, gamma, beta, eps, act, slope, c1):
        B, C2 = y.shape[0], y.shape[1]
        buf = torch.empty((B, c1 + C2) + tuple(y.shape[2:]), dtype=y.dtype, device=y.device)
        x_act, mean, rstd = ops.instnorm_act_forward(y, bias, gamma, beta, eps, act, slope, relu_into=buf, relu_at=c1)
        ctx.save_for_backward(y, bias, gamma, x_act, mean, rstd)
        ctx.act, ctx.slope, ctx.c1 = act, slope, c1
        ctx.has_affine = gamma is not None
        return x_act, buf

    @staticmethod
    def backward(ctx, g_x, g_buf):
        y, bias, gamma, x_act, mean, rstd = ctx.saved_tensors
        need_bias = bias is not None and ctx.needs_input_grad[1]
        need_affine = ctx.has_affine and (ctx.needs_input_grad[2] or ctx.needs_input_grad[3])
        if g_x is None:
            g_x = torch.zeros_like(x_act)
        dx, dg, db, dbias = ops.instnorm_act_backward(g_x.contiguous(), x_act, y, bias, gamma, mean, rstd, ctx.act, ctx.slope, need_affine, need_bias,
                                                      dy2=None if g_buf is None else g_buf.contiguous(), dy2_at=ctx.c1)
        return dx, dbias, dg, db, None, None, None, None


class _InstNormReLUCatInto(torch.autograd.Function):
    """_InstNormReLUCat into a concatenated tensor whose skip half is already there (written by _InstNormActSkip): the first C1
    channels of `buf` receive relu(norm(y)); the gradient of `buf` is handed back whole (its producer reads the skip half in place)."""

    @staticmethod
    def forward(ctx, y, bias, gamma, beta, eps, buf):
        _, mean, rstd = ops.instnorm_act_forward(y, bias, gamma, beta, eps, "relu", 0.0, into=buf, into_at=0)
        ctx.mark_dirty(buf)
        ctx.save_for_backward(y, bias, gamma, buf, mean, rstd)
        ctx.has_affine = gamma is not None
        return buf

    @staticmethod
    def backward(ctx, g):
        y, bias, gamma, out, mean, rstd = ctx.saved_tensors
        need_bias = bias is not None and ctx.needs_input_grad[1]
        need_affine = ctx.has_affine and (ctx.needs_input_grad[2] or ctx.needs_input_grad[3])
        g = g.contiguous()
        dyn, dg, db, dbias = ops.instnorm_act_backward(g, out, y, bias, gamma, mean, rstd, "relu", 0.0, need_affine, need_bias)
        return dyn, dbias, dg, db, None, g


def _tail_norm_channels(block):
    """Output channels of a skip level whose sequence ends in a plain InstanceNorm (then the level can complete a concatenated tensor
    its producer allocated), else None."""
    inner = getattr(block, "model", None)
    if getattr(block, "outermost", True) or not isinstance(inner, FusedSequential) or len(inner) == 0 or not _plain_inorm(inner[-1]):
        return None
    return int(inner[-1].num_features)


def cat_skip(y, x, tail_relu):
    """torch.cat([y, x], 1), with the consumer's in-place ReLU folded in when it asked for that."""
    if tail_relu and y.is_cuda and y.dtype == x.dtype and y.dtype in (torch.float32, torch.bfloat16) \
            and y.is_contiguous() and x.is_contiguous() and y.shape[2:] == x.shape[2:]:
        return _CatReLU.apply(y, x)
    out = torch.cat([y, x], 1)
    return torch.relu_(out) if tail_relu else out


def _act_of(m):
    """(name, slope) when m is an activation the kernels implement."""
    if isinstance(m, nn.LeakyReLU):
        return "leaky", float(m.negative_slope)
    if isinstance(m, nn.ReLU):
        return "relu", 0.0
    return None


def _conv_no_bias(m, x):
    from .hipconv import conv_nobias
    return conv_nobias(m, x)


def _plain_conv(m):
    return type(m) in (nn.Conv2d, nn.ConvTranspose2d) and m.bias is not None and getattr(m, "padding_mode", "zeros") == "zeros"


def _plain_inorm(m):
    return isinstance(m, nn.InstanceNorm2d) and not m.track_running_stats


class FusedSequential(nn.Sequential):
    enabled = True          # class-wide switch (Option.fused_norm_act); False = behave exactly like nn.Sequential
    skip_source = True      # A/B switch (tests): False = a level's producer does NOT write the skip half of the child's concatenated tensor
                            # (no _InstNormActSkip / _BiasActSkip nodes: the child concatenates, autograd adds the two input gradients)

    def _get_name(self):    # prints like the reference's module tree (train.ipynb cell 1 output)
        return 'Sequential'

    def _head_act_of_child(self, m):
        """A skip level whose first module is an in-place activation can have it absorbed by the producer."""
        inner = getattr(m, "model", None)
        if isinstance(inner, FusedSequential) and len(inner) > 0 and getattr(inner[0], "inplace", False):
            return _act_of(inner[0])
        return None

    def _tail_relu_after(self, mods, j):
        """True when mods[j] is the in-place ReLU a skip level's concatenated output runs into (models/networks.py:229,408)."""
        return j < len(mods) and type(mods[j]) is nn.ReLU and mods[j].inplace

    def forward(self, x, head_act_done=False, cat_with=None, cat_buf=None):
        """cat_with: the level's input, when the caller (networks._cat_skip) wants relu(cat([this(x), cat_with], 1)) and this sequence
        may produce it itself if it ends in a norm; the return value is then (tensor, concatenated?)."""
        out = self._forward(x, head_act_done, cat_with, cat_buf)
        if cat_with is None:
            return out[0]
        return out

    def _forward(self, x, head_act_done, cat_with, cat_buf):
        usable = FusedSequential.enabled and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16)
        mods = list(self)
        n = len(mods)
        i = 0
        if head_act_done:                       # the producer already applied this level's leading in-place activation
            assert n > 0 and _act_of(mods[0]) is not None
            i = 1
        if not usable:
            for m in mods[i:]:
                x = m(x)
            return x, False
        while i < n:
            m = mods[i]
            conv = _plain_conv(m)
            norm = mods[i + 1] if (conv and i + 1 < n and _plain_inorm(mods[i + 1])) else (m if _plain_inorm(m) else None)
            if conv and norm is None:
                # conv -> activation (same level, or the child's leading in-place one)
                nxt = mods[i + 1] if i + 1 < n else None
                act = _act_of(nxt) if nxt is not None else None
                child_act = self._head_act_of_child(nxt) if (nxt is not None and act is None) else None
                if act is None and child_act is None:
                    # conv -> something the kernels do not fuse (Tanh of the outermost level, a loss): the plain module, unless a HIP
                    # engine takes one of its passes (netG's last ConvTranspose2d 128 -> 3 @256x256: 0.13 vs 0.24 ms forward) — then
                    # the bias is split off and added (with its gradient as a by-product) by the bias kernel
                    from .hipconv import any_engine
                    if any_engine(m, x):
                        y = _conv_no_bias(m, x)
                        x = _BiasAct.apply(y, m.bias, "none", 0.0) if (y.is_contiguous() and y.dtype in (torch.float32, torch.bfloat16)) \
                            else y + m.bias.view(1, -1, 1, 1).to(y.dtype)
                    else:
                        x = m(x)
                    i += 1
                    continue
                y = _conv_no_bias(m, x)
                if not y.is_contiguous() or y.dtype not in (torch.float32, torch.bfloat16):      # e.g. channels_last: plain modules
                    x = y + m.bias.view(1, -1, 1, 1).to(y.dtype)
                    i += 1
                    continue
                if act is not None:
                    x = _BiasAct.apply(y, m.bias, act[0], act[1])
                    i += 2
                else:
                    tail = self._tail_relu_after(mods, i + 2)
                    c1 = _tail_norm_channels(nxt) if (tail and FusedSequential.skip_source) else None
                    if c1:      # this bias pass also writes the skip half of the child's concatenated tensor
                        xa, buf = _BiasActSkip.apply(y, m.bias, child_act[0], child_act[1], c1)
                        x = nxt(xa, head_act_done=True, tail_relu=tail, cat_buf=buf)
                    else:
                        x = nxt(_BiasAct.apply(y, m.bias, child_act[0], child_act[1]), head_act_done=True, tail_relu=tail)
                    i += 3 if tail else 2
                continue
            if norm is not None:
                if conv:
                    y, bias, j = _conv_no_bias(m, x), m.bias, i + 2
                else:
                    y, bias, j = x, None, i + 1
                hw = y.size(2) * y.size(3)
                if hw > ops.INSTNORM_MAX_PLANE or (hw > 16384 and hw % 4) or hw < 2 or not y.is_contiguous() \
                        or y.dtype not in (torch.float32, torch.bfloat16):
                    if conv:                       # too large for the plane-in-registers kernel: plain modules
                        y = y + m.bias.view(1, -1, 1, 1)
                    x = norm(y)
                    i = j
                    continue
                nxt = mods[j] if j < n else None
                if nxt is None and cat_with is not None and cat_with.is_contiguous() and cat_with.dtype == y.dtype \
                        and cat_with.shape[0] == y.shape[0] and cat_with.shape[2:] == y.shape[2:]:
                    # the level ends here: norm + skip concatenation + the parent's ReLU in one node
                    if cat_buf is not None and cat_buf.dtype == y.dtype and cat_buf.shape[0] == y.shape[0] and cat_buf.shape[2:] == y.shape[2:] \
                            and cat_buf.shape[1] == y.shape[1] + cat_with.shape[1]:
                        return _InstNormReLUCatInto.apply(y, bias, norm.weight, norm.bias, norm.eps, cat_buf), True
                    return _InstNormReLUCat.apply(y, bias, norm.weight, norm.bias, norm.eps, cat_with), True
                act = _act_of(nxt) if nxt is not None else None
                child_act = self._head_act_of_child(nxt) if (nxt is not None and act is None) else None
                a = act or child_act or ("none", 0.0)
                if act is None and child_act is not None:
                    tail = self._tail_relu_after(mods, j + 1)
                    c1 = _tail_norm_channels(nxt) if (tail and FusedSequential.skip_source) else None
                    if c1:      # this norm also writes the skip half of the child's concatenated tensor
                        x, buf = _InstNormActSkip.apply(y, bias, norm.weight, norm.bias, norm.eps, a[0], a[1], c1)
                        x = nxt(x, head_act_done=True, tail_relu=tail, cat_buf=buf)
                    else:
                        x = _InstNormAct.apply(y, bias, norm.weight, norm.bias, norm.eps, a[0], a[1])
                        x = nxt(x, head_act_done=True, tail_relu=tail)
                    i = j + 2 if tail else j + 1
                    continue
                x = _InstNormAct.apply(y, bias, norm.weight, norm.bias, norm.eps, a[0], a[1])
                if act is not None:
                    i = j + 1
                elif child_act is not None:
                    tail = self._tail_relu_after(mods, j + 1)
                    x = nxt(x, head_act_done=True, tail_relu=tail)
                    i = j + 2 if tail else j + 1
                else:
                    i = j
                continue
            x = m(x)
            i += 1
        return x, False
