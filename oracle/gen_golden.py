#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Container-only tool: it imports /root/reference (read-only) and therefore cannot run on the GPU box;
only its outputs (small .npz data files = inputs + the reference's outputs) are committed.  The
reference is pure Python on torch; it hard-codes CUDA tensors through `torch.cuda.is_available`
(used without calling it — models/IPSRFunction.py:28,38, util/NonparametricShift.py:15,
models/InnerCos.py:19, models/InnerCos2.py:22), so this harness aliases the CUDA tensor types to the
CPU ones IN THIS PROCESS ONLY; no reference file is modified or copied.

    python oracle/gen_golden.py            # rewrites tests/golden/*.npz

Cases (all fp32; x = generator feature, ref = stand-in for VGG relu4_3 of the reference image):
  well-conditioned cases use x = |N(0,1)|, ref ~ U(0,1); `signed` is the deliberately
  ill-conditioned one (attention weights a/(a+vmax) blow up on signed features).
"""
import os
import sys
import types
from collections import namedtuple

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from golden_cases import big_case_inputs, grad_seed, reinit_deterministic, trainer_inputs, net_input  # noqa: E402

REF = os.environ.get("IPSR_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _install_cpu_aliases():
    torch.cuda.FloatTensor = torch.FloatTensor
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self


def _import_reference():
    sys.path.insert(0, REF)
    import util.util as rutil  # noqa
    from util.NonparametricShift import NonparametricShift
    from util.MaxCoord import MaxCoord
    from models.IPSRFunction import IPSRFunction
    from models.IPSR_model import IPSR_model
    from models.InnerCos import InnerCos
    from models.InnerCos2 import InnerCos2
    return rutil, NonparametricShift, MaxCoord, IPSRFunction, IPSR_model, InnerCos, InnerCos2


Vgg = namedtuple("VggOutputs", ["relu1_2", "relu2_2", "relu3_3", "relu4_3"])


def center_mask(size, lo, hi):
    m = np.zeros((size, size), np.uint8)
    m[lo:hi, lo:hi] = 1
    return m


def stroke_mask(size, seed, strokes=6):
    """Seeded free-form mask: random-walk strokes (BASELINE.json config 3 shape)."""
    rs = np.random.RandomState(seed)
    m = np.zeros((size, size), np.uint8)
    for _ in range(strokes):
        y, x = rs.randint(0, size, 2)
        wd = rs.randint(size // 20 + 1, size // 6 + 2)
        for _ in range(rs.randint(4, 12)):
            dy, dx = rs.randint(-size // 6, size // 6 + 1, 2)
            steps = max(abs(dy), abs(dx), 1)
            for s in range(steps + 1):
                yy = int(np.clip(y + dy * s / steps, 0, size - 1))
                xx = int(np.clip(x + dx * s / steps, 0, size - 1))
                m[max(0, yy - wd // 2):yy + wd // 2 + 1, max(0, xx - wd // 2):xx + wd // 2 + 1] = 1
            y = int(np.clip(y + dy, 0, size - 1))
            x = int(np.clip(x + dx, 0, size - 1))
    return m


def run_layer_case(R, name, x, ref, mask_img, threshold=5 / 16.0, triple_w=1.0, keep_channels=None,
                   strength=1.0, note=""):
    """Run IPSR_model-equivalent forward/backward of the reference and dump a fixture."""
    rutil, NonparametricShift, MaxCoord, IPSRFunction, IPSR_model, InnerCos, InnerCos2 = R
    B, C, h, w = x.shape
    xt = torch.from_numpy(x.copy())
    reft = Vgg(None, None, None, torch.from_numpy(ref.copy()))
    mask_global = torch.from_numpy(mask_img.astype(bool))[None, None]

    # --- mask side: cal_feat_mask + cal_mask_given_mask_thred through the reference's own wrapper
    layer = IPSR_model(threshold, 1, 1, 1, 1, triple_w)
    feat = layer.set_mask(mask_global, 3, threshold)
    assert tuple(feat.shape) == (h, w), (feat.shape, h, w)
    flag, nonmask_idx, flatten_offsets, mask_idx = rutil.cal_mask_given_mask_thred(
        xt[0], feat, 1, 1, 1)
    sp_x, sp_y = rutil.cal_sps_for_Advanced_Indexing(h, w)

    # --- capture intermediates the Function keeps local: MaxCoord outputs, the dense kbar (= A^T)
    rec = {"ind": [], "vmax": [], "S": [], "kbar": []}
    orig_update = MaxCoord.update_output
    orig_build = NonparametricShift.buildAutoencoder

    def update_output(self, inp, sx, sy):
        o = orig_update(self, inp, sx, sy)
        rec["S"].append(inp.detach().clone().numpy()[0].reshape(inp.size(1), -1))
        rec["ind"].append(o[1].clone().numpy())
        rec["vmax"].append(o[2].clone().numpy())
        return o

    def build(self, *a, **k):
        r = list(orig_build(self, *a, **k))
        dec = r[2]

        class Rec(torch.nn.Module):
            def forward(self_inner, kbar):
                rec["kbar"].append(kbar.detach().clone().numpy()[0].reshape(kbar.size(1), -1))
                return dec(kbar)
        r[2] = Rec()
        return tuple(r)

    MaxCoord.update_output = update_output
    NonparametricShift.buildAutoencoder = build
    try:
        ctx = types.SimpleNamespace()
        with torch.no_grad():
            out = IPSRFunction.forward(ctx, xt, feat, reft, 1, 1, triple_w, flag, nonmask_idx, mask_idx,
                                       flatten_offsets, sp_x, sp_y)
        rs = np.random.RandomState(grad_seed(name))
        g = rs.standard_normal(x.shape).astype(np.float32)
        with torch.no_grad():
            gin = IPSRFunction.backward(ctx, torch.from_numpy(g.copy()))[0]
        # cross-check: the autograd path gives the same numbers
        layer.set_ref(reft)
        xa = torch.from_numpy(x.copy()).requires_grad_(True)
        ya = layer(xa)
        ya.backward(torch.from_numpy(g.copy()))
        assert torch.equal(ya.detach(), out) and torch.equal(xa.grad, gin)
    finally:
        MaxCoord.update_output = orig_update
        NonparametricShift.buildAutoencoder = orig_build

    # --- InnerCos / InnerCos2 on the same tensors (target = stand-in for VGG relu4_3 of the GT)
    opt = types.SimpleNamespace(threshold=threshold)
    tgt = np.random.RandomState(7).rand(B, C, h, w).astype(np.float32)
    ic = InnerCos(strength=strength, skip=0)
    ic.set_mask(mask_global, opt)
    ic.set_target(torch.from_numpy(tgt.copy()))
    xi = torch.from_numpy(x.copy()).requires_grad_(True)
    ic(xi)
    loss1 = ic.loss.detach().numpy().copy()
    ic.backward()
    gloss1 = xi.grad.numpy().copy()

    M = int(mask_idx.numel())
    N = h * w
    for k_ in rec:                            # the autograd cross-check above recorded a second copy
        assert len(rec[k_]) == 2 * B
        rec[k_] = rec[k_][:B]
    kbar = np.stack(rec["kbar"])              # [B, N(k), N(q)]  (A transposed)
    attn_rows = np.stack([kbar[b][:, mask_idx.numpy()].T for b in range(B)]) if M else np.zeros((B, 0, N), np.float32)
    d = dict(
        x=x, ref=ref, mask_img=mask_img, threshold=np.float32(threshold), triple_w=np.float32(triple_w),
        strength=np.float32(strength),
        feat_mask=feat.numpy().astype(np.uint8), flag=flag.numpy(), nonmask_point_idx=nonmask_idx.numpy(),
        flatten_offsets=flatten_offsets.numpy(), mask_point_idx=mask_idx.numpy(),
        sp_x=sp_x.numpy(), sp_y=sp_y.numpy(),
        ind=np.stack(rec["ind"]).astype(np.int64), vmax=np.stack(rec["vmax"]),
        attn_rows=attn_rows.astype(np.float32),
        trunc_kbar_nnz=np.array([int((ctx.ind_lst[b] != 0).sum()) for b in range(B)], np.int64),
        out=out.numpy(), grad_out=g, grad_in=gin.numpy(),
        ic_target=tgt, ic_loss=loss1, ic_grad=gloss1,
        note=np.array(note),
    )
    if N <= 256:
        d["S"] = np.stack(rec["S"])           # [B, N(k), N(q)]
    if keep_channels is not None:             # big case: keep the index/scalar outputs + a channel slice
        ch = np.asarray(keep_channels)
        for k_ in ("out", "grad_in"):
            d[k_ + "_channels"] = ch
            d[k_] = d[k_][:, ch]
        for k_ in ("x", "ref", "grad_out", "ic_target", "ic_grad", "attn_rows"):
            d.pop(k_)
        d["regen"] = np.array("x,ref,grad_out,ic_target are regenerated by tests/golden_cases.py from the seeds")
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **d)
    print("%-28s B=%d C=%d %dx%d M=%d  -> %s (%.1f KB)" % (name, B, C, h, w, M, os.path.relpath(path),
                                                          os.path.getsize(path) / 1024.0))


class _Sink(object):
    """Stands in for the `ind_lst` LongTensor of models/IPSRFunction.py:36 when shift_sz > 1: the reference sizes it
    [bz, h*w, h, w] while kbar is [N', h-p+1, w-p+1], so the assignment at :134 raises.  Everything BEFORE that line
    (the whole forward arithmetic, :46-133) is well defined for any patch size; swallowing that one store lets the
    reference's own forward produce the p>1 output.  Harness-only; no reference file is touched."""
    def __init__(self, *a):
        pass

    def cuda(self, *a, **k):
        return self

    def __setitem__(self, k, v):
        pass


def run_layer_case_patch(R, name, x, ref, mask_img, patch, threshold=5 / 16.0, note=""):
    """Forward-only fixture for shift_sz = patch > 1 (BASELINE config 4's 3x3 patches).  The reference defines no
    usable backward for it (its backward indexes an N x N matrix by h*w, :158-170), so none is recorded."""
    rutil, NonparametricShift, MaxCoord, IPSRFunction, IPSR_model, InnerCos, InnerCos2 = R
    B, C, h, w = x.shape
    xt = torch.from_numpy(x.copy())
    reft = Vgg(None, None, None, torch.from_numpy(ref.copy()))
    mask_global = torch.from_numpy(mask_img.astype(bool))[None, None]
    layer = IPSR_model(threshold, 1, patch, 1, 1, 1.0)
    feat = layer.set_mask(mask_global, 3, threshold)
    assert tuple(feat.shape) == (h, w), (feat.shape, h, w)
    flag, nonmask_idx, flatten_offsets, mask_idx = rutil.cal_mask_given_mask_thred(xt[0], feat, patch, 1, 1)
    sp_x, sp_y = rutil.cal_sps_for_Advanced_Indexing(h, w)
    rec = {"ind": [], "vmax": [], "kbar": []}
    orig_update = MaxCoord.update_output
    orig_build = NonparametricShift.buildAutoencoder
    orig_long = torch.LongTensor

    def update_output(self, inp, sx, sy):
        o = orig_update(self, inp, sx, sy)
        rec["ind"].append(o[1].clone().numpy())
        rec["vmax"].append(o[2].clone().numpy())
        return o

    def build(self, *a, **k):
        r = list(orig_build(self, *a, **k))
        dec = r[2]

        class Rec(torch.nn.Module):
            def forward(self_inner, kbar):
                rec["kbar"].append(kbar.detach().clone().numpy()[0].reshape(kbar.size(1), -1))
                return dec(kbar)
        r[2] = Rec()
        return tuple(r)

    MaxCoord.update_output = update_output
    NonparametricShift.buildAutoencoder = build
    torch.LongTensor = _Sink
    try:
        ctx = types.SimpleNamespace()
        with torch.no_grad():
            out = IPSRFunction.forward(ctx, xt, feat, reft, patch, 1, 1.0, flag, nonmask_idx, mask_idx,
                                       flatten_offsets, sp_x, sp_y)
    finally:
        torch.LongTensor = orig_long
        MaxCoord.update_output = orig_update
        NonparametricShift.buildAutoencoder = orig_build
    M = int(mask_idx.numel())
    kbar = np.stack(rec["kbar"])                                  # [B, N'(k), N'(q)]
    attn_rows = np.stack([kbar[b][:, mask_idx.numpy()].T for b in range(B)]) if M else np.zeros((B, 0, kbar.shape[1]), np.float32)
    d = dict(x=x, ref=ref, mask_img=mask_img, threshold=np.float32(threshold), patch=np.int64(patch),
             feat_mask=feat.numpy().astype(np.uint8), flag=flag.numpy(), mask_point_idx=mask_idx.numpy(),
             ind=np.stack(rec["ind"]).astype(np.int64), vmax=np.stack(rec["vmax"]),
             attn_rows=attn_rows.astype(np.float32), out=out.numpy(), note=np.array(note))
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **d)
    print("%-28s B=%d C=%d %dx%d p=%d N'=%d M=%d  -> %s (%.1f KB)" % (name, B, C, h, w, patch, kbar.shape[1], M,
                                                                  os.path.relpath(path), os.path.getsize(path) / 1024.0))


def run_innercos2_case(R, name):
    rutil, _, _, _, _, InnerCos, InnerCos2 = R
    rs = np.random.RandomState(11)
    B, C, h = 2, 1024, 8     # InnerCos2 narrows to the first 512 channels (models/InnerCos2.py:38)
    x = rs.standard_normal((B, C, h, h)).astype(np.float32)
    tgt = rs.rand(B, 512, h, h).astype(np.float32)
    mask_img = center_mask(64, 16, 48)
    opt = types.SimpleNamespace(threshold=5 / 16.0)
    ic = InnerCos2(strength=2.0, skip=0)
    ic.set_mask(torch.from_numpy(mask_img.astype(bool))[None, None], opt)
    ic.set_target(torch.from_numpy(tgt.copy()))
    xi = torch.from_numpy(x.copy()).requires_grad_(True)
    y = ic(xi)
    assert y is xi
    loss = ic.loss.detach().numpy().copy()
    ic.backward()
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, x=x, ic_target=tgt, mask_img=mask_img, threshold=np.float32(5 / 16.0),
                        strength=np.float32(2.0), ic_loss=loss, ic_grad=xi.grad.numpy().copy(),
                        feat_mask=ic.mask.numpy().astype(np.uint8))
    print("%-28s -> %s (%.1f KB)" % (name, os.path.relpath(path), os.path.getsize(path) / 1024.0))


def run_mask_cases(R, name):
    """cal_feat_mask + cal_mask_given_mask_thred on a spread of mask shapes (incl. the reference's own
    known answer M = 252 for the fineSize=256 / overlap=4 centre mask, util/NonparametricShift.py:17)."""
    rutil = R[0]
    cases = {
        "center64": center_mask(64, 16, 48),
        "center256": center_mask(256, 64, 192),
        "center256_overlap4": center_mask(256, 68, 188),      # models/IPSR.py:40-41
        "center512": center_mask(512, 128, 384),
        "stroke256_a": stroke_mask(256, 1),
        "stroke256_b": stroke_mask(256, 2, strokes=10),
        "stroke128": stroke_mask(128, 3),
        "empty64": np.zeros((64, 64), np.uint8),
        "full64": np.ones((64, 64), np.uint8),
        "edge64": np.pad(np.ones((20, 64), np.uint8), ((0, 44), (0, 0))),
    }
    d = {}
    for key, m in cases.items():
        for thr in (5 / 16.0, 0.5):
            feat = rutil.cal_feat_mask(torch.from_numpy(m.astype(bool))[None, None], 3, thr).squeeze()
            h, w = feat.shape
            dummy = torch.zeros(1, h, w)
            flag, nonmask, fo, midx = rutil.cal_mask_given_mask_thred(dummy, feat, 1, 1, 1)
            tag = "%s__thr%g" % (key, thr)
            d[tag + "__mask"] = m
            d[tag + "__feat"] = feat.numpy().astype(np.uint8)
            d[tag + "__flag"] = flag.numpy()
            d[tag + "__nonmask"] = nonmask.numpy()
            d[tag + "__flatten_offsets"] = fo.numpy()
            d[tag + "__mask_point_idx"] = midx.numpy()
            if key == "center256_overlap4" and thr == 5 / 16.0:
                assert midx.numel() == 252, midx.numel()
    sx, sy = rutil.cal_sps_for_Advanced_Indexing(5, 7)
    d["sps_5x7__sp_x"] = sx.numpy()
    d["sps_5x7__sp_y"] = sy.numpy()
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **d)
    print("%-28s %d masks -> %s (%.1f KB)" % (name, len(cases) * 2, os.path.relpath(path), os.path.getsize(path) / 1024.0))


class _Opt(object):
    """train.ipynb cell 0 defaults (the fields the models read)."""
    def __init__(self, **kw):
        d = dict(batchSize=1, fineSize=256, input_nc=3, input_nc_g=6, output_nc=3, ngf=64, ndf=64,
                 which_model_netD='basic', which_model_netF='feature', which_model_netG='unet_ipsr',
                 which_model_netP='unet_256', triple_weight=1, name='golden', n_layers_D='3', gpu_ids=[],
                 model='ipsr_net', checkpoints_dir='/tmp/ipsr_golden_ckpt', norm='instance', fixed_mask=1,
                 use_dropout=False, init_type='normal', mask_type='random', lambda_A=100, threshold=5 / 16.0,
                 stride=1, shift_sz=1, mask_thred=1, bottleneck=512, gp_lambda=10.0, ncritic=5, constrain='MSE',
                 strength=1, init_gain=0.02, cosis=1, gan_type='lsgan', gan_weight=0.2, overlap=4, skip=0,
                 continue_train=False, epoch_count=1, phase='train', which_epoch='', niter=20, niter_decay=100,
                 beta1=0.5, lr=0.0002, lr_policy='lambda', lr_decay_iters=50, isTrain=True)
        d.update(kw)
        self.__dict__.update(d)


def _sub(t):
    """Keep fixtures small: every 5th pixel of every channel."""
    return t.detach().numpy()[..., ::5, ::5].copy()


def run_network_cases(name):
    """Module tree (state_dict keys + shapes), parameter counts (train.ipynb cell 1 output) and forward
    outputs of the reference's four nets under deterministic weights."""
    import io
    import contextlib
    from models import networks as rnet
    opt = _Opt()
    mask_global = torch.zeros(1, 1, 256, 256, dtype=torch.bool)
    mask_global[:, :, 64:192, 64:192] = 1
    d = {}
    with contextlib.redirect_stdout(io.StringIO()):
        netG, cos1, cos2, csa = rnet.define_G(6, 3, 64, 'unet_ipsr', opt, mask_global, 'instance', False, 'normal', [], 0.02)
        netP, _, _, _ = rnet.define_G(3, 3, 64, 'unet_256', opt, mask_global, 'instance', False, 'normal', [], 0.02)
        netD = rnet.define_D(3, 64, 'basic', '3', 'instance', False, 'normal', [], 0.02)
        netF = rnet.define_D(3, 64, 'feature', '3', 'instance', False, 'normal', [], 0.02)
    for tag, net in (("G", netG), ("P", netP), ("D", netD), ("F", netF)):
        sd = net.state_dict()
        d["keys_" + tag] = np.array(list(sd.keys()))
        d["shapes_" + tag] = np.array([str(tuple(v.shape)) for v in sd.values()])
        d["nparams_" + tag] = np.int64(sum(p.numel() for p in net.parameters()))
        reinit_deterministic(net, 100 + ord(tag))
        net.eval()
    assert [int(d["nparams_" + t]) for t in "GPDF"] == [77692291, 54419459, 2766529, 10487296]
    with torch.no_grad():
        d["out_P"] = _sub(netP(net_input((1, 3, 256, 256), 1)))
        d["out_D"] = netD(net_input((1, 3, 256, 256), 2)).numpy()
        d["out_F"] = netF(net_input((1, 256, 32, 32), 3)).numpy()
        # netG needs the layer state: ref features + InnerCos targets (values unused by the forward output)
        ref_feat = net_input((1, 512, 32, 32), 4).abs()
        csa[0].set_ref(Vgg(None, None, None, ref_feat))
        cos1[0].set_target(ref_feat)
        cos2[0].set_target(ref_feat)
        out_G = netG(net_input((1, 6, 256, 256), 5))
        d["out_G"] = _sub(out_G)
        d["ic_loss_G"] = np.array([cos1[0].loss.item(), cos2[0].loss.item()], np.float32)
    # GANLoss known answers
    gl = rnet.GANLoss(gan_type='lsgan', tensor=torch.FloatTensor)
    a, b = net_input((2, 1, 30, 30), 6), net_input((2, 1, 30, 30), 7)
    d["ganloss"] = np.array([gl(a, b, True).item(), gl(a, b, False).item()], np.float32)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **d)
    print("%-28s -> %s (%.1f KB)" % (name, os.path.relpath(path), os.path.getsize(path) / 1024.0))


def run_trainer_case(name):
    """One optimize_parameters() of the reference's IPSR trainer on CPU.  models/vgg16.py imports
    torchvision, absent from this image; a minimal `torchvision.models.vgg16` with the standard VGG16-D
    `features` layout (random init, no download) is registered for this process so the import succeeds.
    All weights (4 nets + VGG) are then overwritten deterministically, so the stand-in's init is moot."""
    import io
    import contextlib
    import types as _types
    import torch.nn as nn

    def vgg16(pretrained=False, **kw):
        cfg = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']
        layers, cin = [], 3
        for v in cfg:
            if v == 'M':
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        return _types.SimpleNamespace(features=nn.Sequential(*layers))

    tv = _types.ModuleType("torchvision")
    tv.models = _types.ModuleType("torchvision.models")
    tv.models.vgg16 = vgg16
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tv.models
    from models.models import create_model
    import models.IPSR as rIPSR

    opt = _Opt(batchSize=1)
    # models/IPSR.py:19 hard-codes torch.device('cuda'); build, then point the object at the CPU
    orig_device = torch.device
    with contextlib.redirect_stdout(io.StringIO()):
        model = create_model(opt)
    model.device = orig_device('cpu')
    for i, net in enumerate((model.netG, model.netP, model.netD, model.netF, model.vgg)):
        reinit_deterministic(net, 500 + i)
    img, mask, ref = trainer_inputs()
    # the reference keeps kbar in a LongTensor (models/IPSRFunction.py:36,134): capture what ITS forward stored for the
    # backward (ctx.ind_lst) — the truncation that decides which gradient columns exist upstream of the layer
    import models.IPSRFunction as rF
    captured = []
    orig_forward = rF.IPSRFunction.forward

    def capturing_forward(ctx, *a, **k):
        out = orig_forward(ctx, *a, **k)
        captured.append(ctx.ind_lst.detach().clone())
        return out
    rF.IPSRFunction.forward = staticmethod(capturing_forward)
    model.set_input(img, mask, ref)
    model.set_ref_latent()
    model.set_gt_latent()
    model.optimize_parameters()
    errs = model.get_current_errors()

    def sparse_kbar(t):
        kb = t[0].reshape(t.shape[1], -1)                           # [N (patch k), N (position q)] int64
        kk, qq = torch.nonzero(kb, as_tuple=True)
        return kk.numpy().astype(np.int32), qq.numpy().astype(np.int32), kb[kk, qq].numpy().astype(np.int32), np.int64(kb.shape[0])
    assert len(captured) == 1                                       # one layer forward per optimize_parameters
    k1 = sparse_kbar(captured[0])
    d = dict(
        kbar_k=k1[0], kbar_q=k1[1], kbar_v=k1[2], kbar_n=k1[3],
        errors=np.array([errs['G_GAN'], errs['G_L1'], errs['D'], errs['F']], np.float64),
        ng_loss=np.array([float(model.ng_loss_value), float(model.ng_loss_value2)], np.float64),
        loss_G=np.float64(model.loss_G.item()), loss_D=np.float64(model.loss_D.item()),
        get_loss=np.float64(model.get_loss()['GAN']),
        fake_B=_sub(model.fake_B), fake_P=_sub(model.fake_P), real_A=_sub(model.real_A),
        n_visuals=np.int64(len(model.get_current_visuals())),
    )
    # a few post-step parameter values: pins the Adam update + the gradient paths (incl. IPSRFunction.backward)
    for tag, net in (("G", model.netG), ("P", model.netP), ("D", model.netD), ("F", model.netF)):
        sd = net.state_dict()
        ks = [k for k in sd if k.endswith("weight")]
        pick = [ks[0], ks[len(ks) // 2], ks[-1]]
        d["post_keys_" + tag] = np.array(pick)
        named = dict(net.named_parameters())
        for j, k in enumerate(pick):
            d["post_%s_%d" % (tag, j)] = sd[k].detach().numpy().reshape(-1)[:256].copy()
            d["grad_%s_%d" % (tag, j)] = named[k].grad.detach().numpy().reshape(-1)[:256].copy()
    # second iteration's losses: depend on every updated weight
    model.set_input(img, mask, ref)
    model.set_ref_latent()
    model.set_gt_latent()
    model.optimize_parameters()
    rF.IPSRFunction.forward = staticmethod(orig_forward)
    e2 = model.get_current_errors()
    d["errors_iter2"] = np.array([e2['G_GAN'], e2['G_L1'], e2['D'], e2['F']], np.float64)
    assert len(captured) == 2
    k2 = sparse_kbar(captured[1])
    d.update(kbar2_k=k2[0], kbar2_q=k2[1], kbar2_v=k2[2])
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **d)
    print("%-28s -> %s (%.1f KB)  errors=%s" % (name, os.path.relpath(path), os.path.getsize(path) / 1024.0, dict(errs)))


def patch_cases(R):
    def feats(seed, B, C, h):
        rs = np.random.RandomState(seed)
        return np.abs(rs.standard_normal((B, C, h, h))).astype(np.float32), rs.rand(B, C, h, h).astype(np.float32)
    x, ref = feats(201, 2, 16, 12)
    run_layer_case_patch(R, "patch_layer_p3_c16_12x12_center", x, ref, center_mask(96, 32, 64), 3,
                         note="shift_sz=3 forward (BASELINE config 4 patch size)")
    x, ref = feats(202, 1, 64, 16)
    run_layer_case_patch(R, "patch_layer_p3_c64_16x16_stroke", x, ref, stroke_mask(128, 3), 3)
    x, ref = feats(203, 2, 24, 8)
    run_layer_case_patch(R, "patch_layer_p2_c24_8x8_center", x, ref, center_mask(64, 16, 48), 2, note="even patch size")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "patch":          # only the shift_sz > 1 fixtures
        _install_cpu_aliases()
        patch_cases(_import_reference())
        return
    if len(sys.argv) > 1 and sys.argv[1] == "trainer":        # only the trainer-step fixture (same seeds / thread count as the full run)
        _install_cpu_aliases()
        _import_reference()
        torch.manual_seed(0)
        torch.set_num_threads(4)
        os.makedirs(OUT, exist_ok=True)
        run_trainer_case(sys.argv[2] if len(sys.argv) > 2 else "trainer_step")
        return
    _install_cpu_aliases()
    R = _import_reference()
    torch.manual_seed(0)
    torch.set_num_threads(4)
    os.makedirs(OUT, exist_ok=True)

    run_mask_cases(R, "masks")

    def feats(seed, B, C, h, signed=False):
        rs = np.random.RandomState(seed)
        x = rs.standard_normal((B, C, h, h)).astype(np.float32)
        if not signed:
            x = np.abs(x)
        ref = rs.rand(B, C, h, h).astype(np.float32)
        return x, ref

    x, ref = feats(101, 2, 16, 8)
    run_layer_case(R, "layer_c16_8x8_center", x, ref, center_mask(64, 16, 48))

    x, ref = feats(102, 2, 512, 8)          # BASELINE.json config 1
    run_layer_case(R, "layer_c512_8x8_cfg1", x, ref, center_mask(64, 16, 48), note="BASELINE config 1")

    x, ref = feats(103, 1, 32, 16)
    run_layer_case(R, "layer_c32_16x16_stroke", x, ref, stroke_mask(128, 3), triple_w=0.5, strength=0.7)

    x, ref = feats(104, 2, 8, 8)            # duplicated patches + duplicated ref columns: ties -> lowest index
    x[:, :, 1, :] = x[:, :, 0, :]
    x[:, :, 5, 2:6] = x[:, :, 4, 2:6]
    ref[:, :, 3, :] = ref[:, :, 2, :]
    run_layer_case(R, "layer_c8_8x8_ties", x, ref, center_mask(64, 16, 48), note="tie-breaking")

    x, ref = feats(105, 2, 16, 8, signed=True)
    run_layer_case(R, "layer_c16_8x8_signed", x, ref, center_mask(64, 16, 48),
                   note="ill-conditioned: signed features, compare with a relative metric")

    x, ref = feats(106, 1, 8, 8)
    run_layer_case(R, "layer_c8_8x8_nomask", x, ref, np.zeros((64, 64), np.uint8), note="M = 0")

    x, ref = feats(107, 1, 12, 8)
    run_layer_case(R, "layer_c12_8x8_fullmask", x, ref, np.ones((64, 64), np.uint8), note="M = N")

    x, ref = feats(108, 3, 20, 8)           # C not a multiple of 8, B = 3
    run_layer_case(R, "layer_c20_8x8_edge", x, ref, np.pad(np.ones((20, 64), np.uint8), ((0, 44), (0, 0))))

    # BASELINE.json config 2 shape for ONE sample (32x32x512, M=256); inputs are regenerated from the seed
    x, ref = big_case_inputs()
    run_layer_case(R, "layer_c512_32x32_cfg2", x, ref, center_mask(256, 64, 192),
                   keep_channels=np.arange(0, 512, 37), note="BASELINE config 2, one sample")

    patch_cases(R)
    run_innercos2_case(R, "innercos2_c1024_8x8")
    run_network_cases("networks")
    run_trainer_case("trainer_step")


if __name__ == "__main__":
    main()
