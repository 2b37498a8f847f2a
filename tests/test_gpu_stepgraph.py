"""The training step replayed from one HIP graph (deepinpainting_amd/stepgraph.py) against the same step queued from Python.

The reference's loop body (train.ipynb: set_input, set_ref_latent, set_gt_latent, optimize_parameters; models/IPSR.py:120-275) has no
counterpart of the graph, so the oracle here is the eager step of this package, which tests/test_gpu_model.py pins to the
reference's fixtures.  Batch 1 with MIOpen's deterministic solvers is bitwise repeatable (tools/exp_repeatability.py; at batch 8
MIOpen's transposed-convolution forward is not, tools/exp_first_divergence.py), so the comparison is exact.
"""
import contextlib
import io

import numpy as np
import pytest
import torch

import golden_cases

pytestmark = pytest.mark.gpu


def _model(tmp, dropout, amp=False):
    from deepinpainting_amd.options import Option
    from deepinpainting_amd.models.models import create_model
    opt = Option(gpu_ids=[0], batchSize=1, use_dropout=dropout, quiet=True, allow_random_vgg=True, amp_bf16=amp, checkpoints_dir=str(tmp))
    with contextlib.redirect_stdout(io.StringIO()):
        m = create_model(opt)
    for i, net in enumerate((m.netG, m.netP, m.netD, m.netF, m.vgg)):
        golden_cases.reinit_deterministic(net, 700 + i)
    return m


def _data(i):
    g = torch.Generator(device="cuda").manual_seed(40 + i)
    img = torch.rand(1, 3, 256, 256, device="cuda", generator=g) * 2 - 1
    ref = torch.rand(1, 3, 256, 256, device="cuda", generator=g) * 2 - 1
    return img, ref


def _weights(m):
    return torch.cat([p.detach().flatten() for net in (m.netG, m.netP, m.netD, m.netF) for p in net.parameters()]).clone()


@pytest.fixture
def deterministic_miopen():
    was = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    yield
    torch.backends.cudnn.deterministic = was


@pytest.mark.parametrize("dropout", [False, True])
def test_replayed_steps_equal_eager_steps_bit_for_bit(tmp_path, deterministic_miopen, dropout):
    """Four steps on four different batches: plain eager calls, StepGraph(capture=False) (same warm-up and undo, eager steps) and the
    replayed graph give the same losses and the same weights, bit for bit — with Dropout(0.5) on, too: the replay advances torch's
    Philox generator exactly as the eager step does."""
    from deepinpainting_amd.stepgraph import StepGraph
    mask = torch.zeros(1, 1, 256, 256, dtype=torch.bool, device="cuda")
    mask[:, :, 64:192, 64:192] = 1
    got = {}
    for mode in ("eager", "undo", "graph"):
        m = _model(tmp_path / mode, dropout)
        torch.cuda.manual_seed(4242)
        sg = None if mode == "eager" else StepGraph(m, capture=mode == "graph")
        losses = []
        for i in range(4):
            img, ref = _data(i)
            if sg is None:
                m.set_input(img, mask, ref)
                m.set_ref_latent()
                m.set_gt_latent()
                m.optimize_parameters()
            else:
                sg.step(img, mask, ref)
            e = m.get_current_errors()
            losses.append([e[k] for k in ("G_GAN", "G_L1", "D", "F")] + [float(m.ng_loss_value), float(m.ng_loss_value2)])
        assert np.isfinite(losses).all()
        if sg is not None:
            assert sg.recordings == 1
        got[mode] = (losses, _weights(m), m.fake_B.detach().clone())
        del m, sg
        torch.cuda.empty_cache()
    for other in ("undo", "graph"):
        assert got[other][0] == got["eager"][0], (other, got[other][0], got["eager"][0])
        assert torch.equal(got[other][1], got["eager"][1]), other
        assert torch.equal(got[other][2], got["eager"][2]), other


def test_queued_replays_do_not_overlap(tmp_path, deterministic_miopen):
    """Twelve steps queued back to back with no host read in between (bench.py's timed loop): the replays must execute one after the
    other — they share every intermediate buffer.  (Launched on the default stream they did not always: losses read as leftovers of
    other tensors, then NaN weights; StepGraph.step fences the replay on its own stream.)  Same weights as twelve eager steps."""
    from deepinpainting_amd.stepgraph import StepGraph
    mask = torch.zeros(1, 1, 256, 256, dtype=torch.bool, device="cuda")
    mask[:, :, 64:192, 64:192] = 1
    got = {}
    for mode in ("eager", "graph"):
        m = _model(tmp_path / mode, True)
        torch.cuda.manual_seed(99)
        img, ref = _data(0)
        for _ in range(2):                               # eager steps on the default stream first, as bench.py's warm-up
            m.set_input(img, mask, ref); m.set_ref_latent(); m.set_gt_latent(); m.optimize_parameters()
        sg = StepGraph(m) if mode == "graph" else None
        for i in range(12):
            if sg is None:
                m.set_input(img, mask, ref); m.set_ref_latent(); m.set_gt_latent(); m.optimize_parameters()
            else:
                sg.step(img, mask, ref)
        e = m.get_current_errors()
        got[mode] = ([e[k] for k in ("G_GAN", "G_L1", "D", "F")], _weights(m))
        del m, sg
        torch.cuda.empty_cache()
    assert np.isfinite(got["graph"][0]).all() and got["graph"][0] == got["eager"][0]
    assert torch.equal(got["graph"][1], got["eager"][1])


def test_recording_trains_nothing_and_follows_the_learning_rate(tmp_path, deterministic_miopen):
    """`_record` runs warm-up steps and undoes them: weights, Adam moments and step counts are what they were.  A learning-rate change
    (the reference's per-epoch scheduler, models/base_model.py update_learning_rate) and a new mask tensor each record again; the
    replay then uses the new rate (weights equal an eager model's that made the same change)."""
    from deepinpainting_amd.stepgraph import StepGraph
    mask = torch.zeros(1, 1, 256, 256, dtype=torch.bool, device="cuda")
    mask[:, :, 64:192, 64:192] = 1
    img, ref = _data(0)

    m = _model(tmp_path / "a", False)
    m.set_input(img, mask, ref); m.set_ref_latent(); m.set_gt_latent(); m.optimize_parameters()          # Adam state exists now
    w0 = _weights(m)
    st0 = [v.clone() for o in (m.optimizer_D, m.optimizer_F, m.optimizer_G, m.optimizer_P) for s in o.state.values() for v in s.values() if torch.is_tensor(v)]
    sg = StepGraph(m)
    sg._record(img, mask, ref)
    assert torch.equal(_weights(m), w0)
    st1 = [v for o in (m.optimizer_D, m.optimizer_F, m.optimizer_G, m.optimizer_P) for s in o.state.values() for v in s.values() if torch.is_tensor(v)]
    assert len(st0) == len(st1) and all(torch.equal(a, b) for a, b in zip(st0, st1))
    sg.step(img, mask, ref)
    assert sg.recordings == 1 and not torch.equal(_weights(m), w0)
    for o in (m.optimizer_D, m.optimizer_F, m.optimizer_G, m.optimizer_P):
        for g in o.param_groups:
            g['lr'] = g['lr'] * 0.5
    sg.step(img, mask, ref)
    assert sg.recordings == 2
    mask2 = mask.clone()
    mask2[:, :, 64:192, 64:128] = 0                                                                       # a smaller hole: another index list
    sg.step(img, mask2, ref)
    assert sg.recordings == 3 and m.CSA_model[0].mask_point_idx.numel() == 128
    wg = _weights(m)

    e = _model(tmp_path / "b", False)
    for k, msk in enumerate((mask, mask, mask, mask2)):
        if k == 2:
            for o in (e.optimizer_D, e.optimizer_F, e.optimizer_G, e.optimizer_P):
                for g in o.param_groups:
                    g['lr'] = g['lr'] * 0.5
        e.set_input(img, msk, ref); e.set_ref_latent(); e.set_gt_latent(); e.optimize_parameters()
    assert torch.equal(wg, _weights(e))


def test_step_graph_refuses_what_it_cannot_record(tmp_path):
    from deepinpainting_amd.stepgraph import StepGraph
    m = _model(tmp_path, False)
    m._reducer_D = object()
    with pytest.raises(RuntimeError, match="gradient exchange"):
        StepGraph(m)
