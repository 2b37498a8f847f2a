"""World-size-2 `gloo` test (CPU) of the data-parallel gradient exchange (deepinpainting_amd/dist.py) —
the N>1 path of bench.py / the trainer.  On the GPU box the same code runs over RCCL ("nccl")."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_nets(seed):
    torch.manual_seed(seed)
    netA = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.InstanceNorm2d(8, affine=True), nn.ReLU(), nn.Conv2d(8, 4, 3, padding=1))
    netB = nn.Sequential(nn.Conv2d(4, 6, 3, padding=1), nn.ReLU(), nn.Conv2d(6, 1, 3, padding=1))
    unused = nn.Linear(5, 5)      # never takes part in the loss: its bucket is only flushed by finish()
    return netA, netB, unused


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from deepinpainting_amd import dist as idist
    r, w, _ = idist.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    # different init per rank -> broadcast makes them equal to rank 0's
    netA, netB, unused = _make_nets(100 + rank)
    for n in (netA, netB, unused):
        idist.broadcast_module(n, src=0)
    ref = _make_nets(100)
    for n, m in zip((netA, netB, unused), ref):
        for p, q in zip(n.parameters(), m.parameters()):
            assert torch.equal(p, q)
    # tiny bucket size -> several buckets, exercised out of registration order
    red_A = idist.GradBucketReducer([netA, unused], bucket_bytes=256)
    red_B = idist.GradBucketReducer([netB], bucket_bytes=1 << 20)
    assert len(red_A.buckets) > 2
    torch.manual_seed(1000 + rank)
    x = torch.randn(2, 3, 8, 8)
    # step 1: only netB's reducer armed (like backward_D: grads deposited in netA must NOT be exchanged)
    red_B.arm()
    loss = netB(netA(x)).pow(2).mean()
    loss.backward()
    red_B.finish()
    gA_local = [p.grad.clone() for p in netA.parameters()]
    gB_avg = [p.grad.clone() for p in netB.parameters()]
    # step 2: netA's reducer armed, grads of netA accumulate on top (no zero_grad) -> exchanged sum
    red_A.arm()
    loss = netB(netA(x)).pow(2).mean()
    loss.backward()
    red_A.finish()
    gA_after = [p.grad.clone() for p in netA.parameters()]
    torch.save(dict(gA_local=gA_local, gB_avg=gB_avg, gA_after=gA_after,
                    unused_grad=[p.grad.clone() for p in unused.parameters()],
                    scal=idist.all_reduce_mean_scalars([float(rank), 2.0], torch.device("cpu"))),
               os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bucketed_allreduce_world2(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r)) for r in range(world)]
    # single-process recomputation of both ranks' local gradients
    local = []
    for rank in range(world):
        netA, netB, _ = _make_nets(100)
        torch.manual_seed(1000 + rank)
        x = torch.randn(2, 3, 8, 8)
        netB(netA(x)).pow(2).mean().backward()
        local.append(([p.grad.clone() for p in netA.parameters()], [p.grad.clone() for p in netB.parameters()]))
    for rank in range(world):
        # un-armed net keeps its LOCAL gradients
        for g, want in zip(res[rank]["gA_local"], local[rank][0]):
            torch.testing.assert_close(g, want, rtol=1e-6, atol=1e-8)
        # armed net holds the rank MEAN
        for i, g in enumerate(res[rank]["gB_avg"]):
            torch.testing.assert_close(g, (local[0][1][i] + local[1][1][i]) / 2, rtol=1e-5, atol=1e-8)
        # second backward accumulated local grads (2x) and then exchanged: mean over ranks of 2*local
        for i, g in enumerate(res[rank]["gA_after"]):
            torch.testing.assert_close(g, (2 * local[0][0][i] + 2 * local[1][0][i]) / 2, rtol=1e-5, atol=1e-8)
        # parameters that never received a gradient get an (all-zero) averaged gradient, not a hang
        for g in res[rank]["unused_grad"]:
            assert float(g.abs().max()) == 0.0
        assert res[rank]["scal"] == [0.5, 2.0]
    # both ranks end with identical exchanged gradients
    for a, b in zip(res[0]["gA_after"], res[1]["gA_after"]):
        assert torch.equal(a, b)


class _SinkLinear(torch.autograd.Function):
    """y = x @ w^T whose weight gradient is written where models/hipconv.py's kernels write theirs: into the armed bucket slice
    handed out by dist.grad_sink_for (out=sink OVERWRITES, like the HIP weight-gradient kernels), else into a fresh tensor."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w.t()

    @staticmethod
    def backward(ctx, dy):
        from deepinpainting_amd import dist as idist
        x, w = ctx.saved_tensors
        sink = idist.grad_sink_for(w.data_ptr(), w.shape)
        dw = sink if sink is not None else torch.empty_like(w)
        torch.mm(dy.t(), x, out=dw)
        return dy @ w, dw


def _sink_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from deepinpainting_amd import dist as idist
    idist.init_distributed(backend="gloo")
    torch.manual_seed(7)
    once = nn.Linear(6, 5, bias=False)          # applied once per backward  (netG / netP layers)
    twice = nn.Linear(5, 5, bias=False)         # applied to two batches in ONE graph (netD / netF in backward_D)
    red = idist.GradBucketReducer([once, twice], bucket_bytes=1 << 20)
    torch.manual_seed(50 + rank)
    xa, xb = torch.randn(4, 6), torch.randn(4, 6)
    out = {}
    for step in range(2):                        # second step: .grad starts as None again (zero_grad(set_to_none=True))
        for p in list(once.parameters()) + list(twice.parameters()):
            p.grad = None
        idist.SINK_STATS.update(handed=0, copied=0, in_place=0)
        red.arm()
        ha, hb = _SinkLinear.apply(xa, once.weight), _SinkLinear.apply(xb, once.weight.detach())
        loss = (_SinkLinear.apply(ha, twice.weight) * 1.5).pow(2).mean() + _SinkLinear.apply(hb, twice.weight).pow(2).mean()
        loss.backward()
        flat = red.buckets[0]["flat"]
        lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
        inside = [lo <= p.grad.data_ptr() < hi for p in (once.weight, twice.weight)]
        stats_before_finish = dict(idist.SINK_STATS)
        red.finish()
        out["step%d" % step] = dict(g_once=once.weight.grad.clone(), g_twice=twice.weight.grad.clone(), inside=inside,
                                    stats=dict(idist.SINK_STATS), stats_before_finish=stats_before_finish)
    # what plain autograd gives on this rank's data
    once.weight.grad = twice.weight.grad = None
    ha, hb = xa @ once.weight.t(), xb @ once.weight.detach().t()
    ((ha @ twice.weight.t() * 1.5).pow(2).mean() + (hb @ twice.weight.t()).pow(2).mean()).backward()
    out["local"] = dict(g_once=once.weight.grad.clone(), g_twice=twice.weight.grad.clone())
    torch.save(out, os.path.join(out_dir, "sink_rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_gradient_sinks_hand_over_ownership_and_survive_a_module_applied_twice(tmp_path):
    """A weight gradient written into its bucket slice (what the HIP weight-gradient kernels do under DDP) must (a) BE the
    parameter's .grad afterwards — autograd adopts the tensor, the reducer copies nothing for it — and (b) stay correct when the
    module is applied twice in one graph (netD / netF in backward_D): the slice is handed out once, the second backward node
    writes its own tensor and autograd adds the two.  The exchanged result is compared with the true mean of the ranks' plain
    autograd gradients."""
    world, port = 2, _free_port()
    mp.spawn(_sink_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(str(tmp_path), "sink_rank%d.pt" % r)) for r in range(world)]
    for key in ("g_once", "g_twice"):
        want = (res[0]["local"][key] + res[1]["local"][key]) / 2
        for r in range(world):
            for step in ("step0", "step1"):
                torch.testing.assert_close(res[r][step][key], want, rtol=1e-5, atol=1e-7)
    for r in range(world):
        for step in ("step0", "step1"):
            st = res[r][step]
            # applied once: the kernel's output IS .grad (adopted, not cloned) and the reducer copies nothing for it
            assert st["inside"][0], "a sink-written gradient does not live in its bucket slice"
            assert st["stats_before_finish"]["handed"] == 2          # one hand-over per parameter, not per backward node
            # applied twice: autograd sums the two nodes' tensors where it likes; that sum is copied in iff it is not in the slice
            assert st["stats"]["copied"] == (0 if st["inside"][1] else 1), st["stats"]
            assert st["stats"]["in_place"] == 2 - st["stats"]["copied"], st["stats"]


def test_single_process_reducer_is_inert():
    sys.path.insert(0, ROOT)
    from deepinpainting_amd import dist as idist
    netA, _, _ = _make_nets(1)
    red = idist.GradBucketReducer([netA])
    assert red.world == 1 and not red._hooks
    red.arm()
    netA(torch.randn(1, 3, 8, 8)).sum().backward()
    red.finish()
    assert all(p.grad is not None for p in netA.parameters())
    assert idist.init_distributed() == (0, 1, 0)
