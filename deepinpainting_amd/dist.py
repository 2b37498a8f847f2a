"""Data-parallel gradient exchange for the IPSR trainer: one process per GPU, RCCL over xGMI.

The reference is single-GPU (no torch.distributed anywhere); this is the only exchange step
data parallelism needs: the gradients of (netD, netF) after backward_D and of (netG, netP) after
backward_G are summed over ranks and divided by the world size (models/IPSR.py:271-278 is where the
optimizer steps consume them).  Gradients that backward_G deposits in netD/netF are dropped by the next
zero_grad and therefore NOT exchanged.

Design for the MI355X node (8 GPUs, full xGMI mesh, 7 links x ~153 GB/s per GPU):
  * parameters are packed — in reverse registration order, roughly the order autograd produces their
    gradients — into a few large flat fp32 buckets (64 MiB default: 528 MB of G+P gradients -> 8-9
    all-reduces, each big enough to be bandwidth- not latency-bound on the mesh);
  * a bucket's all-reduce is launched asynchronously from the post-accumulate-grad hook of its last
    parameter, i.e. while autograd is still producing earlier layers' gradients (overlap with backward);
  * the 1/world scaling is applied to the bucket before its collective (under the backward); `finish()` only waits and
    points every .grad at its slice of the reduced bucket (no copy back).
Works with any torch.distributed backend ("nccl" == RCCL on ROCm; "gloo" for the CPU tests).

RCCL algorithm / protocol.  An MI355X node is a full mesh of point-to-point xGMI links (7 x ~153 GB/s per GPU), not a
switch: a ring all-reduce moves every byte over ONE link per hop and is bound by that single link, while the
mesh-aware forms (RCCL's direct reduce-scatter + all-gather, one chunk per peer over all 7 links at once) use the whole
fabric.  RCCL picks per message size; `init_distributed(..., rccl_algo=..., rccl_proto=..., rccl_channels=...)` (bench.py:
`--rccl-algo`, `--rccl-proto`, `--rccl-min-channels`) pins the choice through NCCL_ALGO / NCCL_PROTO /
NCCL_MIN_NCHANNELS before the communicator is created — they are read once, at creation.  Intended setting for the
64 MiB gradient buckets on the 8-GPU mesh: leave NCCL_ALGO unset (RCCL's tuner already prefers its direct/mesh kernels
intra-node), NCCL_PROTO=Simple (the LL protocols halve payload per flit and only pay off below ~1 MiB), and
NCCL_MIN_NCHANNELS >= 28 (4 channels per link) so that every link carries traffic.  None of this has been measured on
an 8-GPU node by this repo (the builder has one GPU; the driver runs the scaling bench) — the knobs exist so that the
measurement can be made without code changes.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None, rccl_algo=None, rccl_proto=None, rccl_channels=None):
    """Initialise torch.distributed from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_ADDR / MASTER_PORT).  Returns (rank, world_size, local_rank); a no-op for single-process runs.
    rccl_algo / rccl_proto / rccl_channels: see the module docstring (explicit environment settings win)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # this pool's driver only supports dmabuf IPC
            for var, val in (("NCCL_ALGO", rccl_algo), ("NCCL_PROTO", rccl_proto), ("NCCL_MIN_NCHANNELS", rccl_channels)):
                if val not in (None, ""):
                    os.environ.setdefault(var, str(val))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def broadcast_module(module, src=0, group=None):
    """Make every rank start from rank `src`'s parameters and buffers (one flat broadcast per dtype)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers() if b.is_floating_point()]
    if not tensors:
        return
    flat = torch.cat([t.reshape(-1).to(torch.float32) for t in tensors])
    dist.broadcast(flat, src=src, group=group)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t).to(t.dtype))
        off += n


# Gradient sinks: parameter storage pointer -> its slice of an armed bucket.  A kernel that produces a weight gradient can write it
# THERE (models/hipconv.py does for the Winograd / small-map weight-gradient kernels).  Ownership is HANDED OVER: `grad_sink_for`
# pops the entry and returns the only reference to that view, so
#   * autograd's AccumulateGrad sees a tensor nobody else references and adopts it as .grad instead of cloning it (it clones whenever
#     another reference exists) — the reducer then finds the gradient already in its slice and copies nothing;
#   * a SECOND backward node of the same parameter inside one graph (netD / netF are applied to the fake and the real batch in one
#     backward_D, models/IPSR.py:187-206) gets None, writes a tensor of its own, and autograd adds the two: handing the same slice
#     out twice would make the second kernel overwrite the first result and autograd sum two aliases of it (2 x dW_second).
_GRAD_SINKS = {}
SINK_STATS = {"handed": 0, "copied": 0, "in_place": 0}      # test / diagnostics counters (per process)


def grad_sink_for(param_data_ptr, shape):
    """The bucket slice a gradient of this parameter should be written into right now, or None.  At most once per arm()."""
    v = _GRAD_SINKS.pop(param_data_ptr, None)
    if v is None or tuple(v.shape) != tuple(shape):
        return None
    SINK_STATS["handed"] += 1
    return v


_SLOT_ALIGN = 64        # floats: every parameter's slice of a bucket starts on a 256-byte boundary (the weight-gradient kernels
                        # that write straight into it store 16-byte vectors)


def _slot(p):
    return (p.numel() + _SLOT_ALIGN - 1) // _SLOT_ALIGN * _SLOT_ALIGN


class GradBucketReducer(object):
    """Bucketed, overlapped all-reduce(mean) of the gradients of a fixed set of modules."""

    def __init__(self, modules, bucket_bytes=64 << 20, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        params = [p for m in modules for p in m.parameters() if p.requires_grad]
        params.reverse()
        self.buckets = []          # list of dicts: params, numel, flat
        cur, cur_n = [], 0
        cap = max(1, bucket_bytes // 4)
        for p in params:
            if cur and cur_n + _slot(p) > cap:
                self.buckets.append({"params": cur, "numel": cur_n})
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += _slot(p)
        if cur:
            self.buckets.append({"params": cur, "numel": cur_n})
        self._bucket_of = {}
        for bi, b in enumerate(self.buckets):
            b["flat"] = None
            for p in b["params"]:
                self._bucket_of[p] = bi
        self.armed = False
        self._pending = None
        self._work = []
        self._hooks = []
        # measurement (bench.py --gpus N): with `timing` on, finish() brackets its waits — HIP events on the current stream for
        # device buckets (work.wait() makes THAT stream wait for the collective, so the bracket is the time the compute stream
        # stalls after the backward's last kernel: the EXPOSED part of the all-reduce), the host clock for CPU buckets (gloo)
        self.timing = False
        self._exposed = []
        if self.world > 1:
            for p in params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # -- life cycle of one backward pass ------------------------------------------------------------
    def arm(self):
        """Call right before the backward whose gradients should be exchanged."""
        self.armed = self.world > 1
        if self.armed:
            # publish the bucket slices of parameters that have no gradient yet (a fresh .grad can be produced in place)
            for b in self.buckets:
                dev = b["params"][0].device
                fresh = b["flat"] is None or b["flat"].device != dev
                if fresh:
                    b["flat"] = torch.empty(b["numel"], dtype=torch.float32, device=dev)
                if fresh:
                    b["flat"].zero_()                    # the padding between slices is summed too: keep it finite
                off = 0
                for p in b["params"]:
                    n = p.numel()
                    if p.grad is None:
                        _GRAD_SINKS[p.data_ptr()] = b["flat"][off:off + n].view_as(p)
                    off += _slot(p)
        self._pending = [len(b["params"]) for b in self.buckets]
        self._seen = set()
        self._launched = set()
        self._work = []

    def _on_grad(self, p):
        if not self.armed:
            return
        bi = self._bucket_of[p]
        if bi in self._launched:
            # the bucket already went out with this parameter's gradient in it: a second accumulation (two backward passes
            # through the same parameter while armed, e.g. an extra InnerCos.backward(retain_graph=True)) would be lost
            raise RuntimeError("GradBucketReducer: a gradient arrived for a bucket that is already being all-reduced — "
                               "run every backward of the exchanged nets between ONE arm() / finish() pair, or arm() again")
        if id(p) in self._seen:          # accumulated twice before its bucket was complete: counted once
            return
        self._seen.add(id(p))
        self._pending[bi] -= 1
        assert self._pending[bi] >= 0
        if self._pending[bi] == 0:
            self._launch(bi)

    def _launch(self, bi):
        self._launched.add(bi)
        b = self.buckets[bi]
        dev = b["params"][0].device
        if b["flat"] is None or b["flat"].device != dev:
            b["flat"] = torch.zeros(b["numel"], dtype=torch.float32, device=dev)
        flat = b["flat"]
        views, off = [], 0
        for p in b["params"]:
            n = p.numel()
            views.append(flat[off:off + n].view_as(p))
            off += _slot(p)
        # a parameter whose gradient already LIVES in its bucket slice (a kernel that wrote it there, or .grad left as
        # the view by the previous finish() and accumulated in place) needs no copy; a parameter without a gradient on
        # this rank contributes zeros (another rank may have one: every rank must issue the same collective)
        src, dst = [], []
        for p, v in zip(b["params"], views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
            else:
                SINK_STATS["in_place"] += 1
        if dst:
            SINK_STATS["copied"] += len(dst)
            torch._foreach_copy_(dst, src)
        flat.mul_(1.0 / self.world)          # the mean's scaling happens here, under the backward, not after the wait
        self._work.append((bi, dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True), views))

    def finish(self):
        """Call after the backward, before optimizer.step(): waits for the collectives and writes the
        rank-averaged gradients back.  Buckets whose hooks never fired (unused parameters) are reduced
        here synchronously so every rank issues the same collectives in the same order."""
        if not self.armed:
            return
        launched = {bi for bi, _, _ in self._work}
        late = [bi for bi in range(len(self.buckets)) if bi not in launched]
        mark = self._timing_mark() if self.timing else None
        for bi in late:
            self._launch(bi)
        for bi, work, views in self._work:
            work.wait()
            # the rank-averaged gradients stay where the collective left them: .grad becomes a view of the bucket (no copy
            # back).  The bucket is rewritten only by the next armed backward, after the optimizer has consumed these.
            # NB a parameter that had no gradient on ANY rank ends up with a zero gradient (not None): Adam then still
            # applies its momentum to it, where a single-process run would skip it.  No parameter of the four nets is
            # ever without a gradient in optimize_parameters(), so the trainer is unaffected.
            for p, v in zip(self.buckets[bi]["params"], views):
                p.grad = v
        if mark is not None:
            self._exposed.append((mark, self._timing_mark(), len(late)))
        self._work = []
        self.armed = False
        for b in self.buckets:
            for p in b["params"]:
                _GRAD_SINKS.pop(p.data_ptr(), None)

    def _timing_mark(self):
        dev = self.buckets[0]["params"][0].device if self.buckets else torch.device("cpu")
        if dev.type == "cuda":
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(dev))
            return ev
        import time
        return time.perf_counter()

    def exposed_ms(self, reset=True):
        """Per finish() call since the last reset: milliseconds the consumer waited for the collectives after the backward
        (see `timing`).  Synchronises the device."""
        out = []
        for a, b, _ in self._exposed:
            if isinstance(a, float):
                out.append((b - a) * 1e3)
            else:
                b.synchronize()
                out.append(a.elapsed_time(b))
        if reset:
            self._exposed = []
        return out

    def bytes_per_exchange(self):
        return sum(4 * b["numel"] for b in self.buckets)

    def close(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def all_reduce_mean_scalars(values, device, group=None):
    """Average a few python floats over ranks (logging only)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return list(values)
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, group=group)
    return (t / dist.get_world_size(group)).tolist()
