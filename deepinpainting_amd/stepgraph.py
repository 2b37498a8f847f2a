"""One training step as a HIP graph.

The body of the reference's training loop (train.ipynb, the cell that iterates the data loader) is four calls on the model,

    model.set_input(image, mask, ref); model.set_ref_latent(); model.set_gt_latent(); model.optimize_parameters()

(models/IPSR.py:120-152, :160-165, :184-189, :262-275 of the reference).  On an MI355X those calls queue ~1100-1400 kernels whose
GPU time (21-24 ms at the BASELINE batch sizes) is about what CPython + the dispatcher need to queue them (21-22 ms measured,
bench.py `host_enqueue_ms_per_step`): under bf16 autocast the host is the bound, in fp32 it is 10 % away from being it.  The step has
no data-dependent host decision once the mask is set (the hole's index list lives on the device, the optimizers are the fused Adam
with its step count on the device), so the whole sequence is recorded once (`hipStreamBeginCapture` through torch.cuda.graph) and
replayed with one `hipGraphLaunch`; the host's share of a step becomes three small copies into the graph's input tensors.

What is fixed inside a recording, and what makes `step()` record again:
  * the mask (the recording starts after the first eager step, i.e. with the feature mask and the index list built; a different
    mask tensor — identity or in-place version — records again);
  * the learning rates (kernel arguments of the fused Adam; the reference's schedulers change them once per epoch);
  * the image / reference shapes.
Random numbers (Dropout(0.5) of the U-Nets) come from torch's Philox generator, which registers with the capture: every replay
advances the generator exactly as the eager step does, so eager and replayed steps draw the same masks from the same seed.
"""
import torch


class StepGraph(object):
    def __init__(self, model, warmup=2, capture=True):
        if model.device.type != 'cuda':
            raise RuntimeError("StepGraph records a HIP graph: the model must live on an MI355X (model.device = %s)" % model.device)
        if getattr(model, '_reducer_D', None) is not None or getattr(model, '_reducer_G', None) is not None:
            raise RuntimeError("StepGraph: the data-parallel gradient exchange is not recorded (one process, one GPU only)")
        self.model = model
        self.warmup = int(warmup)
        self.capture = bool(capture)                    # False: the same warm-up / undo, then every step eagerly (the A/B of the tests)
        self.graph = None
        self.stream = torch.cuda.Stream(device=model.device)
        self.recordings = 0
        self._key = None
        self._img = self._ref = None

    # ---- what a recording depends on ----------------------------------------------------------------------------------------------
    def _optimizers(self):
        m = self.model
        return [m.optimizer_D, m.optimizer_F, m.optimizer_G, m.optimizer_P]

    def _state_key(self, img, mask, ref):
        lrs = tuple(float(g['lr']) for o in self._optimizers() for g in o.param_groups)
        return (tuple(img.shape), tuple(ref.shape), id(mask), mask._version, lrs)

    def _eager(self, mask):
        m = self.model
        m.set_input(self._img, mask, self._ref)
        m.set_ref_latent()
        m.set_gt_latent()
        m.optimize_parameters()

    def _net_state(self):
        """Parameters and buffers of the four trained nets (the frozen VGG and the mask structures are read only)."""
        m = self.model
        ts = []
        for net in (m.netG, m.netP, m.netD, m.netF):
            ts += [p for p in net.parameters()] + [b for b in net.buffers()]
        return ts

    @staticmethod
    def _adam_state(o):
        return [v for st in o.state.values() for v in st.values() if torch.is_tensor(v)]

    def _drop_autograd_graph(self):
        """The trainer keeps the last step's results (fake_B, the predictions, the losses, the InnerCos taps' `loss`), and through them
        that step's autograd graph with its AccumulateGrad nodes.  Those nodes remember the stream they were created on and every later
        forward reuses them while they live: after eager steps on the default stream the recorded backward would hop to the default
        stream in the middle of the capture (torch warns; hipStreamEndCapture then faults).  Detaching the kept results frees the graph,
        and the warm-up on the capture stream creates the nodes anew."""
        seen, objs = set(), [self.model]
        for v in vars(self.model).values():
            for mod in (v if isinstance(v, (list, tuple)) else [v]):
                if isinstance(mod, torch.nn.Module):
                    objs += list(mod.modules())
        for o in objs:
            if id(o) in seen:
                continue
            seen.add(id(o))
            for k, v in list(vars(o).items()):
                if torch.is_tensor(v) and v.grad_fn is not None:
                    setattr(o, k, v.detach())

    def _record(self, img, mask, ref):
        """Warm-up steps on the capture stream (MIOpen's solver look-ups, the per-stream workspaces, the Adam state's first
        allocation), undone afterwards: parameters, buffers, optimizer state and the generator go back to their values before the
        call, so recording trains nothing.  Then the capture proper, which executes nothing."""
        from . import _lib
        m = self.model
        for o in self._optimizers():
            for g in o.param_groups:
                g['capturable'] = True                  # fused Adam keeps `step` on the device either way; this lifts torch's capture check
        self._img, self._ref, self._mask = img.clone(), ref.clone(), mask
        _lib.lib().ipsr_profile_enable(0)               # region timing records HIP events on the launch stream: not inside a capture
        self._drop_autograd_graph()
        cur = torch.cuda.current_stream(m.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            live = self._net_state()
            saved = [t.detach().clone() for t in live]
            moments = [[t.clone() for t in self._adam_state(o)] if o.state else None for o in self._optimizers()]
            rng = torch.cuda.get_rng_state(m.device)
            for _ in range(max(self.warmup, 1)):
                self._eager(mask)
            with torch.no_grad():                       # in place on the parameters themselves: their version counters move, and the
                for t, s0 in zip(live, saved):          # filter caches keyed on them (models/hipconv.py) see new weights
                    t.copy_(s0)
            for o, mom in zip(self._optimizers(), moments):
                for i, t in enumerate(self._adam_state(o)):
                    if mom is None:
                        t.zero_()                       # a new optimizer: its moments and step count were created by the warm-up
                    else:
                        t.copy_(mom[i])
            torch.cuda.set_rng_state(rng, m.device)
        cur.wait_stream(self.stream)
        torch.cuda.synchronize(m.device)
        del saved, moments
        if self.capture:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self.stream):
                self._eager(mask)
        self.recordings += 1
        self._key = self._state_key(img, mask, ref)

    # ---- the loop body ----------------------------------------------------------------------------------------------------------------
    def step(self, img, mask, ref):
        """Same meaning as the four calls above on (img, mask, ref): one training step.  The first call (and any change listed in the
        module text) records first."""
        if self._key != self._state_key(img, mask, ref):
            self._record(img, mask, ref)
        if not self.capture:
            self._img.copy_(img, non_blocking=True)
            self._ref.copy_(ref, non_blocking=True)
            self._eager(mask)
            return
        # The replay runs on the stream it was recorded on, fenced by events on both sides: a hipGraphLaunch queued on the default stream was
        # not always ordered against the work queued after it there (the next replay's input copies, a host read of the losses), and two
        # overlapping replays of one recording share every intermediate buffer.
        cur = torch.cuda.current_stream(self.model.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self._img.copy_(img, non_blocking=True)
            self._ref.copy_(ref, non_blocking=True)
            self.graph.replay()
        cur.wait_stream(self.stream)

    __call__ = step
