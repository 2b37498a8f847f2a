"""Input staging for the training loop: pinned host buffers + asynchronous host-to-device copies on a
side stream, the notebook's mask preparation done on the device, and a device-side free-form mask generator.

The reference's loop (train.ipynb cell 2) does `image.cuda()`, `mask.cuda()`, slices `mask[0][0]` and rebuilds a
[1,1,H,W] bool tensor on every iteration, synchronously, from pageable memory.  `DeviceStager` overlaps those copies
with the previous step's kernels; the step itself is unchanged (`model.set_input(image, mask, ref)`).
"""
import torch


class DeviceStager(object):
    """Wraps an iterator of (image, mask, ref) CPU batches; yields (image, mask, ref) on `device` with
    mask = [1,1,H,W] bool (first sample, first channel — reference semantics: one mask per batch), prefetching one
    batch ahead through pinned memory on its own stream."""

    def __init__(self, iterable, device):
        self.it = iter(iterable)
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == 'cuda' else None
        self._next = None
        self._prefetch()

    def _to_device(self, t):
        if self.stream is None:
            return t.to(self.device)
        if not t.is_pinned():
            t = t.contiguous().pin_memory()
        return t.to(self.device, non_blocking=True)

    def _prefetch(self):
        try:
            image, mask, ref = next(self.it)
        except StopIteration:
            self._next = None
            return
        if self.stream is None:
            self._next = (image.to(self.device), prepare_mask(mask.to(self.device)), ref.to(self.device))
            return
        with torch.cuda.stream(self.stream):
            image, mask, ref = self._to_device(image), self._to_device(mask), self._to_device(ref)
            self._next = (image, prepare_mask(mask), ref)

    def __iter__(self):
        return self

    def __next__(self):
        if self._next is None:
            raise StopIteration
        if self.stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
            for t in self._next:
                t.record_stream(torch.cuda.current_stream(self.device))
        batch = self._next
        self._prefetch()
        return batch


def prepare_mask(mask_batch):
    """train.ipynb cell 2: mask[0][0] -> [1,1,H,W] bool."""
    m = mask_batch[0][0] if mask_batch.dim() == 4 else mask_batch
    return (m != 0)[None, None]


def random_stroke_mask(size, generator=None, device='cpu', strokes=(4, 12), width=(12, 40), area=(0.20, 0.40), max_tries=50):
    """Free-form mask [1,1,size,size] bool built on `device`: random-walk brush strokes, accepted when the masked area
    is within `area` (the reference's create_gMask gate: 20 % < area < maxPartition, util/util.py:54-55).  All random
    numbers come from `generator` (a CPU torch.Generator) so that every data-parallel rank can draw its own masks
    reproducibly; the rasterisation itself runs on the device."""
    g = generator
    ys = torch.arange(size, device=device).view(size, 1)
    xs = torch.arange(size, device=device).view(1, size)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=g))

    mask = torch.zeros(size, size, dtype=torch.bool, device=device)
    for _ in range(max_tries):
        mask.zero_()
        for _s in range(ri(*strokes)):
            y, x, wd = ri(0, size - 1), ri(0, size - 1), ri(*width)
            for _seg in range(ri(3, 8)):
                ny = min(max(y + ri(-size // 4, size // 4), 0), size - 1)
                nx = min(max(x + ri(-size // 4, size // 4), 0), size - 1)
                # thick segment (y,x)->(ny,nx): distance of every pixel to the segment <= wd/2
                dy, dx = float(ny - y), float(nx - x)
                den = max(dy * dy + dx * dx, 1.0)
                t = (((ys - y) * dy + (xs - x) * dx) / den).clamp_(0.0, 1.0)
                dist2 = (ys - (y + t * dy)) ** 2 + (xs - (x + t * dx)) ** 2
                mask |= dist2 <= (wd * 0.5) ** 2
                y, x = ny, nx
        frac = float(mask.float().mean())
        if area[0] < frac < area[1]:
            break
    return mask[None, None]
