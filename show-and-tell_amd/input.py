"""Host-side input plumbing on either side of the hot path: the batch invariant the kernels rely
on, and an H2D prefetcher so the PCIe copy of batch i+1 hides under the step of batch i (PCIe note).

`collate_batch` keeps the contract of the reference's `collate_fn` (`/root/reference/data_loader.py:48-62`): samples
sorted by caption length, longest first (equal lengths keep their order), images stacked, captions zero-padded into
one int64 matrix, `lengths` as a Python list.  No GPU work happens here.
"""
import torch


def collate_batch(samples, pin_memory=False):
    """samples: iterable of (image [3,H,W] float tensor, caption 1-D integer tensor, image id).
    Returns (images [B,3,H,W], captions i64 [B,max_len] zero padded, lengths list[int] descending, image ids tuple)."""
    ordered = sorted(samples, key=lambda s: int(s[1].shape[0]), reverse=True)      # stable, like list.sort(reverse=True)
    if not ordered:
        raise ValueError("empty batch")
    lengths = [int(s[1].shape[0]) for s in ordered]
    if lengths[-1] < 1:
        raise ValueError("empty caption")
    images = torch.stack([s[0] for s in ordered], 0)
    captions = torch.nn.utils.rnn.pad_sequence([s[1].long() for s in ordered], batch_first=True, padding_value=0)
    if pin_memory:
        images, captions = images.pin_memory(), captions.pin_memory()
    return images, captions, lengths, tuple(s[2] for s in ordered)


def collate_on_device(images, flat_captions, lengths, image_ids=None):
    """`collate_fn` (data_loader.py:48-62) with the data movement on the MI355X: `images` f32 [B,3,H,W] and the ragged
    captions (`flat_captions` int64, samples back to back, dataset order) are already in HBM -- e.g. staged by
    `DevicePrefetcher` -- and `lengths` (host ints, dataset order) say where each caption ends.  Returns the reference's
    batch: images in decreasing-length order (ties keep dataset order), captions zero-padded [B, max_len], lengths list,
    image ids tuple.  The order itself is host integer work on lengths the host already holds: no device sync."""
    from . import _lib as L
    lib = L.load()
    L.require_gpu(images, "images")
    L.require_gpu(flat_captions, "flat_captions")
    lengths = [int(l) for l in lengths]
    B = len(lengths)
    if B == 0 or images.shape[0] != B:
        raise ValueError("one length per image")
    if min(lengths) < 1 or sum(lengths) != flat_captions.numel():
        raise ValueError("lengths must be >= 1 and sum to flat_captions.numel()")
    if images.dtype != torch.float32 or flat_captions.dtype != torch.int64:
        raise TypeError("images float32, captions int64")
    order = sorted(range(B), key=lambda i: -lengths[i])                       # stable: ties keep dataset order
    offs = [0]
    for l in lengths:
        offs.append(offs[-1] + l)
    dev = images.device
    order_d = torch.tensor(order, dtype=torch.int32, device=dev)
    offs_d = torch.tensor(offs, dtype=torch.int64, device=dev)
    tmax = lengths[order[0]]
    caps = torch.empty(B, tmax, dtype=torch.int64, device=dev)
    images = images.contiguous()
    out_im = torch.empty_like(images)
    st = L.stream()
    L.check(lib.sat_collate_captions(flat_captions.contiguous().data_ptr(), offs_d.data_ptr(), order_d.data_ptr(), B, tmax,
                                     caps.data_ptr(), st), "sat_collate_captions")
    L.check(lib.sat_gather_rows_f32(images.data_ptr(), order_d.data_ptr(), B, images[0].numel(), out_im.data_ptr(), st),
            "sat_gather_rows_f32")
    ids = tuple(image_ids[i] for i in order) if image_ids is not None else tuple(order)
    return out_im, caps, [lengths[i] for i in order], ids


class _Upcoming(list):
    """the list `DevicePrefetcher.upcoming_images()` returns; `.last`: the data ends inside it (`EncoderCNN.prefetch_many`)"""
    last = None


class DevicePrefetcher:
    """Wraps an iterable of (images, captions, lengths, ...) host batches: yields the same tuples with the two tensors
    resident on `device`; the copies of the next `depth` batches run on a side HIP stream while the caller works on the
    current one.  Pinned host tensors make the copies truly asynchronous (`collate_batch(..., pin_memory=True)` or a
    DataLoader with pin_memory).  `upcoming_images()` (inside the loop) returns the device image tensors of the batches
    that follow, in order -- what `TrainStep.step(..., next_images=...)` / `EncoderCNN.prefetch` want for the encoder
    look-ahead; the very same tensor objects are yielded later."""

    def __init__(self, batches, device, depth=None):
        """depth: batches staged ahead of the one being consumed; None = 6, the encoder's default look-ahead window
        (`EncoderCNN.lookahead_depth`: a smaller depth still works -- `upcoming_images()` tells the look-ahead whether the data
        ends inside the list -- but leaves run slots of the look-ahead empty)"""
        self.batches, self.device = batches, torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DevicePrefetcher copies to the MI355X; got device %s" % (device,))
        self.stream = torch.cuda.Stream(self.device)
        self.depth = max(1, int(depth if depth is not None else 6))
        self._queue = []          # [(staged batch tuple, copy-done event)]
        self._done = False        # the wrapped iterable is exhausted: everything that will ever come is staged

    def _stage(self, batch):
        images, captions = batch[0], batch[1]
        with torch.cuda.stream(self.stream):
            d_images = images.to(self.device, non_blocking=True)
            d_captions = captions.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return (d_images, d_captions) + tuple(batch[2:]), ev

    def upcoming_images(self, wait=False):
        """device image tensors of the staged batches that follow the one just yielded.  Their H2D copies may still be in
        flight.  wait=False (what the encoder look-ahead wants): each tensor carries its copy-done event (`_sat_ready_event`),
        which `EncoderCNN.prefetch` / `prefetch_many` make THEIR side stream wait for -- the caller's compute stream is not held
        up by copies of batches it will only consume later; it waits for a batch's copy when `__iter__` yields that
        batch.  Hand such tensors to nothing but the look-ahead.  wait=True: the current stream waits for every staged copy
        first, so the tensors are safe for ANY use on it."""
        out = _Upcoming()
        out.last = self._done                              # the data ends inside this list: a batch without a partner will not get one
        cur = torch.cuda.current_stream(self.device) if wait else None
        for staged, ev in self._queue:
            if wait:
                cur.wait_event(ev)
                staged[0].record_stream(cur)
            staged[0]._sat_ready_event = ev
            out.append(staged[0])
        return out

    def __iter__(self):
        it = iter(self.batches)
        self._queue = []
        self._done = False
        while True:
            while not self._done and len(self._queue) < self.depth + 1:      # the batch to yield + `depth` copies in flight
                try:
                    self._queue.append(self._stage(next(it)))
                except StopIteration:
                    self._done = True
            if not self._queue:
                return
            ready, ev = self._queue.pop(0)
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)                           # the staged copy is complete before the caller's kernels read it
            for t in ready[:2]:
                t.record_stream(cur)                     # allocator: these blocks are in use on the compute stream
            yield ready
