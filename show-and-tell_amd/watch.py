"""Device-side status words of kernels that need ALL their workgroups resident at once (the persistent LSTM recurrence,
`sat_lstm_persist.hip`; the fused conv3 + BatchNorm + add + ReLU launch with its grid-wide statistics barrier,
`sat_conv3_fused.hip`).  Every wait inside such a kernel is bounded; when one runs out (co-tenants on the device kept part of the
grid from being scheduled) the kernel sets a sticky status word and leaves -- the outputs of that call are garbage.  That must
never pass silently: the word is copied to pinned host memory behind every call and looked at when the next call is
submitted (or at once with `poll(block=True)`; `TrainStep.check_ids()` does); non-zero raises RuntimeError after the caller's
`on_error` hook has switched the process to the form that needs no co-residency."""
import torch


class ResidencyWatch:
    DEPTH = 16
    _by_device = {}

    @classmethod
    def get(cls, device):
        key = str(device)
        w = cls._by_device.get(key)
        if w is None:
            w = cls._by_device[key] = cls()
        return w

    def __init__(self):
        self.host = torch.zeros(self.DEPTH, dtype=torch.int32).pin_memory()
        self.pending = []           # (slot, event, what, on_error), oldest first
        self.slot = 0

    # what an exception says about the model's state when the caller gave no note of its own: the status word is seen one call
    # later at the latest, and whatever consumed the call's outputs in between -- a torch optimizer step after
    # `loss.backward()` -- has used them.  `trainer.TrainStep` gates its update on the device instead and says so.
    DEFAULT_NOTE = ("the outputs of that call are invalid, and anything computed from them since -- an optimizer step taken "
                    "after loss.backward() included -- has used them: restore the parameters from a checkpoint")

    def submit(self, word, what, on_error=None, note=None):
        """word: a 1-element int32 device view of the status word the call that was just enqueued on the current stream may set;
        note: what the exception tells the user about parameters / optimizer state (DEFAULT_NOTE)"""
        self.poll(block=False)
        while len(self.pending) >= self.DEPTH:
            self._retire(block=True)
        slot = self.slot
        self.slot = (slot + 1) % self.DEPTH
        self.host[slot:slot + 1].copy_(word, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending.append((slot, ev, what, (on_error, note)))

    def _retire(self, block):
        slot, ev, what, (on_error, note) = self.pending[0]
        if block:
            ev.synchronize()
        elif not ev.query():
            return False
        self.pending.pop(0)
        code = int(self.host[slot])
        if code != 0:
            torch.cuda.synchronize()
            self.host.zero_()
            self.pending.clear()
            if on_error is not None:
                on_error()
            raise RuntimeError("show-and-tell_amd: %s timed out waiting for its workgroups to be resident together (status 0x%x: other "
                               "work shares the device); %s.  Later calls use the form without a device-wide wait"
                               % (what, code & 0xffffffff, note or self.DEFAULT_NOTE))
        return True

    def poll(self, block=False):
        while self.pending and self._retire(block):
            pass
