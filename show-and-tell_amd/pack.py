"""Packed-sequence bookkeeping (host integers only): `pack_padded_sequence`'s batch_sizes for lengths sorted
descending (`/root/reference/models.py:51`, `/root/reference/train.py:135`; the sort is `collate_fn`'s
invariant, `/root/reference/data_loader.py:50`)."""
import ctypes as C

import torch


class PackInfo:
    """batch_sizes (host int32 array for the C ABI), prefix sums (device int32), N rows."""
    _cache = {}

    def __init__(self, lengths, device):
        lengths = [int(l) for l in lengths]
        if not lengths or lengths[-1] < 1:
            raise ValueError("lengths must be >= 1 (pack_padded_sequence)")
        if any(lengths[i] < lengths[i + 1] for i in range(len(lengths) - 1)):
            raise ValueError("lengths must be sorted in decreasing order (pack_padded_sequence, enforce_sorted)")
        self.lengths = lengths
        self.B = len(lengths)
        self.T = lengths[0]
        self.batch_sizes = [sum(1 for l in lengths if l > t) for t in range(self.T)]
        self.prefix = [0]
        for n in self.batch_sizes:
            self.prefix.append(self.prefix[-1] + n)
        self.N = self.prefix[-1]
        self.bs_c = (C.c_int32 * self.T)(*self.batch_sizes)
        self.prefix_dev = torch.tensor(self.prefix, dtype=torch.int32, device=device)

    def prev_rows(self):
        """int64 [N] on the device: for packed row (t, b) the row of the table [h_0 (B rows) ; packed h (N rows)] that holds
        h_{t-1} of sequence b (b for t = 0, B + prefix[t-1] + b after) -- one gather builds `h_prev` for every packed row."""
        idx = getattr(self, "_prev_rows", None)
        if idx is None:
            rows = list(range(self.batch_sizes[0]))
            for t in range(1, self.T):
                rows += [self.B + self.prefix[t - 1] + b for b in range(self.batch_sizes[t])]
            idx = self._prev_rows = torch.tensor(rows, dtype=torch.int64, device=self.prefix_dev.device)
        return idx

    @classmethod
    def get(cls, lengths, device):
        key = (tuple(int(l) for l in lengths), str(device))
        pi = cls._cache.get(key)
        if pi is None:
            if len(cls._cache) > 256:
                cls._cache.clear()
            pi = cls._cache[key] = cls(lengths, device)
        return pi


def pack_targets(captions, lengths):
    """train.py:134-135: targets = pack(captions[:,1:], lengths-1).data, as one gather on the device.
    Returns (targets int64 [N], lengths-1)."""
    l1 = [int(l) - 1 for l in lengths]
    pi = PackInfo.get(l1, captions.device)
    rows = torch.cat([captions[:n, t + 1] for t, n in enumerate(pi.batch_sizes)])
    return rows.contiguous(), l1
