"""`clip_gradient(optimizer, grad_clip)` + `optim.Adam(...).step()` (`/root/reference/train.py:55-56, 88-91, 145-146`) as ONE
launch for the drop-in training loop (the loop that keeps `model(...)`, `criterion`, `loss.backward()` as they are).

`FusedClampAdam(params, lr, clip=0.1)` is a `torch.optim.Optimizer`: `param_groups[0]["lr"]` is what `set_lr` (train.py:93-95)
writes, `step()` / `zero_grad()` / `state_dict()` / `load_state_dict()` keep their meaning and the state dict has
`torch.optim.Adam`'s layout.  The parameters are re-homed into one flat f32 buffer (so build it AFTER `model.cuda()`), with
flat gradient / exp_avg / exp_avg_sq twins; `step()` is `sat_clamp_adam_step` over the whole buffer -- torch's single-tensor
Adam arithmetic in its order, elementwise clamp first when `clip` is set (then drop the separate `clip_gradient` call).
A torch Adam over the 12 Show-Attend-Tell parameters is ~10 multi-tensor launches + 19 clamp launches per step."""
import torch

from . import _lib as L


def _pad4(n):
    return (n + 3) // 4 * 4


class FusedClampAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, clip=None):
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("no trainable parameters")
        if any(p.dtype != torch.float32 for p in params):
            raise TypeError("FusedClampAdam holds f32 parameters")
        L.require_gpu(params[0], "parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps))
        self.clip = clip
        self.lib = L.load()
        dev = params[0].device
        self._slices, off = [], 0
        for p in params:
            self._slices.append((off, p.numel()))
            off += _pad4(p.numel())
        self.n = off
        self.flat = torch.zeros(off, device=dev)
        self.flat_grad = torch.zeros(off, device=dev)
        self.exp_avg = torch.zeros(off, device=dev)
        self.exp_avg_sq = torch.zeros(off, device=dev)
        self.step_count = 0
        self._params = params
        self._views = []
        for p, (o, n) in zip(params, self._slices):
            self.flat[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + n].view(p.shape)
            self._views.append(self.flat_grad[o:o + n].view(p.shape))
            p.grad = self._views[-1]

    def zero_grad(self, set_to_none=False):
        """One memset; the `.grad` of every parameter stays the view into the flat gradient buffer, so the backward
        accumulates in place.  (`model.zero_grad()` -- train.py:137 -- drops the views; `step()` then copies the gradients in.)"""
        self.flat_grad.zero_()
        for p, v in zip(self._params, self._views):
            p.grad = v

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        src, dst, missing = [], [], []
        for i, (p, v) in enumerate(zip(self._params, self._views)):
            if p.grad is None:
                missing.append(i)
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
        if src:
            torch._foreach_copy_(dst, src)
        g = self.param_groups[0]
        self.step_count += 1
        clip = float(self.clip) if self.clip else 0.0
        # torch skips a parameter whose grad is None: run the segments between such parameters
        segs, start = [], 0
        for i in missing:
            o, n = self._slices[i]
            if o > start:
                segs.append((start, o))
            start = o + _pad4(n)
        if start < self.n:
            segs.append((start, self.n))
        for a, b in segs:
            L.check(self.lib.sat_clamp_adam_step(self.flat.data_ptr() + a * 4, self.flat_grad.data_ptr() + a * 4,
                                                 self.exp_avg.data_ptr() + a * 4, self.exp_avg_sq.data_ptr() + a * 4, b - a,
                                                 float(g["lr"]), g["betas"][0], g["betas"][1], float(g["eps"]), clip,
                                                 self.step_count, L.stream()), "sat_clamp_adam_step")
        # the kernel wrote the parameters through raw pointers: move their version counters as an in-place torch update
        # would, so that caches keyed on `_version` (kernel-layout weight copies of a conv stack) see the step
        torch.autograd.graph.increment_version(self._params)
        return loss

    def state_dict(self):
        state = {}
        if self.step_count > 0:
            for i, (p, (o, n)) in enumerate(zip(self._params, self._slices)):
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.exp_avg[o:o + n].view(p.shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[o:o + n].view(p.shape).clone()}
        g = self.param_groups[0]
        group = {"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": 0, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": False,
                 "params": list(range(len(self._params)))}
        return {"state": state, "param_groups": [group], "grad_clip": self.clip}

    def load_state_dict(self, sd):
        group = sd["param_groups"][0]
        if len(group["params"]) != len(self._params):
            raise ValueError("optimizer state has %d parameters, this optimizer %d" % (len(group["params"]), len(self._params)))
        g = self.param_groups[0]
        g["lr"], g["betas"], g["eps"] = float(group["lr"]), tuple(group["betas"]), float(group["eps"])
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        steps = set()
        for i, (p, (o, n)) in enumerate(zip(self._params, self._slices)):
            st = sd["state"].get(group["params"][i])
            if st is None:
                continue
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError("optimizer state %d has shape %s, expected %s" % (i, tuple(st["exp_avg"].shape), tuple(p.shape)))
            self.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(st["step"]))
        if len(steps) > 1:
            raise ValueError("per-parameter Adam step counts differ (%s): one flat step count is kept" % sorted(steps))
        self.step_count = steps.pop() if steps else 0
