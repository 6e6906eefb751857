"""Fused optimisers over the engine's flat fp32 parameter / gradient buffers (one kernel per step).

Same update rules and defaults as the torch.optim classes the reference wires up
(/root/reference/chexpert.py:470 Adam(lr); :479 SGD(momentum=0.9, nesterov=True); :499
RMSprop(momentum=0.9, eps=1e-3)), with the schedulers of :480 (MultiStepLR[40000, 60000]) and :500
(ExponentialLR(gamma)) folded in as `scheduler_step()`.
"""
import torch

from . import ops


class _Flat:
    def __init__(self, model, lr):
        self.model = model
        self.lr = float(lr)
        self.base_lr = float(lr)
        self.step_count = 0
        self.sched_steps = 0
        self._state = None

    def _bufs(self, n):
        eng = self.model._eng()
        if eng.flat is None:
            raise RuntimeError("run a forward pass first (parameters are bound to the flat buffer lazily)")
        if self._state is None or self._state[0].numel() != eng.flat.numel() or self._state[0].device != eng.flat.device:
            self._state = [torch.zeros_like(eng.flat) for _ in range(n)]
            pend = getattr(self, "_pending_state", None)
            if pend is not None:
                for dst, src in zip(self._state, pend):
                    dst.copy_(src)
                self._pending_state = None
        return eng.flat, eng.flat_grad, self._state

    def zero_grad(self, set_to_none=True):
        self.model.zero_grad(set_to_none=set_to_none)

    # ---- device-resident hyper-parameters (graph replay: chexpert_amd/graph.py)
    SCHED = (0, 1.0, (0, 0))        # (kind, gamma, milestones): 0 none, 1 ExponentialLR, 2 MultiStepLR

    def hyper(self, warmup_steps=0):
        """float[8] on the device: {lr, steps_done, sched_kind, gamma, lr_warmup_steps, milestone0, milestone1, base_lr}
        (include/chexpert_hip.h, cx_optim_tick).  Created from the host-side state on first use; from then on the device
        copy is the truth for `step_dev()` / `tick()` and `sync_from_device()` reads it back."""
        if getattr(self, "_hyper", None) is None:
            kind, gamma, ms = self.SCHED if not hasattr(self, "_sched") else self._sched
            eng = self.model._eng()
            self._hyper = torch.tensor([self.lr, float(self.step_count), float(kind), float(gamma), float(warmup_steps),
                                        float(ms[0]), float(ms[1]), self.base_lr], dtype=torch.float32, device=eng.flat.device)
        return self._hyper

    def tick(self):
        """steps_done += 1 and the scheduler step of chexpert.py:165, on the device."""
        ops.optim_tick(self.hyper())

    def state_dict(self):
        """What the reference saves as `optim_checkpoint_latest.pt` (chexpert.py:188-189), for the flat-buffer state."""
        self.sync_from_device()
        return {"kind": type(self).__name__, "lr": self.lr, "base_lr": self.base_lr, "step_count": self.step_count,
                "sched_steps": self.sched_steps, "state": None if self._state is None else [t.detach().cpu().clone() for t in self._state]}

    def load_state_dict(self, sd):
        if sd.get("kind") != type(self).__name__:
            raise RuntimeError("optimizer checkpoint was written by %s, this is %s" % (sd.get("kind"), type(self).__name__))
        self.lr, self.base_lr, self.step_count, self.sched_steps = sd["lr"], sd["base_lr"], sd["step_count"], sd["sched_steps"]
        self._pending_state = sd["state"]              # copied into the flat-buffer state once the engine is bound
        self._hyper = None

    def sync_from_device(self):
        if getattr(self, "_hyper", None) is not None:
            h = self._hyper.cpu()
            self.lr, self.step_count = float(h[0]), int(h[1])
            # cx_optim_tick steps the scheduler inside the graph: after minibatch number `step` it has been stepped
            # step - max(warm-up, 1) + 1 times (chexpert.py:157-165), which is what an eager resume continues from
            if int(h[2]) != 0:
                self.sched_steps = max(0, int(h[1]) - max(int(h[4]), 1) + 1)


class FusedAdam(_Flat):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(model, lr)
        self.betas, self.eps, self.wd = betas, eps, weight_decay

    def step(self, grad_scale=1.0):
        p, g, (m, v) = self._bufs(2)
        self.step_count += 1
        ops.adam_step(p, g, m, v, self.lr, self.betas[0], self.betas[1], self.eps, self.wd, self.step_count, grad_scale)

    def step_dev(self, grad_scale=1.0):
        p, g, (m, v) = self._bufs(2)
        ops.adam_step_dev(p, g, m, v, self.hyper(), self.betas[0], self.betas[1], self.eps, self.wd, grad_scale)


class FusedSGDNesterov(_Flat):
    def __init__(self, model, lr, momentum=0.9, weight_decay=0.0, milestones=(40000, 60000), gamma=0.1):
        super().__init__(model, lr)
        self.momentum, self.wd, self.milestones, self.gamma = momentum, weight_decay, tuple(milestones), gamma
        ms = (tuple(milestones) + (1 << 30, 1 << 30))[:2]
        self._sched = (2, gamma, ms)

    def step_dev(self, grad_scale=1.0):
        p, g, (buf,) = self._bufs(1)
        ops.sgd_nesterov_step_dev(p, g, buf, self.hyper(), self.momentum, self.wd, grad_scale)

    def step(self, grad_scale=1.0):
        p, g, (buf,) = self._bufs(1)
        ops.sgd_nesterov_step(p, g, buf, self.lr, self.momentum, self.wd, self.step_count == 0, grad_scale)
        self.step_count += 1

    def scheduler_step(self):
        self.sched_steps += 1
        self.lr = self.base_lr * self.gamma ** sum(self.sched_steps >= m for m in self.milestones)


class FusedRMSprop(_Flat):
    def __init__(self, model, lr, alpha=0.99, eps=1e-3, momentum=0.9, weight_decay=0.0, decay=0.97):
        super().__init__(model, lr)
        self.alpha, self.eps, self.momentum, self.wd, self.decay = alpha, eps, momentum, weight_decay, decay
        self._sched = (1, decay, (0, 0))

    def step_dev(self, grad_scale=1.0):
        p, g, (sq, buf) = self._bufs(2)
        ops.rmsprop_step_dev(p, g, sq, buf, self.hyper(), self.alpha, self.eps, self.momentum, self.wd, grad_scale)

    def step(self, grad_scale=1.0):
        p, g, (sq, buf) = self._bufs(2)
        ops.rmsprop_step(p, g, sq, buf, self.lr, self.alpha, self.eps, self.momentum, self.wd, grad_scale)
        self.step_count += 1

    def scheduler_step(self):
        self.sched_steps += 1
        self.lr = self.base_lr * self.decay ** self.sched_steps
