"""One training step (chexpert.py:159-165: forward, BCE loss, backward, optimiser step, scheduler step) captured once as a
hipGraph and replayed per minibatch.

Why: a DenseNet121 step is ~900 kernel launches; enqueued one by one through ctypes they cost ~18 ms of host time, which
becomes the floor once the GPU side of the step is faster than that.  A replay costs ~10-20 us of host time (one
`hipGraphLaunch`).  Shapes are static: the caller copies each minibatch into `x` / `target` (device tensors owned by this
object) and calls `replay()`.  Learning rate and step count live in device memory (`optim._Flat.hyper`), so the optimiser and
the scheduler advance inside the graph.  Two-stream sections of backward (weight gradients on the side stream) are captured
as forks/joins of the graph.
"""
import torch


class GraphedTrainStep:
    def __init__(self, model, optimizer, x, target, warmup_steps=0, warmup_iters=2):
        if not x.is_cuda:
            raise RuntimeError("GraphedTrainStep needs device tensors (no CPU path)")
        self.model, self.opt = model, optimizer
        self.x = x.clone()
        self.target = target.clone()
        eng = model._eng()
        if getattr(eng, "reducer", None) is not None:
            raise RuntimeError("graph capture of the data-parallel step is not supported (collectives are enqueued eagerly)")
        if any(getattr(m_, "p", 0) for m_ in model.modules() if type(m_).__name__ == "DropMarker"):
            raise RuntimeError("graph capture would freeze the Dropout / DropConnect seeds (host-drawn per step); use the eager step")
        model.train()
        # construction is free of side effects on the model: the warm-up steps below really run (they update BatchNorm running
        # statistics), so the module buffers and the num_batches_tracked bookkeeping are put back afterwards
        saved = [(b, b.detach().clone()) for b in model.buffers()]
        nbt = getattr(model, "_nbt_pending", 0)
        # eager warm-up on a side stream: binds the engine, allocates the workspaces, sets kernel attributes, creates the
        # optimiser state -- none of which may happen during capture
        s = torch.cuda.Stream(device=x.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(max(1, warmup_iters)):
                model.zero_grad()
                self.loss, self.logits = model.forward_backward(self.x, self.target)
            if optimizer is not None:
                optimizer._bufs(2 if hasattr(optimizer, "betas") or hasattr(optimizer, "alpha") else 1)
                optimizer.hyper(warmup_steps)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            model.zero_grad()
            self.loss, self.logits = model.forward_backward(self.x, self.target)
            if optimizer is not None:
                optimizer.step_dev()
                optimizer.tick()
        with torch.no_grad():
            for b, v in saved:
                b.copy_(v)
        if hasattr(model, "_nbt_pending"):
            model._nbt_pending = nbt
        self.replays = 0

    def replay(self, x=None, target=None):
        """Run one step; returns (loss, logits) device tensors that the NEXT replay overwrites."""
        if x is not None:
            self.x.copy_(x, non_blocking=True)
        if target is not None:
            self.target.copy_(target, non_blocking=True)
        self.graph.replay()
        self.replays += 1
        # the captured optimiser kernel has just changed the fp32 masters without touching tensor versions: an eval forward that
        # follows must repack the weights (Engine.pack(train=False) compares versions)
        eng = self.model._eng()
        if hasattr(eng, "packed_version"):
            eng.packed_version = None
        self.model._nbt_pending += 1           # BatchNorm num_batches_tracked is host-side bookkeeping (flushed by state_dict())
        return self.loss, self.logits
