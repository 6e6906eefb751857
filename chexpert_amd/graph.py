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
        # (Dropout / DropConnect masks: their step counter lives in device memory and is bumped inside the captured step --
        # efficientnet.py, cx_dropout_mask_dev -- so every replay draws new masks)
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


_TEST_FAIL_CAPTURE = 0          # tests/test_dp_gpu.py: N > 0 raises at the N-th cut, i.e. in the middle of a captured backward pass


class SegmentedTrainStep:
    """The data-parallel training step as a CHAIN of hipGraphs.

    Collectives are enqueued from Python (parallel.GradReducer), so the whole step cannot be one graph.  Enqueued launch by
    launch it pays a dependent-launch gap at every kernel boundary that a graph does not have: 37.7 ms instead of 30.3 ms per
    DenseNet121 step on one MI355X, 66.5 instead of 56.6 for ResNet152 (bench.py's eager replica against its graph replay).
    Here the step is captured once with the reducer in capture mode: wherever backward completes a gradient bucket
    (`GradReducer.ready`) the capture is cut, so a replay is

        segment 0 (zero_grad, forward, loss, backward up to bucket 0) -> all-reduce(bucket 0) on the reducer's stream
        segment 1 (backward up to bucket 1)                            -> all-reduce(bucket 1) ...
        ...                                                            -> wait for the collectives
        last segment (whatever backward does after the join, optimiser step, scheduler tick)

    with the same launches, the same order and the same collectives as the eager data-parallel step: the kernels between two
    cuts run as a graph, the all-reduces overlap the following segments as before.  All segments share one memory pool."""

    def __init__(self, model, optimizer, x, target, warmup_steps=0, warmup_iters=2):
        if not x.is_cuda:
            raise RuntimeError("SegmentedTrainStep needs device tensors (no CPU path)")
        eng = model._eng()
        red = getattr(eng, "reducer", None)
        if red is None:
            raise RuntimeError("SegmentedTrainStep is the data-parallel form (engine.enable_data_parallel() first); a single process uses GraphedTrainStep")
        self.model, self.opt, self.red = model, optimizer, red
        self.x, self.target = x.clone(), target.clone()
        model.train()
        saved = [(b, b.detach().clone()) for b in model.buffers()]
        nbt = getattr(model, "_nbt_pending", 0)
        s = torch.cuda.Stream(device=x.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(max(1, warmup_iters)):                  # eager warm-up: real collectives, every rank alike
                model.zero_grad()
                self.loss, self.logits = model.forward_backward(self.x, self.target)
            if optimizer is not None:
                optimizer._bufs(2 if hasattr(optimizer, "betas") or hasattr(optimizer, "alpha") else 1)
                optimizer.hyper(warmup_steps)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.segs = []                                              # [(graph, actions)]; an action: ("launch", lo, hi) | ("finish",)
        self._pool = torch.cuda.graph_pool_handle()
        self._g = None
        red.capture = self
        try:
            with torch.cuda.stream(s):
                try:
                    self._begin()
                    model.zero_grad()
                    self.loss, self.logits = model.forward_backward(self.x, self.target)
                    if optimizer is not None:
                        optimizer.step_dev()
                        optimizer.tick()
                    self._g.capture_end()
                    self.segs.append((self._g, ()))
                    self._g = None
                except BaseException:
                    # A capture must be ended on the stream (and thread) that began it, and BEFORE its graph object dies: ending
                    # it after `with torch.cuda.stream(s)` had restored the caller's stream failed torch's stream check, the
                    # stream stayed in capture mode, and tearing the objects down later aborted the process (round 3 parked them
                    # and left through os._exit).  Here the open segment is closed while `s` is still current; a capture the
                    # runtime has invalidated raises from capture_end too, but hipStreamEndCapture has then already taken the
                    # stream out of capture mode.
                    if self._g is not None:
                        try:
                            self._g.capture_end()
                        except Exception:
                            pass
                    self._g, self.segs = None, []
                    raise
        except BaseException as capture_error:
            # the clean-up below touches the GPU; if the failed capture left the device in a state where that raises too, the caller
            # must still see the ORIGINAL error (bench.py logs it and falls back to the eager step), chained, not replaced
            red.capture = None
            try:
                self._restore(s, saved, model, nbt)
            except Exception as cleanup_error:
                raise capture_error from cleanup_error
            raise
        red.capture = None
        self._restore(s, saved, model, nbt)
        self.replays = 0

    @staticmethod
    def _restore(s, saved, model, nbt):
        """Joins the capture stream and puts the buffers the warm-up steps changed back (they really ran)."""
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        with torch.no_grad():
            for b, v in saved:
                b.copy_(v)
        if hasattr(model, "_nbt_pending"):
            model._nbt_pending = nbt

    def _begin(self):
        self._g = torch.cuda.CUDAGraph()
        # thread_local: the process group's watchdog / proxy threads may query events while this thread captures
        self._g.capture_begin(pool=self._pool, capture_error_mode="thread_local")

    def cut(self, *actions):
        """Called by the reducer in capture mode: end the current segment here; `actions` run at this point of every replay."""
        self._g.capture_end()
        self.segs.append((self._g, actions))
        self._begin()
        if _TEST_FAIL_CAPTURE and len(self.segs) == _TEST_FAIL_CAPTURE:
            raise RuntimeError("forced capture failure (test hook)")

    def replay(self, x=None, target=None):
        if x is not None:
            self.x.copy_(x, non_blocking=True)
        if target is not None:
            self.target.copy_(target, non_blocking=True)
        red = self.red
        red.begin()
        for g, actions in self.segs:
            g.replay()
            for action in actions:
                if action[0] == "launch":
                    red._launch(action[1], action[2])
                else:
                    red.wait()
        self.replays += 1
        eng = self.model._eng()
        if hasattr(eng, "packed_version"):
            eng.packed_version = None
        self.model._nbt_pending += 1
        return self.loss, self.logits
