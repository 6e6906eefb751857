#!/usr/bin/env python
"""Headline benchmark: DenseNet121 forward+backward on synthetic 320x320 X-rays, bf16 storage.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
           bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one minibatch resident in HBM: forward (batch-statistic
BatchNorm), BCE loss, full backward into the flat fp32 gradient buffer, (N>1) the RCCL gradient
all-reduce, and the fused Adam update of the fp32 masters (also timed on its own and reported in `config`).  Prints ONE JSON line
(rank 0) with `roofline` (dominant kernel, HIP-event timed inside the timed region) and `cpu_baseline`
(the oracle's fp32 CPU restatement of the same step on a bounded sample).
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

# torch is imported by main() AFTER the launcher decision: a parent that only spawns the ranks (`python bench.py --gpus N` without
# WORLD_SIZE) never imports it, so it cannot have touched HIP
torch = None

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
ALG_BYTES_PER_IMG = 283.1e6    # densenet121@320 bf16, fwd+bwd
ALG_BYTES = {"densenet121": 283.1e6, "aadensenet121": 295.4e6, "resnet152": 555.0e6, "aaresnet152": 555.0e6, "efficientnet-b4": 594.5e6,
             "efficientnet-b0": 166.2e6}
ALG_BYTES_FP32 = {k_: 2.0 * v_ for k_, v_ in ALG_BYTES.items()}   # fp32 storage: twice the bf16 bytes (SURVEY.md section 8d, fp32 column)


class KernelTimer:
    """HIP events around the launches of the conv kernels (on torch's current stream, which is the
    stream the library launches on).  Tags follow the kernel template instantiation."""

    def __init__(self, ops):
        self.ops = ops
        self.orig = (ops.conv_gemm, ops.conv_wgrad)
        self.records = {}
        self.only = None
        self.enabled = False

    # Tags are the names of the kernel instantiations the library dispatched to, as the library itself reports them
    # (cx_last_kernel(), include/chexpert_hip.h): what rocprofv3 --kernel-trace lists, with no second copy of the dispatch rules.
    @staticmethod
    def _dims(t):
        B, H, W, C = t.shape
        return B * H * W, C

    def install(self):
        ops = self.ops
        og, ow = self.orig

        def conv_gemm(x, w, y, **kw):
            if not self.enabled:
                return og(x, w, y, **kw)
            mi, ci = self._dims(x)
            mo, co = self._dims(y)
            alg = (4.0 if x.dtype == torch.float32 else 2.0) * (mi * ci + mo * co)      # |X| + |Y| elements (SURVEY 8d rule, per kernel)
            if kw.get("fused_dw") is not None:
                alg += 2.0 * mo * co                 # the weight-gradient half also needs the layer input (|dZ| counted once)
            return self._timed(alg, og, x, w, y, **kw)

        def conv_wgrad(g, x, dw, **kw):
            if not self.enabled:
                return ow(g, x, dw, **kw)
            mg, cg = self._dims(g)
            mx, cx = self._dims(x)
            return self._timed((4.0 if g.dtype == torch.float32 else 2.0) * (mg * cg + mx * cx), ow, g, x, dw, **kw)
        ops.conv_gemm, ops.conv_wgrad = conv_gemm, conv_wgrad

    def _timed(self, alg, fn, *a, **kw):
        """Events around the kernel of one call, filed under the kernel name the library reports for it (`only`: other kernels'
        records are dropped).  The engines defer the slab sums of their weight gradients to one table-driven launch per pass
        (ops.wgrad_defer_begin), so an entry point is ONE kernel launch between the two events -- what rocprofv3 reports for it."""
        e0, e1 = self._event(), self._event()
        e0.record()
        r = fn(*a, **kw)
        e1.record()
        tag = self._raw().cx_last_kernel().decode()
        if self.only is None or tag == self.only:
            self.records.setdefault(tag, []).append((e0, e1, alg))
        return r

    def _event(self):
        """An event whose hipEvent already exists (torch creates it at the first record): created in batches outside the brackets."""
        pool = getattr(self, "_pool", None)
        if not pool:
            pool = self._pool = [torch.cuda.Event(enable_timing=True) for _ in range(256)]
            for e in pool:
                e.record()
        return pool.pop()

    def _raw(self):
        if getattr(self, "_rawlib", None) is None:
            from chexpert_amd import _lib
            self._rawlib = ctypes.CDLL(_lib.LIB_PATH)
            self._rawlib.cx_last_kernel.restype = ctypes.c_char_p
        return self._rawlib

    def summary(self):
        out = {}
        for tag, recs in self.records.items():
            ms = sum(e0.elapsed_time(e1) for e0, e1, _ in recs)
            out[tag] = dict(launches=len(recs), ms=ms, alg_bytes=sum(a for _, _, a in recs))
        return out


def _committed(suffix, args):
    """Latest profiles/*<suffix>*.json taken on this workload (key model:dtype:batch:size), else None."""
    d = os.path.join(ROOT, "profiles")
    key = "%s:%s:%d:%d" % (args.model, args.dtype, args.batch, args.size)
    best = None
    for f in sorted(os.listdir(d)) if os.path.isdir(d) else []:
        if suffix in f and f.endswith(".json"):
            j = json.load(open(os.path.join(d, f)))
            if j.get("workload_key", "densenet121:bf16:256:320") == key:
                best = dict(j, _file="profiles/" + f)
    return best


def _lookup(kernels, tag):
    """Entry of a committed per-kernel table for a bench tag: the exact instantiation name, else the only instantiation of the
    same kernel template (tags that do not spell the template arguments, e.g. conv3x3_strip_wgrad_kernel)."""
    if tag in kernels:
        return kernels[tag]
    same = [k for k in kernels if k.split("<")[0] == tag.split("<")[0]]
    return kernels[same[0]] if len(same) == 1 and "<" not in tag else None


def pmc_traffic(kernel, args):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/*_pmc_traffic*.json: FETCH_SIZE x2 + WRITE_SIZE,
    collected with rocprofv3 --pmc in their own runs); null when no pass was taken on this workload."""
    j = _committed("_pmc_traffic", args)
    k = None if j is None else _lookup(j["kernels"], kernel)
    return (None, None) if k is None else (k["hbm_bytes_per_launch"], j["_file"])


COPY_GUIDE_GBS = 6290.0        # MI355X_MICROARCH.md, chip-level parameters: float4 copy, 79 % of the 8 TB/s specification


def copy_bandwidth(dev, ops):
    """Device-to-device copy of 1 GiB (read + write = 2 GiB of HBM traffic) through the library's own 16-byte-per-lane grid-stride
    copy kernel (cx_copy_stream): the measured stream rate beside the 8 TB/s specification and the guide's 6.29 TB/s."""
    a = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    ops.copy_stream(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.copy_stream(a, b)
    e1.record()
    torch.cuda.synchronize()
    return 5 * 2 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9


def host_cores():
    """Cores this process may actually use (cgroup / affinity aware; the GPU box gives 16 per GPU)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, p_ = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p_))))
    except Exception:
        pass
    return max(1, min(n, 32))


def log(msg):
    print("[bench %.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def cpu_baseline(n_classes, steps=3, batch=4, size=320):
    """The reference CPU path as restated by oracle/ (fp32, all host cores): the same step as the GPU line -- forward, loss,
    backward and the Adam update (chexpert.py:159-164, :470)."""
    from chexpert_amd import synth
    from oracle import nets, step
    cores = host_cores()
    torch.set_num_threads(cores)
    spec = nets.densenet_spec(n_classes)
    sd = synth.fill_state_dict_(nets.zeros_state_dict(spec), 5)
    x, t = synth.xray_batch(11, batch, size), synth.targets(12, batch, n_classes)
    fwd = lambda s, xx: nets.densenet_forward(s, xx, train=True)
    for k in step.trainable(sd):
        sd[k].requires_grad_(True)
    opt, _ = step.make_optimizer("adam", [sd[k] for k in step.trainable(sd)], 1e-4)
    step.train_step(fwd, sd, x, t, opt)     # warm-up
    t0 = time.perf_counter()
    done = 0
    while done < steps or (time.perf_counter() - t0 < 12.0 and done < 400):      # a bounded sample: >= 3 steps and ~12 s of CPU work
        step.train_step(fwd, sd, x, t, opt)
        done += 1
    steps = done
    dt = time.perf_counter() - t0
    return {"value": round(batch * steps / dt, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "densenet121 fp32 CPU (oracle restatement of chexpert.py:159-164 fwd+loss+bwd+Adam), bs=%d, %d steps, %dx%d"
                      % (batch, steps, size, size)}


def committed_counter(kernel, args, key):
    """Per-kernel figures from the committed SQ counter pass (profiles/*_sq_counters*.json: SQ_VALU_MFMA_BUSY_CYCLES over
    GRBM_GUI_ACTIVE, collected with rocprofv3 --pmc in its own run); null when no pass was taken on this workload."""
    j = _committed("_sq_counters", args)
    k = None if j is None else _lookup(j["kernels"], kernel)
    return (None, None) if k is None else (k.get(key), j["_file"])


def build_model(name, classes, size, dtype, dev):
    """The drop-in constructors of chexpert.py:461-500 (random initialisation: there are no checkpoints offline)."""
    attn = {"k": 0.2, "v": 0.1, "nh": 8, "relative": True, "input_dims": (size, size)}      # chexpert.py:476
    if name == "densenet121":
        from chexpert_amd.models import densenet121
        model = densenet121(num_classes=classes)
    elif name == "aadensenet121":
        from chexpert_amd.models import DenseNet
        model = DenseNet(32, (6, 12, 24, 16), 64, num_classes=classes, attn_params=attn)
    elif name == "resnet152":
        from chexpert_amd.models import resnet152
        model = resnet152(num_classes=classes)
    elif name == "aaresnet152":
        from chexpert_amd.models import Bottleneck, ResNet
        model = ResNet(Bottleneck, [3, 8, 36, 3], num_classes=classes, attn_params=attn)
    else:
        from chexpert_amd.models import construct_model
        model = construct_model(name, classes)
    return model.storage_dtype(dtype).to(dev).train()


# BASELINE.json configs[2..4] at their per-GPU batch, with the optimiser chexpert.py wires to each (:479 SGD + Nesterov for the
# attention-augmented DenseNet, :485 Adam for resnet152, :499 RMSprop for EfficientNet)
OTHER_CONFIGS = [("aadensenet121", 128, 320, "sgd_nesterov"), ("resnet152", 128, 320, "adam"), ("efficientnet-b4", 64, 380, "rmsprop")]


def make_optimizer(kind, model):
    from chexpert_amd.optim import FusedAdam, FusedRMSprop, FusedSGDNesterov
    if kind == "adam":
        return FusedAdam(model, lr=1e-4)
    if kind == "sgd_nesterov":
        return FusedSGDNesterov(model, lr=1e-4)
    return FusedRMSprop(model, lr=1e-4)


def run_other_config(name, batch, size, opt_kind, classes, dev, steps=20, warmup=3):
    """One of the other BASELINE configurations on this GPU: `steps` replays of the captured training step (forward, loss, backward,
    optimiser + scheduler tick), inputs resident in HBM; the same timing brackets as the headline."""
    from chexpert_amd import synth
    from chexpert_amd.graph import GraphedTrainStep
    t_in = time.perf_counter()
    model = build_model(name, classes, size, "bf16", dev)
    # BASELINE configs[4] is "efficientnet-b4 + data aug": its input is the decoded grey image as uint8 (1 byte per pixel, what the
    # loader hands over), jittered on the GPU every step (brightness / contrast +-0.25: the reference's `_data_aug` rows, notebook
    # cell 6) and whitened + expanded in the first kernel (cx_u8_to_nhwc8); the other configurations take the whitened fp32 batch
    aug = name.startswith("efficientnet")
    if aug:
        from chexpert_amd import ops as _ops
        x_u8 = synth.xray_u8(1000, batch, size).view(batch, 1, size, size).contiguous().to(dev)
        u = synth.uniform(4242, (steps + warmup + 4, 3, batch), 0.0, 1.0).to(dev)
        jit = [((0.75 + 0.5 * u[i, 0]).contiguous(), (0.75 + 0.5 * u[i, 1]).contiguous(), (u[i, 2] > 0.5).to(torch.int32).contiguous())
               for i in range(u.shape[0])]
        x = _ops.u8_jitter(x_u8, *jit[0])
    else:
        x = synth.xray_batch(1000, batch, size).to(dev)
    t = synth.targets(2000, batch, classes).to(dev)
    opt = make_optimizer(opt_kind, model)
    out = {"batch": batch, "size": size, "optimizer": opt_kind, "steps": steps}
    if aug:
        out["input"] = "uint8 grey bytes resident in HBM; cx_u8_jitter (brightness / contrast +-0.25, new factors every step) inside the timed region"
    gstep = None
    try:
        gstep = GraphedTrainStep(model, opt, x, t)
        launch = "hipGraph replay"
        run = gstep.replay
    except Exception as e:
        log("%s: graph capture failed (%s: %s); timing the eager step" % (name, type(e).__name__, e))
        out["capture_failed"] = 1
        launch = "eager enqueue"

        def run():
            model.zero_grad()
            r = model.forward_backward(x, t)
            opt.step()
            return r
    if aug:                                  # every step: jitter the bytes with that step's factors, hand them to the step
        plain, it = run, iter(range(1, len(jit)))
        xj = torch.empty_like(x_u8)
        if gstep is not None:
            def run():
                _ops.u8_jitter(x_u8, *jit[next(it)], out=xj)
                return gstep.replay(xj)
        else:
            def run():
                _ops.u8_jitter(x_u8, *jit[next(it)], out=x)
                return plain()
    for _ in range(warmup):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _ = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    img_s = batch * steps / dt
    out.update(img_s=round(img_s, 1), ms_per_step=round(dt / steps * 1e3, 3), launch=launch, loss=round(float(loss.item()), 5),
               model_hbm_roofline_frac=round(img_s * ALG_BYTES[name] / (HBM_PEAK_GBS * 1e9), 4))
    log("%s bs=%d: %.0f img/s, %.2f ms/step (%s; %.1f s in all)" % (name, batch, img_s, dt / steps * 1e3, launch, time.perf_counter() - t_in))
    del gstep, run, model, opt, x, t
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return out


def launcher_command(n, argv, port=None):
    """`python bench.py --gpus N` without WORLD_SIZE: the command that starts the N ranks (one process per GPU over RCCL), exactly
    what the driver runs for N > 1."""
    if port is None:
        import socket
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def spawn_ranks(n, argv):
    """Runs the ranks as a CHILD process and returns its exit code; rank 0's JSON line reaches stdout through the inherited pipe.
    The parent has not imported torch, let alone initialised HIP (never exec from a process that has: it takes the box down)."""
    assert "torch" not in sys.modules or torch is None, "the launcher parent must not have imported torch"
    cmd = launcher_command(n, argv)
    print("[bench] spawning %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU minibatch (BASELINE config: 256)")
    ap.add_argument("--size", type=int, default=320)
    ap.add_argument("--classes", type=int, default=14)
    ap.add_argument("--model", default="densenet121", choices=["densenet121", "aadensenet121", "resnet152", "aaresnet152", "efficientnet-b4", "efficientnet-b0"],
                    help="densenet121 is the headline (BASELINE configs[1]); the others are reported for reference only")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"],
                    help="activation storage type: bf16 (the headline, BASELINE configs[1]) or the fp32 parity mode (densenet121 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="enqueue every launch from Python instead of replaying the captured hipGraph")
    ap.add_argument("--roofline-kernel", default=None, help="kernel tag to time (default: the one with the largest share)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the BASELINE configs[2..4] that the headline run reports under config.other_configs (N = 1 only)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    global torch
    import torch
    import torch.distributed as dist
    from chexpert_amd import ops, synth
    from chexpert_amd.optim import FusedAdam

    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.gpus != world:
        sys.exit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d (or without WORLD_SIZE)" % (
            args.gpus, world, args.gpus))
    # CHEXPERT_BENCH_BACKEND=gloo rehearses the N>1 path on a box with fewer GPUs than ranks (ranks share a device); the
    # driver's runs use RCCL ("nccl"), one rank per GPU
    backend = os.environ.get("CHEXPERT_BENCH_BACKEND", "nccl")
    local_dev = local if backend == "nccl" else local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    # CHEXPERT_BENCH_FORCE_DP=1 (under torch.distributed.run --nproc-per-node 1): the data-parallel code path -- reducer, RCCL
    # all-reduce launches between the graph segments -- on a communicator of ONE rank: what a one-GPU box can rehearse of it
    dp = world > 1 or os.environ.get("CHEXPERT_BENCH_FORCE_DP") == "1"
    if dp:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    torch.manual_seed(1234)
    # (the fp32 storage mode -- north_star's 1e-3 parity mode -- covers every model family)
    model = build_model(args.model, args.classes, args.size, args.dtype, dev)
    model.train()
    x = synth.xray_batch(1000 + rank, args.batch, args.size).to(dev)
    t = synth.targets(2000 + rank, args.batch, args.classes).to(dev)
    opt = FusedAdam(model, lr=1e-4)             # the reference's optimiser (chexpert.py:473): Adam, lr 1e-4, over the flat fp32 masters

    timer = KernelTimer(ops)
    timer.install()

    def step():
        return model.forward_backward(x, t)

    log("model + data ready")
    if dp:                                      # replicas start identical and average their gradients from the first step on
        from chexpert_amd.parallel import broadcast_module_state
        model._eng().bind(dev)
        broadcast_module_state(model)
        model._eng().enable_data_parallel()
    loss, _ = step()                           # binds the engine (N = 1), allocates workspaces
    torch.cuda.synchronize()
    log("first step done, loss %.4f" % loss.item())
    try:
        opt.step()                              # binds the moment buffers
    except (AttributeError, RuntimeError) as e: # an engine without flat parameter buffers: forward+backward only
        log("no fused optimiser for this model (%s)" % e)
        opt = None
    for _ in range(max(0, args.warmup - 1)):
        model.zero_grad()
        step()
    torch.cuda.synchronize()

    # one instrumented step over every conv kernel family to find the dominant one
    only = args.roofline_kernel
    if only is None:
        timer.enabled, timer.only = True, None
        model.zero_grad()
        step()
        torch.cuda.synchronize()
        fam = timer.summary()
        only = max(fam, key=lambda k: fam[k]["ms"])
        for k in sorted(fam, key=lambda k: -fam[k]["ms"]):
            log("  %-34s %4d launches %7.2f ms  %6.0f GB/s" % (k, fam[k]["launches"], fam[k]["ms"],
                                                             fam[k]["alg_bytes"] / fam[k]["ms"] / 1e6))
        timer.records = {}
    timer.enabled, timer.only = False, only
    log("warm-up done; roofline kernel = %s" % only)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def eager_steps(n):
        l = None
        for _ in range(n):                      # a full training step: zero, forward, loss, backward (+ all-reduce), Adam update
            model.zero_grad()
            l, _ = step()
            if opt is not None:
                opt.step()
        return l

    # N = 1: the step is captured once as a hipGraph (chexpert_amd/graph.py) and the timed region replays it -- the ~900
    # launches of a step cost ~18 ms of host time when enqueued one by one.  N > 1 enqueues eagerly (RCCL collectives are
    # issued from Python between the kernels).  HIP events cannot be recorded inside a captured graph, so the dominant
    # kernel's launch time is taken with events in an eager replica of the same K steps right after the timed region.
    use_graph = opt is not None and not args.no_graph and os.environ.get("CHEXPERT_BENCH_GRAPH", "1") != "0"
    gstep, capture_failed = None, 0
    if use_graph:
        # N > 1: the same step as a chain of graph segments cut where a gradient bucket is complete; the all-reduces between
        # them are enqueued from Python exactly as in the eager step (graph.SegmentedTrainStep)
        from chexpert_amd.graph import GraphedTrainStep, SegmentedTrainStep
        timer.enabled = False
        try:
            gstep = GraphedTrainStep(model, opt, x, t) if not dp else SegmentedTrainStep(model, opt, x, t)
        except Exception as e:                   # capture is an optimisation of the host side only: fall back loudly
            log("graph capture failed (%s: %s); timing the eager step" % (type(e).__name__, e))
            gstep, capture_failed = None, 1
        if dp:
            # every rank takes the same branch: the warm-up replays and the probe below contain collectives, so a capture that
            # failed on ANY rank sends all of them to the eager step (a rank that skipped them would leave the others waiting in
            # RCCL).  The constructor's own warm-up steps ran their collectives on every rank before the capture began.
            flag = torch.tensor([capture_failed], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            capture_failed = int(flag.item())
            if capture_failed:
                gstep = None
        if gstep is not None:
            for _ in range(max(1, args.warmup)):
                gstep.replay()
            torch.cuda.synchronize()
            log("hipGraph captured and warmed up" + ("" if not dp else " (%d segments)" % len(gstep.segs)))
            if dp:
                # Both forms issue the same kernels and collectives; which is faster depends on how the backend's collectives
                # share the queues with the replayed segments -- measured here on a few steps, decided for all ranks alike
                def timed(fn, n):
                    barrier()
                    a = time.perf_counter()
                    fn(n)
                    torch.cuda.synchronize()
                    tt_ = torch.tensor([time.perf_counter() - a], device=dev, dtype=torch.float64)
                    dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
                    return tt_.item() / n
                n_probe = max(2, min(5, args.warmup))
                t_seg = timed(lambda n: [gstep.replay() for _ in range(n)], n_probe)
                opt.sync_from_device()
                t_eag = timed(eager_steps, n_probe)
                log("probe: %.1f ms/step as graph segments, %.1f ms/step enqueued eagerly" % (t_seg * 1e3, t_eag * 1e3))
                if t_eag < t_seg:
                    gstep = None
    barrier()
    t0 = time.perf_counter()
    if gstep is not None:
        for _ in range(args.steps):
            loss, _ = gstep.replay()
    else:
        # N = 1 eager (--no-graph): the dominant kernel's events are taken inside the timed region.  N > 1: nothing but the step
        # runs between the barriers; the events come from an eager replica afterwards, as for the graph
        timer.enabled, timer.only = not dp, only
        loss = eager_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if dp:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    log("timed region done: %.1f ms/step (%s)" % (dt / args.steps * 1e3, "graph replay" if gstep is not None else "eager"))
    if gstep is not None or dp:
        if gstep is not None:
            opt.sync_from_device()
        timer.enabled, timer.only, timer.records = True, only, {}
        e0 = time.perf_counter()
        eager_steps(args.steps)
        torch.cuda.synchronize()
        log("eager replica for the kernel events: %.1f ms/step" % ((time.perf_counter() - e0) / args.steps * 1e3))
    timer.enabled = False
    ksum = timer.summary()[only]

    # optimiser step timed on its own as well (it IS inside the timed region above; reported for reference)
    torch.cuda.synchronize()
    o0 = time.perf_counter()
    for _ in range(5):
        if opt is not None:
            opt.step()
    torch.cuda.synchronize()
    opt_ms = (time.perf_counter() - o0) / 5 * 1e3

    copy_gbs = copy_bandwidth(dev, ops) if rank == 0 else 0.0
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = world * args.batch * args.steps / dt
        avg_ms = ksum["ms"] / ksum["launches"]
        achieved = ksum["alg_bytes"] / (ksum["ms"] * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(only, args)
        mfma_util, mfma_src = committed_counter(only, args, "mfma_util")
        out = {
            "metric": "images/sec fwd+bwd %s %dx%d %s (per-GPU rate in config.images_per_sec_per_gpu)" % (
            "DenseNet121" if args.model == "densenet121" else args.model, args.size, args.size, args.dtype),
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.dtype == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": "%s %s %dxMI355X %dx%d bs=%d per GPU, U-Ones labels (%s); %d classes; "
                                   "random-init weights, synthetic uint8 X-rays" % (
                                       args.model, args.dtype, world, args.size, args.size, args.batch,
                                       "BASELINE configs[1]" if (args.model, args.dtype) == ("densenet121", "bf16") else
                                       "not the headline config", args.classes),
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                       "rccl_ranks": (dist.get_world_size() if dp else 1), "collective_backend": (dist.get_backend() if dp else None),
                       "images_per_sec_per_gpu": round(value / world, 2),
                       "model_hbm_roofline_frac": round(value / world * (ALG_BYTES_FP32 if args.dtype == "fp32" else ALG_BYTES)[
                           args.model] / (HBM_PEAK_GBS * 1e9), 4),
                       "optimizer_step_ms": round(opt_ms, 3), "loss": round(float(loss.item()), 5),
                       "measured_copy_GBs": round(copy_gbs, 1), "guide_copy_GBs": COPY_GUIDE_GBS,
                       "copy_kernel": "cx_copy_stream (16 B per lane, contiguous 16 KB pieces, non-temporal; 1 GiB read + 1 GiB written)",
                       "capture_failed": capture_failed,
                       "launch": ("hipGraph replay" if not dp else "hipGraph segments between the all-reduces") if gstep is not None else "eager enqueue"},
            "roofline": {"bound": "hbm", "kernel": only, "launches_per_step": ksum["launches"] // args.steps,
                         "avg_launch_ms": round(avg_ms, 4), "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_src and ("committed rocprofv3 --pmc pass %s (not measured in this run)" % traffic_src),
                         "alg_bytes_per_launch": round(ksum["alg_bytes"] / ksum["launches"]),
                         "mfma_util": mfma_util, "mfma_util_source": mfma_src and ("committed rocprofv3 --pmc pass %s (not measured in this run)" % mfma_src),
                         "timing": "hip events around each kernel launch, eager replica of the timed steps" if (gstep is not None or dp)
                         else "hip events around each kernel launch inside the timed region"},
        }
        headline = (args.model, args.dtype, world) == ("densenet121", "bf16", 1) and not dp
        if headline and not args.no_other_configs:
            # BASELINE.json configs[2..4] at N = 1, after the headline's timed region (they share nothing with it): the headline
            # model's workspaces go first
            del gstep, model, opt
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            out["config"]["other_configs"] = {}
            for name, batch, size, opt_kind in OTHER_CONFIGS:
                out["config"]["other_configs"]["%s_bs%d_%d" % (name, batch, size)] = run_other_config(name, batch, size, opt_kind, args.classes, dev)
        if not args.no_cpu_baseline and headline:
            log("cpu baseline on %d cores ..." % host_cores())
            out["cpu_baseline"] = cpu_baseline(args.classes)
        print(json.dumps(out))
    if dp:
        dist.barrier()                 # rank 0 is still measuring the copy bandwidth / printing: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
