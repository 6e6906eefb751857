"""Host-side enqueue cost of one eager training step (zero_grad + forward + loss + backward + fused optimiser) per model: the step at
a batch so small that the GPU finishes long before the host does is pure Python / ctypes / launch time.  This is what every rank of a
data-parallel run pays per step (collectives are issued from Python between the kernels, so N > 1 cannot replay one graph):
    python scratch/host_rate.py            ->  one line per model (profiles/r03_host_enqueue.txt)"""
import sys, time, torch
sys.path.insert(0, '.')
from chexpert_amd import synth
from chexpert_amd.models import DenseNet, Bottleneck, ResNet, construct_model, densenet121, resnet152
from chexpert_amd.optim import FusedAdam
dev = torch.device('cuda:0')
ATT = lambda s: {"k": 0.2, "v": 0.1, "nh": 8, "relative": True, "input_dims": (s, s)}
MODELS = {
    "densenet121": (lambda: densenet121(num_classes=14), 320, 256, 29.2),
    "aadensenet121": (lambda: DenseNet(32, (6, 12, 24, 16), 64, num_classes=14, attn_params=ATT(320)), 320, 128, 29.8),
    "resnet152": (lambda: resnet152(num_classes=14), 320, 128, 55.3),
    "efficientnet-b4": (lambda: construct_model("efficientnet-b4", 14), 380, 64, 36.1),
}
for name, (ctor, S, B, gpu_ms) in MODELS.items():
    m = ctor().to(dev).train()
    x, t = synth.xray_batch(1, 2, S).to(dev), synth.targets(2, 2, 14).to(dev)
    opt = None
    for _ in range(3):
        m.zero_grad(); m.forward_backward(x, t)
        if opt is None:
            try:
                opt = FusedAdam(m, lr=1e-4); opt.step()
            except Exception:
                opt = False
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        m.zero_grad(); m.forward_backward(x, t)
        if opt:
            opt.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    # (the GPU step times are the committed bench lines of profiles/r03_bench_*.json, quoted for comparison)
    print("%-16s host enqueue %.1f ms/step (batch 2: the GPU is idle most of the step); GPU step at batch %d: %.1f ms" % (
        name, (t1 - t0) / n * 1e3, B, gpu_ms), flush=True)
    del m
    torch.cuda.empty_cache()
