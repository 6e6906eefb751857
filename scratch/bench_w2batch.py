"""3x3 weight gradients of a dense block: per-layer launches against one batched launch (cx_conv3x3_wgrad_batch), B = 256.
python scratch/bench_w2batch.py   (CX_SW_BATCH_SPLITS=<n> changes the pixel-range splits per layer of the batch)"""
import os, sys, torch
sys.path.insert(0, '.')
from chexpert_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16; B = 256
ops.set_det_wgrad(True)
def t(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("CX_SW_BATCH_SPLITS=%s" % os.environ.get("CX_SW_BATCH_SPLITS", "default"))
for hw, n in ((40, 12), (20, 24), (10, 16)):
    items = []
    for i in range(n):
        g = torch.randn(B, hw, hw, 32, device=dev).to(bf)
        x = torch.randn(B, hw, hw, 128, device=dev).to(bf)
        items.append((g, x, torch.rand(128, device=dev) + 0.5, torch.rand(128, device=dev) - 0.5, torch.zeros(32, 128, 3, 3, device=dev)))
    def per_layer():
        ops.wgrad_defer_begin(dev)
        for g, x, pa, pb, dw in items:
            ops.conv_wgrad(g, x, dw, kh=3, kw=3, pad=1, x_prologue=ops.PRO_AFFINE_RELU, pa=pa, pb=pb)
        ops.wgrad_defer_flush(dev)
    def batched():
        ops.wgrad_defer_begin(dev)
        assert ops.conv3x3_wgrad_batch(items)
        ops.wgrad_defer_flush(dev)
    a, b = t(per_layer), t(batched)
    by = n * B * hw * hw * (32 + 128) * 2
    print("%dx%d  %2d layers: per layer %.1f us (%.1f each), batched %.1f us (%.1f each, %.2f TB/s)" % (hw, hw, n, a, a / n, b, b / n, by / b / 1e6))
