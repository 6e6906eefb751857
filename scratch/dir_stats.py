"""Per-tensor statistics of the recorded gradient elements (tests/test_golden_smooth_gpu.py `_direction`) for one fixture: which metric
separates a corrupted tensor from bf16 noise."""
import json, os, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_golden_smooth_gpu as T
from chexpert_amd import synth
tag, copies = sys.argv[1], int(sys.argv[2])
rec = json.load(open('tests/golden/nets_smooth.json'))[tag]
dev = torch.device('cuda:0')
model, _ = T._make(tag, rec["n_classes"]); model = model.to(dev)
x8 = synth.xray_batch(rec["x_seed"], rec["B"], rec["S"]); t8 = synth.targets(rec["t_seed"], rec["B"], rec["n_classes"])
model.train(); model.zero_grad()
model.forward_backward(x8.repeat(copies, 1, 1, 1).to(dev), t8.repeat(copies, 1).to(dev))
gmax = max(r["l2"] for r in rec["grads"].values())
rows = []
for k, p in model.named_parameters():
    r = rec["grads"][k]
    if r["l2"] < 1e-3 * gmax: continue
    n = p.numel(); scale = r["l2"] / max(1.0, n ** 0.5)
    got = T._sampled(p.grad.detach().flatten(), n) / scale
    want = torch.tensor(r["head"] + r["samples"], dtype=torch.float64) / scale
    d = got - want
    cosk = float((got * want).sum() / (got.norm() * want.norm() + 1e-30))
    rows.append((float(d.abs().max()), float((d ** 2).mean().sqrt()), cosk, float(want.norm() / 4), k, n))
import numpy as np
mx = np.array([r[0] for r in rows]); rm = np.array([r[1] for r in rows]); ck = np.array([r[2] for r in rows])
print("%s x%d: %d tensors; max-element err: median %.3f p90 %.3f p99 %.3f max %.3f | rms err: median %.3f p90 %.3f p99 %.3f max %.3f | per-tensor cosine: median %.4f p10 %.4f p1 %.4f min %.4f" % (
    tag, copies, len(rows), np.median(mx), np.quantile(mx, .9), np.quantile(mx, .99), mx.max(), np.median(rm), np.quantile(rm, .9), np.quantile(rm, .99), rm.max(),
    np.median(ck), np.quantile(ck, .1), np.quantile(ck, .01), ck.min()))
for r in sorted(rows, key=lambda r: -r[1])[:8]:
    print("   rms %.3f max %.3f cos %.4f |want|rms %.2f  %s (%d)" % (r[1], r[0], r[2], r[3], r[4], r[5]))
for r in sorted(rows, key=lambda r: r[2])[:5]:
    print("   lowest cos %.4f rms %.3f  %s (%d)" % (r[2], r[1], r[4], r[5]))
