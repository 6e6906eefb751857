import os, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from chexpert_amd import synth
import test_resnet_gpu as T
dev = torch.device('cuda:0')
layers = tuple(int(a) for a in sys.argv[1].split(',')) if len(sys.argv) > 1 else (1, 3, 1, 1)
B, S, n_cls = 8, 128, 5
x, t = synth.xray_batch(1234, B, S), synth.targets(99, B, n_cls)
res = {}
for fuse in ("1", "0", "0"):
    os.environ["CHEXPERT_JOIN_FUSE"] = fuse
    model, _ = T._build(layers, n_cls, 21, dev, smooth=True)
    model.train()
    out = model(x.to(dev))
    loss = torch.nn.BCEWithLogitsLoss(reduction="none")(out, t.to(dev)).sum(1).mean(0)
    model.zero_grad(); loss.backward()
    key = fuse if fuse not in res else fuse + "b"
    res[key] = {k: p.grad.detach().float().cpu().clone() for k, p in model.named_parameters()}
names = list(res["0"].keys())[::-1]
for k in names:
    a, b, c = res["1"][k], res["0"][k], res["0b"][k]
    print("%-40s fused-vs-sep %.3e   sep-vs-sep %.3e   norm %.3e" % (k, (a - b).norm() / (b.norm() + 1e-20), (c - b).norm() / (b.norm() + 1e-20), b.norm()))
