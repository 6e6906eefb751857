"""Build container only: the CPU restatement (oracle/) against the imported reference on the BASELINE configs[0] step
(densenet121, 320x320, bs=4, fp32, forward + BCE + backward + Adam; chexpert.py:159-164), same threads.  The reference cannot
travel to the GPU box, so bench.py's cpu_baseline times the oracle there; this shows the two run at the same speed here.
    python scratch/oracle_vs_reference_speed.py > profiles/r02_oracle_vs_reference_cpu.txt"""
import importlib.util
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)
mg.install_standins()
from models.attn_aug_conv import DenseNet          # noqa: E402  (the reference's, /root/reference is first on sys.path now)
from chexpert_amd import synth                     # noqa: E402
from oracle import nets, step                      # noqa: E402

torch.set_num_threads(8)
n_cls, B, S = 5, 4, 320
sd = synth.fill_state_dict_(nets.zeros_state_dict(nets.densenet_spec(n_cls)), 5)
x, t = synth.xray_batch(11, B, S), synth.targets(12, B, n_cls)


def timed(fn, n=6):
    fn()
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n


ref = DenseNet(32, (6, 12, 24, 16), 64, num_classes=n_cls)
ref.load_state_dict(sd, strict=True)
ref.train()
opt_r = torch.optim.Adam(ref.parameters(), lr=1e-4)
loss_fn = torch.nn.BCEWithLogitsLoss(reduction="none")


def ref_step():
    out = ref(x)
    loss = loss_fn(out, t).sum(1).mean(0)
    opt_r.zero_grad()
    loss.backward()
    opt_r.step()


sdo = {k: v.clone() for k, v in sd.items()}
for k in step.trainable(sdo):
    sdo[k].requires_grad_(True)
opt_o, _ = step.make_optimizer("adam", [sdo[k] for k in step.trainable(sdo)], 1e-4)
fwd = lambda s, xx: nets.densenet_forward(s, xx, train=True)
tr, to = timed(ref_step), timed(lambda: step.train_step(fwd, sdo, x, t, opt_o))
print("densenet121 320x320 bs=4 fp32, forward + loss + backward + Adam, %d threads (%s)" % (torch.get_num_threads(), torch.__version__))
print("reference (imported /root/reference, chexpert.py:159-164): %.3f s/step = %.2f img/s" % (tr, B / tr))
print("oracle restatement (oracle/nets.py + oracle/step.py):        %.3f s/step = %.2f img/s" % (to, B / to))
print("ratio oracle / reference: %.3f" % (to / tr))
