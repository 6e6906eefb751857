import sys, torch
sys.path.insert(0, '.')
from chexpert_amd import synth
from chexpert_amd.models import DenseNet
from oracle import nets, step
cfg = tuple(int(c) for c in sys.argv[1].split(',')); B = int(sys.argv[2]); S = int(sys.argv[3])
dev = torch.device('cuda:0')
spec = nets.densenet_spec(5, block_config=cfg)
sd = synth.fill_state_dict_(nets.zeros_state_dict(spec), 21)
if len(sys.argv) > 4 and sys.argv[4] == 'smooth':
    for k in sd:
        if k.endswith('.bias') and sd[k].dim()==1 and 'classifier' not in k: sd[k] = sd[k]*0 + 2.5
        if k.endswith('.weight') and sd[k].dim()==1: sd[k] = synth.uniform(7, sd[k].shape, 0.8, 1.2)
model = DenseNet(32, cfg, 64, num_classes=5); model.load_state_dict(sd); model.to(dev)
x, t = synth.xray_batch(1234, B, S), synth.targets(99, B, 5)
def run(q):
    s = {k: v.clone() for k, v in sd.items()}
    return step.train_step(lambda ss, xx: nets.densenet_forward(ss, xx, cfg, train=True, q=q), s, x, t)
lo, lg, go = run(None)
lq, lgq, gq = run(nets.bf16_storage)
model.train()
loss, logits = model.forward_backward(x.to(dev), t.to(dev))
print("logits rel mine-fp32 %.3e  bf16oracle-fp32 %.3e" % ((logits.cpu()-lg).abs().max()/lg.abs().max(), (lgq-lg).abs().max()/lg.abs().max()))
def cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a*b).sum()/(a.norm()*b.norm()+1e-30)), float(a.norm()/(b.norm()+1e-30))
names = [k for k, _ in model.named_parameters()]
for k, p in list(model.named_parameters())[::-1]:
    if not (k.endswith('conv1.weight') or k.endswith('conv2.weight') or 'conv0' in k or 'transition' in k or 'norm5' in k or 'classifier' in k or 'norm0' in k): continue
    c1, n1 = cos(p.grad.cpu(), go[k]); c2, n2 = cos(gq[k], go[k]); c3, n3 = cos(p.grad.cpu(), gq[k])
    print("%-55s mine/fp32 cos %.4f nr %.3f | bf16or/fp32 cos %.4f nr %.3f | mine/bf16or cos %.4f" % (k, c1, n1, c2, n2, c3))
