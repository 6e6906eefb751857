import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from chexpert_amd import synth
from chexpert_amd.models import Bottleneck, ResNet
from oracle import nets
from oracle.nets import _bn
dev = torch.device('cuda:0')
layers, B, S, n_cls = (1, 1, 1, 1), 4, 128, 5
spec = nets.resnet_spec(n_cls, layers=layers, attn=dict(k=.2, v=.1, nh=8), input_hw=(S, S))
sd = synth.fill_state_dict_(nets.zeros_state_dict(spec), 21)
for k in sd:
    if k.endswith(".bias") and not k.startswith("fc"): sd[k] = torch.full_like(sd[k], 1.0)
    if k.endswith(".weight") and sd[k].dim() == 1: sd[k] = synth.uniform(7, sd[k].shape, 0.8, 1.2)
model = ResNet(Bottleneck, list(layers), num_classes=n_cls, attn_params={"k": .2, "v": .1, "nh": 8, "relative": True, "input_dims": (S, S)})
model.load_state_dict(sd, strict=True); model = model.to(dev).eval()
x = synth.xray_batch(1234, B, S)
eng = model._eng()
ws = eng.forward(x.to(dev), False)
torch.cuda.synchronize()
nchw = lambda t: t.float().permute(0, 3, 1, 2).cpu()
rel = lambda a, b: (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)
# oracle pieces
w = lambda k: sd[k]
xx = F.conv2d(x, w("conv1.weight"), stride=2, padding=3)
xx = F.max_pool2d(F.relu(_bn(sd, "bn1", xx, False)), 3, 2, 1)
print("pool0", rel(nchw(ws.pool0), xx))
bi = 0
for L in (1, 2):
    p = "layer%d.0" % L
    s = 2 if L > 1 else 1
    y1 = F.conv2d(xx, w(p + ".conv1.weight"))
    t = ws.blk[bi]
    print(p, "y1", rel(nchw(t["y1"]), y1))
    a = F.relu(_bn(sd, p + ".bn1", y1, False))
    if p + ".conv2.weight" in sd:
        y2 = F.conv2d(a, w(p + ".conv2.weight"), stride=s, padding=1)
    else:
        from oracle import aaconv
        cw = w(p + ".conv2.conv.weight")
        conv_out = F.conv2d(a, cw, stride=s, padding=1)
        cc = cw.shape[0]
        print(p, "conv branch", rel(nchw(t["y2"][..., :cc]), conv_out))
        qkv = F.conv2d(a, w(p + ".conv2.in_proj_qkv.weight"), stride=s)
        print(p, "qkv", rel(nchw(t["QKV"]), qkv))
        y2 = nets._aa(sd, p + ".conv2", a, s, 8)
        print(p, "attn out", rel(nchw(t["y2"][..., cc:]), y2[:, cc:]), "y2 all", rel(nchw(t["y2"]), y2))
    print(p, "y2", rel(nchw(t["y2"]), y2))
    y = F.relu(_bn(sd, p + ".bn2", y2, False))
    y3 = F.conv2d(y, w(p + ".conv3.weight"))
    print(p, "y3", rel(nchw(t["y3"]), y3))
    y = _bn(sd, p + ".bn3", y3, False)
    xd = _bn(sd, p + ".downsample.1", F.conv2d(xx, w(p + ".downsample.0.weight"), stride=s), False)
    xx = F.relu(y + xd)
    print(p, "out", rel(nchw(t["out"]), xx))
    bi += 1
# ---- attention pieces from the engine's own QKV
t = ws.blk[1]; p = "layer2.0.conv2"
aa = model.layer2[0].conv2
qkv = nchw(t["QKV"])
import oracle.aaconv as oa
print([n for n in dir(oa) if not n.startswith('_')])
Bq, _, H, W = qkv.shape
dk, dv, nh = aa.dk, aa.dv, aa.nh
q, k, v = qkv.split([dk, dk, dv], dim=1)
dkh, dvh = dk // nh, dv // nh
fq = q.reshape(Bq, nh, dkh, H * W) * dkh ** -0.5
fk = k.reshape(Bq, nh, dkh, H * W); fv = v.reshape(Bq, nh, dvh, H * W)
logits = torch.matmul(fq.transpose(2, 3), fk)
logits = logits + oa.attention_logits(q * dkh ** -0.5 if False else qkv[:, :dk], sd[p + ".key_rel_h"], sd[p + ".key_rel_w"], nh) if False else logits
wts = None
try:
    full, wts = nets._aa(sd, p, torch.zeros(1), 1, nh, True)
except Exception as e:
    pass
O_eng = t["O"].cpu()          # (B, HW, dv)
print("O stats", O_eng.abs().max().item(), O_eng.shape)
# oracle aaconv2d on the same input a (recompute): get attention output before out_proj by using identity out_proj
xx0 = F.conv2d(x, w("conv1.weight"), stride=2, padding=3)
xx0 = F.max_pool2d(F.relu(_bn(sd, "bn1", xx0, False)), 3, 2, 1)
p1 = "layer1.0"
y = F.relu(_bn(sd, p1 + ".bn1", F.conv2d(xx0, w(p1 + ".conv1.weight")), False))
y = F.relu(_bn(sd, p1 + ".bn2", F.conv2d(y, w(p1 + ".conv2.weight"), padding=1), False))
y = _bn(sd, p1 + ".bn3", F.conv2d(y, w(p1 + ".conv3.weight")), False)
xd = _bn(sd, p1 + ".downsample.1", F.conv2d(xx0, w(p1 + ".downsample.0.weight")), False)
x1 = F.relu(y + xd)
a = F.relu(_bn(sd, "layer2.0.bn1", F.conv2d(x1, w("layer2.0.conv1.weight")), False))
eye = torch.eye(dv).view(dv, dv, 1, 1)
att = oa.aaconv2d(a, None, sd[p + ".in_proj_qkv.weight"], eye, sd[p + ".key_rel_h"], sd[p + ".key_rel_w"], stride=2, dk=dk, dv=dv, nh=nh)
O_ref = att.permute(0, 2, 3, 1).reshape(Bq, H * W, dv)
print("O vs oracle", rel(O_eng, O_ref))
outp = F.conv2d(att, sd[p + ".out_proj.weight"])
print("outproj(O_eng) vs y2 slice", rel(nchw(t["y2"][..., 120:]), F.conv2d(O_eng.reshape(Bq, H, W, dv).permute(0, 3, 1, 2), sd[p + ".out_proj.weight"])))
print("outproj oracle vs y2 slice", rel(nchw(t["y2"][..., 120:]), outp))
