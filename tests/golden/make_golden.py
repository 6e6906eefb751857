#!/usr/bin/env python
"""Generate tests/golden/* by running the REAL reference (/root/reference) on CPU.

Run in the build container only (the reference does not travel to the GPU box):

    python tests/golden/make_golden.py            # all fixtures
    python tests/golden/make_golden.py aaconv     # a subset

What it does
  * provides a build-owned stand-in for the four torchvision-0.3.0 symbols the reference imports
    (`conv1x1`, `conv3x3`, `_DenseLayer`, `_DenseBlock`; torchvision is not installed here) and
    empty stubs for `torchvision.transforms`, `torchvision.models.{densenet121,resnet152}` and
    `tensorboardX.SummaryWriter` so that `chexpert.py` imports;
  * builds the reference models, loads hash-filled state_dicts (`chexpert_amd.synth`, strict=True,
    which also pins the state_dict key names / shapes of SURVEY.md section 8b), runs them on hash-made
    inputs and records small outputs: logits, loss, per-parameter grad norms + leading elements,
    BN running stats, AAConv outputs / grads / attention rows, Grad-CAM maps, AUROC cases,
    parameter counts;
  * prints the deviation of `oracle/` from the reference on the same inputs (informational; the
    committed check is tests/test_oracle_golden.py against the fixtures written here).

Only data (inputs' seeds and expected outputs) is written; no reference source text is stored.
"""
import json
from collections import OrderedDict
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from chexpert_amd import synth  # noqa: E402


# ----------------------------------------------------------------------------- stand-ins
def install_standins():
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvr = types.ModuleType("torchvision.models.resnet")
    tvd = types.ModuleType("torchvision.models.densenet")
    tvt = types.ModuleType("torchvision.transforms")

    def conv3x3(cin, cout, stride=1, groups=1, dilation=1):
        return nn.Conv2d(cin, cout, 3, stride, dilation, dilation, groups, bias=False)

    def conv1x1(cin, cout, stride=1):
        return nn.Conv2d(cin, cout, 1, stride, bias=False)

    class _DenseLayer(nn.Sequential):          # torchvision 0.3.0 semantics
        def __init__(self, num_input_features, growth_rate, bn_size, drop_rate):
            super().__init__()
            mid = bn_size * growth_rate
            self.add_module("norm1", nn.BatchNorm2d(num_input_features))
            self.add_module("relu1", nn.ReLU(inplace=True))
            self.add_module("conv1", nn.Conv2d(num_input_features, mid, 1, 1, bias=False))
            self.add_module("norm2", nn.BatchNorm2d(mid))
            self.add_module("relu2", nn.ReLU(inplace=True))
            self.add_module("conv2", nn.Conv2d(mid, growth_rate, 3, 1, 1, bias=False))
            self.drop_rate = drop_rate

        def forward(self, x):
            new = super().forward(x)
            if self.drop_rate > 0:
                new = nn.functional.dropout(new, self.drop_rate, self.training)
            return torch.cat([x, new], 1)

    class _DenseBlock(nn.Sequential):
        def __init__(self, num_layers, num_input_features, bn_size, growth_rate, drop_rate):
            super().__init__()
            for i in range(num_layers):
                self.add_module("denselayer%d" % (i + 1),
                                _DenseLayer(num_input_features + i * growth_rate, growth_rate, bn_size, drop_rate))

    tvr.conv1x1, tvr.conv3x3 = conv1x1, conv3x3
    tvd._DenseLayer, tvd._DenseBlock = _DenseLayer, _DenseBlock
    tvm.resnet, tvm.densenet = tvr, tvd
    tvm.densenet121 = tvm.resnet152 = None
    tv.models, tv.transforms = tvm, tvt
    tbx = types.ModuleType("tensorboardX")
    tbx.SummaryWriter = object
    sys.modules.update({"torchvision": tv, "torchvision.models": tvm, "torchvision.models.resnet": tvr,
                        "torchvision.models.densenet": tvd, "torchvision.transforms": tvt, "tensorboardX": tbx})
    sys.path.insert(0, REF)


# ----------------------------------------------------------------------------- helpers
def summarise(t: torch.Tensor):
    """Small, layout-sensitive summary of a tensor: l2, sum, first 8, 8 strided samples."""
    f = t.detach().double().flatten()
    n = f.numel()
    idx = (torch.arange(8) * max(1, n // 8) + (n // 16)).clamp(max=n - 1)
    return {"l2": float(f.norm()), "sum": float(f.sum()), "head": f[:8].tolist(), "samples": f[idx].tolist(),
            "n": int(n)}


def filled_sd(spec, seed):
    from oracle import nets
    return synth.fill_state_dict_(nets.zeros_state_dict(spec), seed)


def run_net(model, sd, x, target, tag, out, oracle_fwd, taps_ref=None):
    """Reference eval logits, then one reference train step (chexpert.py:159-163); compare oracle."""
    from oracle import step as ostep
    model.load_state_dict(sd, strict=True)
    rec = {}
    model.eval()
    with torch.no_grad():
        rec["logits_eval"] = model(x).tolist()
    model.train()
    logits = model(x)
    loss = nn.BCEWithLogitsLoss(reduction="none")(logits, target).sum(1).mean(0)
    model.zero_grad()
    loss.backward()
    rec["logits_train"] = logits.tolist()
    rec["loss"] = float(loss)
    rec["grads"] = {k: summarise(p.grad) for k, p in model.named_parameters()}
    after = model.state_dict()
    rec["running"] = {k: summarise(v) for k, v in after.items() if k.endswith(("running_mean", "running_var"))
                      and (k.count(".") <= 2 or "norm5" in k or "denselayer1." in k)}
    rec["n_params"] = sum(p.numel() for p in model.parameters())
    rec["keys"] = len(after)
    out[tag] = rec
    # oracle deviation (informational)
    sd_e = {k: v.clone() for k, v in sd.items()}
    with torch.no_grad():
        le = oracle_fwd(sd_e, x, train=False)
    sd_t = {k: v.clone() for k, v in sd.items()}
    lo, lg, grads = ostep.train_step(lambda s, xx: oracle_fwd(s, xx, train=True), sd_t, x, target)
    gmax = max(rec["grads"][k]["l2"] for k in grads)
    dev_g = max(abs(float(grads[k].double().norm()) - rec["grads"][k]["l2"]) / (rec["grads"][k]["l2"] + 1e-3 * gmax)
                for k in grads)
    print("[%s] oracle-vs-reference: eval %.2e train %.2e loss %.2e grad-l2(rel) %.2e | logits absmax %.3f" % (
        tag, (le - torch.tensor(rec["logits_eval"])).abs().max(), (lg - logits).abs().max(),
        abs(float(lo) - rec["loss"]), dev_g, logits.abs().max()))


# ----------------------------------------------------------------------------- fixtures
def gen_aaconv(out_dir):
    from models.attn_aug_conv import AAConv2d
    from oracle.aaconv import aaconv2d
    cases = {
        "small_s2": dict(B=2, cin=16, cout=24, k=3, stride=2, dk=8, dv=8, nh=4, hin=(10, 14)),
        "small_s1": dict(B=2, cin=8, cout=16, k=3, stride=1, dk=16, dv=4, nh=2, hin=(6, 5)),
        "t1_like": dict(B=1, cin=256, cout=128, k=3, stride=2, dk=160, dv=8, nh=8, hin=(16, 16)),
        "attn_only": dict(B=1, cin=8, cout=8, k=1, stride=1, dk=8, dv=8, nh=2, hin=(4, 6)),
    }
    arrays, meta = {}, {}
    for name, c in cases.items():
        H, W = c["hin"][0] // c["stride"], c["hin"][1] // c["stride"]
        torch.manual_seed(0)
        m = AAConv2d(c["cin"], c["cout"], c["k"], c["stride"], c["dk"], c["dv"], c["nh"], True, (H, W))
        sd = m.state_dict()
        synth.fill_state_dict_(sd, 7)
        m.load_state_dict(sd)
        x = synth.uniform(11, (c["B"], c["cin"]) + tuple(c["hin"]), -1.5, 1.5).requires_grad_(True)
        gy = synth.uniform(13, (c["B"], c["cout"], H, W), -1, 1)
        y = m(x)
        (y * gy).sum().backward()
        arrays[name + ".y"] = y.detach().numpy()
        arrays[name + ".dx"] = x.grad.numpy()
        for k, p in m.named_parameters():
            arrays[name + ".d_" + k] = p.grad.numpy()
        arrays[name + ".weights_rows"] = m.weights.detach()[:, :, :3].numpy()     # first 3 query rows / head
        meta[name] = dict(c, keys={k: list(v.shape) for k, v in sd.items()})
        # oracle deviation
        x2 = x.detach().clone().requires_grad_(True)
        sd2 = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        y2, P = aaconv2d(x2, sd2.get("conv.weight"), sd2["in_proj_qkv.weight"], sd2["out_proj.weight"],
                         sd2["key_rel_h"], sd2["key_rel_w"], stride=c["stride"], dk=c["dk"], dv=c["dv"], nh=c["nh"],
                         return_weights=True)
        (y2 * gy).sum().backward()
        print("[aaconv %s] oracle-vs-reference: y %.2e dx %.2e P %.2e d_rel_h %.2e" % (
            name, (y2 - y).abs().max(), (x2.grad - x.grad).abs().max(), (P - m.weights).abs().max(),
            (sd2["key_rel_h"].grad - m.key_rel_h.grad).abs().max()))
    np.savez_compressed(os.path.join(out_dir, "aaconv.npz"), **arrays)
    json.dump(meta, open(os.path.join(out_dir, "aaconv.json"), "w"), indent=1)


def gen_nets(out_dir, which):
    from models.attn_aug_conv import DenseNet, ResNet, Bottleneck, BasicBlock, WideResNet
    from models.efficientnet import construct_model
    from oracle import nets
    out = {}
    path = os.path.join(out_dir, "nets.json")
    if os.path.exists(path):
        out = json.load(open(path))
    n_cls = 5
    attn = dict(k=0.2, v=0.1, nh=8)

    def ref_attn(hw):
        return {"k": 0.2, "v": 0.1, "nh": 8, "relative": True, "input_dims": hw}   # chexpert.py:476

    jobs = {
        "densenet121_320_b2": lambda: (DenseNet(32, (6, 12, 24, 16), 64, num_classes=n_cls),
                                       nets.densenet_spec(n_cls), 2, 320,
                                       lambda s, x, train: nets.densenet_forward(s, x, train=train)),
        "densenet_tiny_64_b3": lambda: (DenseNet(32, (2, 2, 2, 2), 64, num_classes=n_cls),
                                        nets.densenet_spec(n_cls, block_config=(2, 2, 2, 2)), 3, 64,
                                        lambda s, x, train: nets.densenet_forward(s, x, (2, 2, 2, 2), train=train)),
        "aadensenet121_320_b1": lambda: (DenseNet(32, (6, 12, 24, 16), 64, num_classes=n_cls, attn_params=ref_attn((320, 320))),
                                         nets.densenet_spec(n_cls, attn=attn), 1, 320,
                                         lambda s, x, train: nets.densenet_forward(s, x, train=train, nh=8)),
        "aadensenet_tiny_64_b2": lambda: (DenseNet(32, (6, 4, 2, 2), 64, num_classes=n_cls, attn_params=ref_attn((64, 64))),
                                          nets.densenet_spec(n_cls, block_config=(6, 4, 2, 2), attn=attn, input_hw=(64, 64)), 2, 64,
                                          lambda s, x, train: nets.densenet_forward(s, x, (6, 4, 2, 2), train=train, nh=8)),
        "resnet152_320_b2": lambda: (ResNet(Bottleneck, [3, 8, 36, 3], num_classes=n_cls),
                                     nets.resnet_spec(n_cls), 2, 320,
                                     lambda s, x, train: nets.resnet_forward(s, x, train=train)),
        "aaresnet152_320_b1": lambda: (ResNet(Bottleneck, [3, 8, 36, 3], num_classes=n_cls, attn_params=ref_attn((320, 320))),
                                       nets.resnet_spec(n_cls, attn=attn), 1, 320,
                                       lambda s, x, train: nets.resnet_forward(s, x, train=train, nh=8)),
        "resnet_tiny_64_b2": lambda: (ResNet(Bottleneck, [1, 1, 1, 1], num_classes=n_cls),
                                      nets.resnet_spec(n_cls, layers=(1, 1, 1, 1)), 2, 64,
                                      lambda s, x, train: nets.resnet_forward(s, x, (1, 1, 1, 1), train=train)),
        # BasicBlock networks of the CIFAR harness (models/test_model.py; attn_aug_conv.py:107-156, :311-404)
        "resnet18_128_b4": lambda: (ResNet(BasicBlock, [2, 2, 2, 2], num_classes=n_cls),
                                    nets.basic_resnet_spec(n_cls), 4, 128,
                                    lambda s, x, train: nets.basic_resnet_forward(s, x, train=train)),
        "wrn16_4_32_b8": lambda: (WideResNet(BasicBlock, 16, 4, num_classes=n_cls),
                                  nets.basic_resnet_spec(n_cls, wide=(16, 4)), 8, 32,
                                  lambda s, x, train: nets.basic_resnet_forward(s, x, wide=(16, 4), train=train)),
        # ... and their attention-augmented forms (AAConv2d as conv1 of the BasicBlocks from stage 2 on; the harness's --attn)
        "aawrn16_4_32_b8": lambda: (WideResNet(BasicBlock, 16, 4, num_classes=n_cls, attn_params=ref_attn((32, 32))),
                                    nets.basic_resnet_spec(n_cls, wide=(16, 4), attn=attn, input_hw=(32, 32)), 8, 32,
                                    lambda s, x, train: nets.basic_resnet_forward(s, x, wide=(16, 4), train=train, nh=8)),
        "aaresnet18_128_b4": lambda: (ResNet(BasicBlock, [2, 2, 2, 2], num_classes=n_cls, attn_params=ref_attn((128, 128))),
                                      nets.basic_resnet_spec(n_cls, attn=attn, input_hw=(128, 128)), 4, 128,
                                      lambda s, x, train: nets.basic_resnet_forward(s, x, train=train, nh=8)),
        "efficientnet-b0_224_b2": lambda: (construct_model("efficientnet-b0", n_cls),
                                           nets.efficientnet_spec("efficientnet-b0", n_cls), 2, 224,
                                           lambda s, x, train: nets.efficientnet_forward(s, x, "efficientnet-b0", train=train)),
        "efficientnet-b4_380_b2": lambda: (construct_model("efficientnet-b4", n_cls),
                                           nets.efficientnet_spec("efficientnet-b4", n_cls), 2, 380,
                                           lambda s, x, train: nets.efficientnet_forward(s, x, "efficientnet-b4", train=train)),
    }
    for tag, job in jobs.items():
        if which and not any(w in tag for w in which):
            continue
        model, spec, B, S, fwd = job()
        if "efficientnet" in tag:       # deterministic part only: DropConnect / Dropout masks are RNG-bound
            for m in model.modules():
                if isinstance(m, (nn.Dropout, nn.Dropout3d)):
                    m.p = 0.0
        sd = filled_sd(spec, 21)
        x = synth.xray_batch(1234, B, S)
        t = synth.targets(99, B, n_cls)
        run_net(model, sd, x, t, tag, out, fwd)
        out[tag].update(B=B, S=S, n_classes=n_cls, sd_seed=21, x_seed=1234, t_seed=99)
        json.dump(out, open(path, "w"))
    # parameter counts pinned by the reference constructors
    counts = {}
    counts["densenet121@14"] = sum(p.numel() for p in DenseNet(32, (6, 12, 24, 16), 64, num_classes=14).parameters())
    counts["densenet121@1000"] = sum(p.numel() for p in DenseNet(32, (6, 12, 24, 16), 64).parameters())
    counts["aaresnet152@5"] = sum(p.numel() for p in ResNet(Bottleneck, [3, 8, 36, 3], num_classes=5,
                                                            attn_params=ref_attn((320, 320))).parameters())
    json.dump(counts, open(os.path.join(out_dir, "param_counts.json"), "w"), indent=1)


def gen_smooth(out_dir, which, yard_only=False):
    """Well-conditioned fixtures from the REAL reference at B = 8 (tests/golden/nets_smooth.json): the state of
    `synth.smooth_state_dict_` (BatchNorm gains in [0.8, 1.2], biases 2.5 / 1.0, kaiming-scale convolutions).  These are the
    fixtures the bf16 path is held to north_star's 1e-2 on, with literal bounds (the hash-weight fixtures of gen_nets amplify
    storage rounding chaotically and only serve as order-of-magnitude checks).  A batch made of copies of these 8 images has
    the same batch statistics, so the same records pin the BASELINE batch sizes (256 / 128 / 64)."""
    import gc
    from models.attn_aug_conv import DenseNet, ResNet, Bottleneck
    from models.efficientnet import construct_model
    from oracle import nets
    path = os.path.join(out_dir, "nets_smooth.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    n_cls = 5
    attn = dict(k=0.2, v=0.1, nh=8)

    def ref_attn(hw):
        return {"k": 0.2, "v": 0.1, "nh": 8, "relative": True, "input_dims": hw}   # chexpert.py:476

    jobs = {
        "densenet121_320_b8": lambda: (DenseNet(32, (6, 12, 24, 16), 64, num_classes=n_cls), nets.densenet_spec(n_cls), 8, 320, 2.5,
                                       lambda s, x, train, q=None: nets.densenet_forward(s, x, train=train, q=q)),
        "aadensenet121_320_b8": lambda: (DenseNet(32, (6, 12, 24, 16), 64, num_classes=n_cls, attn_params=ref_attn((320, 320))),
                                         nets.densenet_spec(n_cls, attn=attn), 8, 320, 2.5,
                                         lambda s, x, train, q=None: nets.densenet_forward(s, x, train=train, nh=8, q=q)),
        "resnet152_320_b8": lambda: (ResNet(Bottleneck, [3, 8, 36, 3], num_classes=n_cls), nets.resnet_spec(n_cls), 8, 320, 1.0,
                                     lambda s, x, train, q=None: nets.resnet_forward(s, x, train=train, q=q)),
        "efficientnet-b4_380_b8": lambda: (construct_model("efficientnet-b4", n_cls), nets.efficientnet_spec("efficientnet-b4", n_cls),
                                           8, 380, 1.0,
                                           lambda s, x, train, q=None: nets.efficientnet_forward(s, x, "efficientnet-b4", train=train)),
        "efficientnet-b0_224_b8": lambda: (construct_model("efficientnet-b0", n_cls), nets.efficientnet_spec("efficientnet-b0", n_cls),
                                           8, 224, 1.0,
                                           lambda s, x, train, q=None: nets.efficientnet_forward(s, x, "efficientnet-b0", train=train)),
        "aaresnet152_320_b8": lambda: (ResNet(Bottleneck, [3, 8, 36, 3], num_classes=n_cls, attn_params=ref_attn((320, 320))),
                                       nets.resnet_spec(n_cls, attn=attn), 8, 320, 1.0,
                                       lambda s, x, train, q=None: nets.resnet_forward(s, x, train=train, nh=8, q=q)),
        # the CIFAR harness's Densenet-BC (models/test_model.py:304-306: DenseNet(k, ((L-4)//6,)*3, 2k)) at its default growth 12:
        # L = 40 (6 layers per block) and the default L = 100 (16 per block), 32x32 images
        "densenetbc_k12_L40_32_b8": lambda: (DenseNet(12, (6, 6, 6), 24, num_classes=n_cls),
                                             nets.densenet_spec(n_cls, growth=12, block_config=(6, 6, 6), init_features=24), 8, 32, 2.5,
                                             lambda s, x, train, q=None: nets.densenet_forward(s, x, (6, 6, 6), train=train, q=q)),
        "densenetbc_k12_L100_32_b8": lambda: (DenseNet(12, (16, 16, 16), 24, num_classes=n_cls),
                                              nets.densenet_spec(n_cls, growth=12, block_config=(16, 16, 16), init_features=24), 8, 32, 2.5,
                                              lambda s, x, train, q=None: nets.densenet_forward(s, x, (16, 16, 16), train=train, q=q)),
        # ... and with the harness's --attn defaults (k 0.2, v 0.1, 8 heads): both transitions are InstanceNorm -> ReLU ->
        # AAConv2d(3x3, stride 2) with dk 160, dv 8 on 16x16 / 8x8 maps
        "aadensenetbc_k12_L100_32_b8": lambda: (DenseNet(12, (16, 16, 16), 24, num_classes=n_cls, attn_params=ref_attn((32, 32))),
                                                nets.densenet_spec(n_cls, growth=12, block_config=(16, 16, 16), init_features=24,
                                                                   attn=attn, input_hw=(32, 32)), 8, 32, 2.5,
                                                lambda s, x, train, q=None: nets.densenet_forward(s, x, (16, 16, 16), train=train, nh=8, q=q)),
        # ... and with the value-channel ratio of the reference's CIFAR result rows (models/readme.md:34-38: Nh 8, k 0.2, v 0.7):
        # heads of 9 / 13 value channels, out-projections of 72 / 104 channels
        "aadensenetbcv07_k12_L100_32_b8": lambda: (DenseNet(12, (16, 16, 16), 24, num_classes=n_cls,
                                                            attn_params=dict(ref_attn((32, 32)), v=0.7)),
                                                   nets.densenet_spec(n_cls, growth=12, block_config=(16, 16, 16), init_features=24,
                                                                      attn=dict(attn, v=0.7), input_hw=(32, 32)), 8, 32, 2.5,
                                                   lambda s, x, train, q=None: nets.densenet_forward(s, x, (16, 16, 16), train=train, nh=8, q=q)),
    }
    for tag, job in jobs.items():
        if which and not any(w in tag for w in which):
            continue
        model, spec, B, S, bias, fwd = job()
        if "efficientnet" in tag:       # deterministic part only: DropConnect / Dropout masks are RNG-bound
            for m in model.modules():
                if isinstance(m, (nn.Dropout, nn.Dropout3d)):
                    m.p = 0.0
        sd = synth.smooth_state_dict_(filled_sd(spec, 21), bias)
        x = synth.xray_batch(1234, B, S)
        t = synth.targets(99, B, n_cls)
        if not yard_only:              # (`yardstick <tags>`: the reference's records stay, only the storage yardstick below is renewed)
            run_net(model, sd, x, t, tag, out, fwd)
            out[tag].update(B=B, S=S, n_classes=n_cls, sd_seed=21, x_seed=1234, t_seed=99, smooth_bias=bias)
        del model
        gc.collect()
        if "efficientnet" not in tag:
            # What bf16 STORAGE alone does to this fixture: the fp32 oracle with nothing but the tensors the HIP path stores rounded to
            # bf16, against the reference's records -- train logits, loss, gradient norms of the weights / of the norm parameters.  An
            # ill-conditioned fixture (aaresnet152: 47 softmax layers) is held to a multiple of THIS, not to a hand-set number.
            from oracle import step as ostep
            lo_q, lq, gq = ostep.train_step(lambda s, xx: fwd(s, xx, train=True, q=nets.bf16_storage), {k: v.clone() for k, v in sd.items()}, x, t)
            want = torch.tensor(out[tag]["logits_train"])
            out[tag]["bf16_storage_logits_rel"] = float((lq - want).abs().max() / want.abs().max())
            gmax = max(r["l2"] for r in out[tag]["grads"].values())
            dev_w, dev_n = 0.0, 0.0
            for k, r in out[tag]["grads"].items():
                if r["l2"] < 1e-3 * gmax or k not in gq:
                    continue
                d = abs(float(gq[k].double().norm()) / r["l2"] - 1.0)
                if gq[k].dim() > 1:
                    dev_w = max(dev_w, d)
                else:
                    dev_n = max(dev_n, d)
            out[tag]["bf16_storage_yardstick"] = {"logits": out[tag]["bf16_storage_logits_rel"],
                                                  "loss": abs(float(lo_q) - out[tag]["loss"]) / abs(out[tag]["loss"]),
                                                  "weight_grad_norm": dev_w, "norm_grad_norm": dev_n}
            print("[%s] storage-rounded oracle vs reference: %s" % (tag, out[tag]["bf16_storage_yardstick"]))
        json.dump(out, open(path, "w"))


def gen_options(out_dir):
    """Constructor options beyond the chexpert.py defaults, from the REAL reference on small networks (tests/golden/options.json;
    tests/test_oracle_golden.py holds the oracle's branches to them): ResNet `replace_stride_with_dilation`, `groups` +
    `width_per_group`, wide Bottlenecks (attn_aug_conv.py:218-220, :168, :183, :266-271), DenseNet `drop_rate` (:453, :479-481; eval
    mode, and train mode with the keep decisions INJECTED through F.dropout -- torch's own random stream is not reproducible
    elsewhere), `attn_params['relative'] = False` (:38, :76) and a value ratio of 0.4 (:417-427).  The parameter shapes are part of
    the fixture: the oracle is driven by the state_dict alone."""
    from models.attn_aug_conv import DenseNet, ResNet, Bottleneck
    from oracle import nets
    n_cls, out = 5, {}
    cfg = (2, 2, 2, 2)

    def attn(v=0.1, relative=True, hw=(64, 64)):
        return {"k": 0.2, "v": v, "nh": 8, "relative": relative, "input_dims": hw}

    P_DROP, SEED = 0.3, 777
    jobs = {
        "resnet_dilate_64_b2": (lambda: ResNet(Bottleneck, [1, 2, 1, 1], num_classes=n_cls, replace_stride_with_dilation=[False, True, True]), 2, 64,
                                lambda s, x, train: nets.resnet_forward(s, x, (1, 2, 1, 1), train=train, dilate=(False, True, True))),
        "resnet_groups4_w16_64_b2": (lambda: ResNet(Bottleneck, [1, 1, 1, 1], num_classes=n_cls, groups=4, width_per_group=16), 2, 64,
                                     lambda s, x, train: nets.resnet_forward(s, x, (1, 1, 1, 1), train=train)),
        "resnet_wide128_64_b2": (lambda: ResNet(Bottleneck, [1, 1, 1, 1], num_classes=n_cls, width_per_group=128), 2, 64,
                                 lambda s, x, train: nets.resnet_forward(s, x, (1, 1, 1, 1), train=train)),
        "densenet_drop_64_b4": (lambda: DenseNet(32, cfg, 64, drop_rate=P_DROP, num_classes=n_cls), 4, 64, None),
        "aadensenet_norel_64_b2": (lambda: DenseNet(32, (6, 4, 2, 2), 64, num_classes=n_cls, attn_params=attn(relative=False)), 2, 64,
                                   lambda s, x, train: nets.densenet_forward(s, x, (6, 4, 2, 2), train=train, nh=8)),
        "aadensenet_v04_64_b2": (lambda: DenseNet(32, (6, 4, 2, 2), 64, num_classes=n_cls, attn_params=attn(v=0.4)), 2, 64,
                                 lambda s, x, train: nets.densenet_forward(s, x, (6, 4, 2, 2), train=train, nh=8)),
    }
    for tag, (make, B, S, fwd) in jobs.items():
        model = make()
        shapes = [(k, list(v.shape)) for k, v in model.state_dict().items()]
        spec = OrderedDict((k, tuple(sh)) for k, sh in shapes)
        sd = filled_sd(spec, 21)
        x = synth.xray_batch(1234, B, S)
        t = synth.targets(99, B, n_cls)
        if fwd is None:
            # drop_rate: eval applies no dropout; train with injected keep decisions (block b, layer l in forward order)
            calls = []
            real_dropout = nn.functional.dropout

            def injected(y, p=0.5, training=True, inplace=False):
                assert training and abs(p - P_DROP) < 1e-12
                i = len(calls)
                b, l = i // cfg[0] + 1, i % cfg[0] + 1          # (every block of this fixture has two layers)
                calls.append((b, l))
                Bn, C_, H_, W_ = y.shape
                keep = nets.drop_keep(SEED, (b - 1) * 256 + (l - 1), (Bn, H_, W_, C_), p).permute(0, 3, 1, 2)
                return torch.where(keep, y / (1 - p), torch.zeros(()))

            def drop(b, l, y):
                Bn, C_, H_, W_ = y.shape
                keep = nets.drop_keep(SEED, (b - 1) * 256 + (l - 1), (Bn, H_, W_, C_), P_DROP).permute(0, 3, 1, 2)
                return torch.where(keep, y / (1 - P_DROP), torch.zeros(()))
            fwd = lambda s_, xx, train: nets.densenet_forward(s_, xx, cfg, train=train, drop=drop)
            nn.functional.dropout = lambda y, p=0.5, training=True, inplace=False: (injected(y, p, training, inplace) if training else y)
            try:
                run_net(model, sd, x, t, tag, out, fwd)
            finally:
                nn.functional.dropout = real_dropout
            assert calls == [(b, l) for b in range(1, 5) for l in range(1, cfg[0] + 1)], calls
            out[tag].update(drop_rate=P_DROP, drop_seed=SEED)
        else:
            run_net(model, sd, x, t, tag, out, fwd)
        out[tag].update(B=B, S=S, n_classes=n_cls, sd_seed=21, x_seed=1234, t_seed=99, shapes=shapes)
        del model
    json.dump(out, open(os.path.join(out_dir, "options.json"), "w"))


def gen_dataset(out_dir):
    """The dataset logic of the reference on a generated table (tests/golden/dataset.json): U-Ones processing with and without a
    data filter (dataset.py:134-153), the `vis` subset (:50-68), `mini_data`, `test` mode (:33-37), item labels / source indices
    (:73-89) and extract_patient_ids (:156-160).  The table itself (hash-made, the reference's csv columns) is part of the fixture.
    torch >= 2.6 refuses to unpickle the DataFrame the reference caches with torch.save (weights_only default): torch.load is
    wrapped to pass weights_only=False -- an environment adaptation, every dataset statement executed is the reference's own."""
    import tempfile
    import pandas as pd
    import dataset as ref_ds
    _load = torch.load
    ref_ds.torch.load = lambda f, *a, **k: _load(f, *a, **dict(k, weights_only=False))
    C = ref_ds.ChexpertSmall
    cols = ["Path", "Sex", "Age", "Frontal/Lateral", "AP/PA"] + C.attr_all_names

    def table(n, seed, split, complete):
        u = synth.uniform01(seed, n * (len(C.attr_all_names) + 2)).reshape(n, -1)
        rows = []
        for i in range(n):
            lat = u[i, 0] < 0.25
            labs = []
            for j in range(len(C.attr_all_names)):
                v = u[i, 2 + j]
                if complete:
                    labs.append(1.0 if v < 0.3 else 0.0)
                else:
                    labs.append(1.0 if v < 0.22 else (0.0 if v < 0.5 else (-1.0 if v < 0.65 else None)))
            rows.append(["%s/%s/patient%05d/study%d/view1_%s.jpg" % (C.dir_name, split, 100 + i // 2, 1 + i % 2, "lateral" if lat else "frontal"),
                         "Male" if u[i, 1] < 0.5 else "Female", 20 + int(60 * u[i, 1]), "Lateral" if lat else "Frontal",
                         None if lat else ("AP" if u[i, 1] < 0.7 else "PA")] + labs)
        return rows

    train_rows, valid_rows = table(48, 4242, "train", False), table(40, 4343, "valid", True)
    rec = {"columns": cols, "train_rows": train_rows, "valid_rows": valid_rows, "attr_names": C.attr_names}
    flt = {"Frontal/Lateral": "Frontal"}
    for tag, data_filter in (("plain", None), ("filtered", flt)):
        with tempfile.TemporaryDirectory() as root:
            d = os.path.join(root, C.dir_name)
            os.makedirs(d)
            pd.DataFrame(train_rows, columns=cols).to_csv(os.path.join(d, "train.csv"), index=False)
            pd.DataFrame(valid_rows, columns=cols).to_csv(os.path.join(d, "valid.csv"), index=False)
            tr = C(root, "train", data_filter=data_filter)
            r = {"train_index": [int(i) for i in tr.data.index], "train_labels": tr.data[C.attr_names].values.astype(float).tolist()}
            # labels / source index of items as __getitem__ returns them (:81-87), without opening the image file
            r["train_items"] = [[tr.data.iloc[k, tr.attr_idxs].values.astype(np.float32).tolist(), int(tr.data.index[k])] for k in (0, 5, len(tr) - 1)]
            if data_filter is None:
                mini = C(root, "train", mini_data=7)
                r["mini_len"] = len(mini)
                va = C(root, "valid")
                r["valid_len"], r["valid_labels_head"] = len(va), va.data[C.attr_names].values[:6].astype(float).tolist()
                vis = C(root, "vis")
                r["vis_attrs"], r["vis_idxs"] = vis.vis_attrs, [[int(i) for i in g] for g in vis.vis_idxs]
                r["vis_index"] = [int(i) for i in vis.data.index]
                r["patient_ids"] = ref_ds.extract_patient_ids(va, [0, 3, 17]).tolist()
                tcsv = os.path.join(root, "paths.csv")
                pd.DataFrame({"Path": [row[0] for row in valid_rows[:5]]}).to_csv(tcsv, index=False)
                te = C(tcsv, "test")
                r["test_len"], r["test_labels"] = len(te), te.data[C.attr_names].values.astype(float).tolist()
            rec[tag] = r
    json.dump(rec, open(os.path.join(out_dir, "dataset.json"), "w"))
    print("[dataset] %d train rows -> %d after the Frontal filter; vis groups %s" % (
        len(rec["plain"]["train_index"]), len(rec["filtered"]["train_index"]), [len(g) for g in rec["plain"]["vis_idxs"]]))


def gen_gradcam(out_dir):
    """Runs the reference `grad_cam` itself.  torch 2.10 rejects its in-place normalisation of an
    autograd view (chexpert.py:290-294), so `chexpert.F.relu` alone is wrapped to hand back a detached
    tensor at :285 -- the values are unchanged and every other statement is the reference's own."""
    import chexpert as ref
    from models.attn_aug_conv import DenseNet
    from oracle import nets, gradcam

    class _F:
        def __getattr__(self, k):
            return getattr(torch.nn.functional, k)

        @staticmethod
        def relu(t, *a, **kw):
            return torch.nn.functional.relu(t, *a, **kw).detach()
    ref.F = _F()
    cfg = (2, 2, 2, 2)
    model = DenseNet(32, cfg, 64, num_classes=5)
    sd = filled_sd(nets.densenet_spec(5, block_config=cfg), 21)
    model.load_state_dict(sd)
    x = synth.xray_batch(77, 3, 64)
    cam = ref.grad_cam(model, x, {"forward": model.features.norm5, "backward": model.classifier})
    np.savez_compressed(os.path.join(out_dir, "gradcam.npz"), cam=cam.detach().numpy())
    taps = {}
    with torch.no_grad():
        nets.densenet_forward({k: v.clone() for k, v in sd.items()}, x, cfg, train=False, taps=taps)
    mine = gradcam.grad_cam_from_features(torch.relu(taps["norm5"]), 5, x.shape[2:])
    print("[gradcam] oracle-vs-reference: %.2e (cam max %.3f)" % ((mine - cam).abs().max(), cam.max()))
    # the other hook targets of chexpert.py:484 (resnet: layer4 / fc) and :498 (efficientnet: head[1] / head[-1])
    from models.attn_aug_conv import ResNet, Bottleneck
    from models.efficientnet import construct_model
    extra = {}
    layers = (1, 1, 1, 1)
    rn = ResNet(Bottleneck, list(layers), num_classes=5)
    sd = filled_sd(nets.resnet_spec(5, layers=layers), 22)
    rn.load_state_dict(sd)
    x = synth.xray_batch(78, 3, 64)
    cam = ref.grad_cam(rn, x, {"forward": rn.layer4, "backward": rn.fc})
    extra["cam_resnet"] = cam.detach().numpy()
    taps = {}
    with torch.no_grad():
        nets.resnet_forward({k: v.clone() for k, v in sd.items()}, x, layers, train=False, taps=taps)
    mine = gradcam.grad_cam_from_features(taps["layer4"], 5, x.shape[2:])
    print("[gradcam resnet] oracle-vs-reference: %.2e (cam max %.3f)" % ((mine - cam).abs().max(), cam.max()))
    en = construct_model("efficientnet-b0", n_classes=5)
    sd = filled_sd(nets.efficientnet_spec("efficientnet-b0", 5), 23)
    en.load_state_dict(sd)
    x = synth.xray_batch(79, 2, 96)
    cam = ref.grad_cam(en, x, {"forward": en.head[1], "backward": en.head[-1]})
    extra["cam_efficientnet"] = cam.detach().numpy()
    taps = {}
    with torch.no_grad():
        nets.efficientnet_forward({k: v.clone() for k, v in sd.items()}, x, "efficientnet-b0", train=False, taps=taps)
    f = taps["head1"]
    mine = gradcam.grad_cam_from_features(f, 5, x.shape[2:], pooled=(f * torch.sigmoid(f)).mean((2, 3)))
    print("[gradcam efficientnet] oracle-vs-reference: %.2e (cam max %.3f)" % ((mine - cam).abs().max(), cam.max()))
    np.savez_compressed(os.path.join(out_dir, "gradcam_more.npz"), **extra)


def gen_auroc(out_dir):
    import chexpert as ref
    from oracle import metrics
    cases = {}
    rng = [(50, 5, 3), (200, 5, 4), (64, 5, 5), (30, 3, 6)]
    for i, (n, c, seed) in enumerate(rng):
        logits = synth.uniform(seed, (n, c), -3, 3).numpy().astype(np.float64)
        tg = synth.targets(seed + 100, n, c, p=0.35).numpy()
        if i == 1:
            logits = np.round(logits)            # heavy ties
        if i == 2:
            tg[:, 1] = 0                         # one class only -> nan (sklearn warns)
            tg[:, 3] = 1
        if i == 3:
            logits[:, 0] = 0.25                  # all scores tied -> 0.5
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = ref.compute_metrics(torch.from_numpy(logits), torch.from_numpy(tg), torch.zeros(n, c))
        aucs = [float(m["aucs"][k]) for k in range(c)]
        cases["case%d" % i] = dict(n=n, c=c, seed=seed, variant=i, aucs=[None if np.isnan(a) else a for a in aucs],
                                   nanmean=None if np.all(np.isnan(aucs)) else float(np.nanmean(aucs)))
        mine, _ = metrics.per_class_auc(logits, tg)
        print("[auroc case%d] oracle-vs-sklearn: %.2e" % (i, np.nanmax(np.abs(np.array(list(mine.values())) - np.array(aucs)))))
    json.dump(cases, open(os.path.join(out_dir, "auroc.json"), "w"), indent=1)


if __name__ == "__main__":
    assert os.path.isdir(REF), "the reference is only present in the build container"
    install_standins()
    torch.set_num_threads(8)
    which = sys.argv[1:]
    if not which or "aaconv" in which:
        gen_aaconv(HERE)
    if not which or "auroc" in which:
        gen_auroc(HERE)
    if not which or "gradcam" in which:
        gen_gradcam(HERE)
    if not which or "dataset" in which:
        gen_dataset(HERE)
        if which == ["dataset"]:
            sys.exit(0)
    if "smooth" in which:              # python make_golden.py smooth [tag substrings]
        gen_smooth(HERE, [w for w in which if w != "smooth"])
        sys.exit(0)
    if "options" in which:             # python make_golden.py options
        gen_options(HERE)
        sys.exit(0)
    if "yardstick" in which:           # python make_golden.py yardstick <tag substrings>: renew bf16_storage_* of recorded fixtures only
        gen_smooth(HERE, [w for w in which if w != "yardstick"], yard_only=True)
        sys.exit(0)
    net_sel = [w for w in which if w not in ("aaconv", "auroc", "gradcam", "nets", "dataset")]
    if not which or "nets" in which or net_sel:
        gen_nets(HERE, net_sel)
    if not which:
        gen_smooth(HERE, [])
