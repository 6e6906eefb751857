"""GPU: the bf16 HIP schedule against WELL-CONDITIONED fixtures recorded from the REAL reference (tests/golden/nets_smooth.json,
written by tests/golden/make_golden.py `smooth`): BatchNorm gains in [0.8, 1.2], biases 2.5 / 1.0, kaiming-scale convolutions,
B = 8 -- the regime in which north_star's 1e-2 (relative to the logit abs-max) is a statement about the kernels and not about
what bf16 storage does to a chaotic fixture.  Every bound below is a literal.

Two families of checks:
  * one training step (chexpert.py:159-163) at the fixture's own batch: train logits, loss, every parameter-gradient norm, running
    statistics;
  * the same step at the BASELINE batch geometry (256 / 128 / 64 images: other grids, split counts, 32-bit offsets against GB-sized
    buffers): the batch is the fixture's 8 images repeated, which has the SAME batch statistics, so every copy must reproduce the
    golden logits and the loss / gradients must equal the golden ones (mean over the batch).
"""
import json
import os

import pytest
import torch

from chexpert_amd import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
ATTN = {"k": 0.2, "v": 0.1, "nh": 8, "relative": True, "input_dims": (320, 320)}      # chexpert.py:476


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from chexpert_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def golden():
    return json.load(open(os.path.join(G, "nets_smooth.json")))


def _rel(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)


def _make(tag, n_cls):
    """(model, state_dict) of a fixture tag: the drop-in constructors of the reference (chexpert.py:461-500)."""
    from chexpert_amd.models import Bottleneck, DenseNet, ResNet, construct_model
    from chexpert_amd.models.efficientnet import DropMarker
    from oracle import nets
    attn = dict(k=.2, v=.1, nh=8)
    if tag.startswith(("densenetbc", "aadensenetbc")):     # the CIFAR harness's Densenet-BC (models/test_model.py:304-306), growth 12
        n = int(tag.split("_")[2][1:])
        n = (n - 4) // 6
        aa = tag.startswith("aa")                          # --attn with the harness defaults (k 0.2, v 0.1, 8 heads, 32x32 input)
        vv = 0.7 if "v07" in tag else 0.1                  # ... or the v = 0.7 of the reference's result rows (models/readme.md:34-38)
        spec = nets.densenet_spec(n_cls, growth=12, block_config=(n, n, n), init_features=24, attn=dict(attn, v=vv) if aa else None,
                                  input_hw=(32, 32))
        model = DenseNet(12, (n, n, n), 24, num_classes=n_cls, attn_params=dict(ATTN, input_dims=(32, 32), v=vv) if aa else None)
        bias = 2.5
    elif tag.startswith("densenet121"):
        spec, model, bias = nets.densenet_spec(n_cls), DenseNet(32, (6, 12, 24, 16), 64, num_classes=n_cls), 2.5
    elif tag.startswith("aadensenet121"):
        spec, model, bias = nets.densenet_spec(n_cls, attn=attn), DenseNet(32, (6, 12, 24, 16), 64, num_classes=n_cls, attn_params=dict(ATTN)), 2.5
    elif tag.startswith("resnet152"):
        spec, model, bias = nets.resnet_spec(n_cls), ResNet(Bottleneck, [3, 8, 36, 3], num_classes=n_cls), 1.0
    elif tag.startswith("aaresnet152"):
        spec, model, bias = nets.resnet_spec(n_cls, attn=attn), ResNet(Bottleneck, [3, 8, 36, 3], num_classes=n_cls, attn_params=dict(ATTN)), 1.0
    else:
        name = tag.split("_")[0]
        spec, model, bias = nets.efficientnet_spec(name, n_cls), construct_model(name, n_cls), 1.0
        for mod in model.modules():                 # deterministic part, as recorded (make_golden.py sets p = 0)
            if isinstance(mod, DropMarker):
                mod.p = 0.0
    sd = synth.smooth_state_dict_(synth.fill_state_dict_(nets.zeros_state_dict(spec), 21), bias)
    assert list(model.state_dict().keys()) == list(spec.keys())
    model.load_state_dict(sd, strict=True)
    return model, sd


def _sampled(f, n):
    """The 16 elements of a flattened gradient that the fixture records (make_golden.py `summarise`: first 8 + 8 strided)."""
    idx = (torch.arange(8) * max(1, n // 8) + (n // 16)).clamp(max=n - 1)
    return torch.cat([f[:8], f[idx.to(f.device)]]).double().cpu()


def _direction(named_grads, rec):
    """Element-level agreement with the reference gradients: for every parameter the recorded elements (`head`, `samples`) against
    the golden values in units of the tensor's RMS (l2 / sqrt(n), as tests/test_oracle_golden.py does), and the cosine over all
    recorded elements of all tensors, each tensor scaled to unit RMS.  A norm cannot see a permuted, transposed or sign-flipped
    tile; these can.  Third figure (round 5): the RMS of a tensor's 16 element errors -- on the 152-layer ResNet single elements of
    the noisiest tensors are 1.4 RMS off in bf16, as far as a transposed tile moves an element, so the worst ELEMENT cannot tell the
    two apart; a transposed tile moves 7 of the 16 recorded elements at once (RMS error ~0.95 against <= 0.55 of bf16 noise).
    Returns (worst [(err, name)], cosine, worst [(rms err, name)])."""
    gmax = max(r["l2"] for r in rec["grads"].values())
    errs, got_all, want_all, rmss = [], [], [], []
    for k, g in named_grads:
        r = rec["grads"][k]
        if r["l2"] < 1e-3 * gmax:
            continue
        n = g.numel()
        scale = r["l2"] / max(1.0, n ** 0.5)
        got = _sampled(g.detach().flatten(), n) / scale
        want = torch.tensor(r["head"] + r["samples"], dtype=torch.float64) / scale
        errs.append((float((got - want).abs().max()), k))
        rmss.append((float(((got - want) ** 2).mean().sqrt()), k))
        got_all.append(got)
        want_all.append(want)
    errs.sort(reverse=True)
    rmss.sort(reverse=True)
    a, b = torch.cat(got_all), torch.cat(want_all)
    return errs, float((a * b).sum() / (a.norm() * b.norm())), rmss


def _check_step(tag, rec, model, dev, copies, lim_logits, lim_loss, lim_norm, lim_norm_1d):
    n_cls = rec["n_classes"]
    x8 = synth.xray_batch(rec["x_seed"], rec["B"], rec["S"])
    t8 = synth.targets(rec["t_seed"], rec["B"], n_cls)
    x, t = x8.repeat(copies, 1, 1, 1).to(dev), t8.repeat(copies, 1).to(dev)
    model.train()
    model.zero_grad()
    loss, logits = model.forward_backward(x, t)
    want = torch.tensor(rec["logits_train"])
    lg = logits.cpu().view(copies, rec["B"], n_cls)
    e = max(_rel(lg[i], want) for i in range(copies))
    e_loss = abs(loss.item() - rec["loss"]) / abs(rec["loss"])
    gmax = max(r["l2"] for r in rec["grads"].values())
    worst = []
    for k, p in model.named_parameters():
        r = rec["grads"][k]
        assert p.grad is not None and torch.isfinite(p.grad).all().item(), k
        if r["l2"] < 1e-3 * gmax:
            continue
        worst.append((abs(p.grad.double().norm().item() / r["l2"] - 1.0), k))
    worst.sort(reverse=True)
    w_nd = [w for w in worst if dict(model.named_parameters())[w[1]].dim() > 1][:3]
    w_1d = [w for w in worst if dict(model.named_parameters())[w[1]].dim() == 1][:3]
    print("%s x%d: train logits rel %.3e, loss rel %.3e, worst grad-norm deviation weights %s, norm parameters %s"
          % (tag, copies, e, e_loss, [(round(a, 4), b) for a, b in w_nd], [(round(a, 4), b) for a, b in w_1d]))
    errs, cos, rmss = _direction([(k, p.grad) for k, p in model.named_parameters()], rec)
    lim_dir, lim_cos, lim_rms = DIRECTION[tag]
    print("%s x%d: recorded gradient elements: worst deviation %s of the tensor RMS, cosine over all of them %.5f, worst RMS error of a tensor's 16 elements %s"
          % (tag, copies, [(round(a, 3), b) for a, b in errs[:3]], cos, [(round(a, 3), b) for a, b in rmss[:2]]))
    assert cos > lim_cos, "gradient direction: cosine %.4f over the recorded elements" % cos
    assert errs[0][0] < lim_dir, errs[:5]
    assert rmss[0][0] < lim_rms, rmss[:5]
    assert e < lim_logits, "train logits %.3e of the abs-max" % e
    assert e_loss < lim_loss
    assert not w_nd or w_nd[0][0] < lim_norm, w_nd
    assert not w_1d or w_1d[0][0] < lim_norm_1d, w_1d
    return model


# tag -> (logits, loss, weight-gradient norm, norm-parameter gradient norm) literal limits
# The two DenseNets and EfficientNet-b0 meet north_star's 1e-2 with room (1.2e-3, 2.8e-3, 4.0e-3 measured).  resnet152 meets it since
# round 4 (9.2e-3 at the fixture batch, 6.9e-3 at 128 images; deterministic, so numbers, not spreads): its residual stream keeps 16
# significant bits through the 42 joins of layer2 / layer3 (bf16 hi plane + int8 lo plane, csrc/common.h cx_join2; rounded to bf16 at
# every join the stream alone was 1.0e-2 away, the whole path 1.12e-2 with a 1.3e-2 bound).  Where a limit is still wider the fixture
# is the reason: aaresnet152's 47 softmax layers make it ill-conditioned -- the fp32 oracle with NOTHING but the conv weights
# rounded to bf16 is 1.3e-2 away from the reference, with only the stored conv outputs rounded 4.7e-2, with only the residual stream
# rounded 2.4e-2 (scratch/aares_storage_model.py; `bf16_storage_logits_rel` in the fixture: 5.1e-2 for everything the path stores);
# the HIP path measures 5.9e-2, the out-projection gradients 16 %.  Its 1e-3 answer is the fp32 storage mode (test_fp32_gpu.py: 7.5e-6).
CASES = {
    "densenet121_320_b8": (1e-2, 1e-2, 0.05, 0.05),
    "aadensenet121_320_b8": (1e-2, 1e-2, 0.05, 0.05),
    # resnet152: north_star's 1e-2 on logits and loss, 5 % on the weight-gradient norms; the norm-parameter entry is the stem's bn1.weight
    # alone (every other norm parameter is within 1e-4): its gradient is a sum over the 160x160 map x batch with heavy cancellation, and the
    # fixture's own `bf16_storage_yardstick` -- the fp32 oracle with nothing but the stored tensors rounded to bf16 -- puts it 9.2 % off.
    # The HIP path measured 3.4 % with the tiled 1x1 kernel's statistic rows and 5.9 % with the activation-stationary kernel's (same
    # products, bit-identical outputs, another order of the fp32 row sums): a draw inside what storage alone does.  "yard": 1 x that figure.
    "resnet152_320_b8": (1e-2, 1e-2, 0.05, "yard"),
    # aaresnet152 (not a BASELINE configuration): an ill-conditioned fixture, held to a multiple of what bf16 STORAGE alone does to it --
    # None: the limits are YARD x the fixture's `bf16_storage_yardstick` (the fp32 oracle with nothing but the stored tensors rounded
    # to bf16 against the reference: logits 5.1e-2, loss 6.0e-3, weight-gradient norms 9.9 %, norm-parameter norms 9.7 %; recorded by
    # `make_golden.py yardstick`).  Not a hand-set number: round 4 widened hand-set bounds when the measurement moved 4.7e-2 -> 5.9e-2;
    # the bisection asked for (profiles/r05_bisect_aares.txt) shows no switch owns that move -- single-plane stream 4.5e-2, default
    # 4.9e-2, separate joins 5.0e-2, lo plane on EVERY join (strictly more precise) 6.2e-2 / 5.3e-2: the fixture amplifies any change of
    # rounding or summation order by +-1e-2.  The statement about the kernels is the fp32 mode's 7.5e-6 (test_fp32_gpu.py).
    "aaresnet152_320_b8": None,
    # EfficientNets: logits 4.5e-3 / 7.4e-3 (deterministic engine: the same at every batch geometry).  Gradient norms agree to 5 % except
    # the squeeze-excite reduce convolutions (blocks.*.6.1 / .3.1: 7.6 % on b0, 10.2 % on b4): ds = sum_hw du * swish(bn(y)) is a sum
    # with heavy cancellation over bf16-rounded du -- the same tensors are 1e-5 from the reference in the fp32 mode (test_fp32_gpu.py)
    # Densenet-BC k = 12 of the CIFAR harness on the channel-padded twin (models/densenet.py _PaddedEngine): widths 24 + 12 i
    "densenetbc_k12_L40_32_b8": (1e-2, 1e-2, 0.05, 0.05),
    "densenetbc_k12_L100_32_b8": (1e-2, 1e-2, 0.05, 0.05),
    "aadensenetbc_k12_L100_32_b8": (1e-2, 1e-2, 0.05, 0.05),      # ... with attention-augmented transitions (--attn defaults)
    # ... at v = 0.7 (heads of 9 / 13 value channels; 72 / 104 of the 108 / 150 transition channels are attention output): logits
    # 3.4e-3 -- bf16 storage alone moves this fixture 3.9e-3 (`bf16_storage_logits_rel`, three times the v = 0.1 fixture) -- weight
    # gradients 3.6 %, the BatchNorm gains of block 1's first layers 6.1 %
    "aadensenetbcv07_k12_L100_32_b8": (1e-2, 1e-2, 0.06, 0.08),
    "efficientnet-b0_224_b8": (1e-2, 1e-2, 0.12, 0.06),
    "efficientnet-b4_380_b8": (1e-2, 1e-2, 0.12, 0.06),
}


# tag -> (worst deviation of a recorded gradient element in units of its tensor's RMS, cosine over all recorded elements, worst RMS
# error of one tensor's 16 recorded elements); set
# from the measured values (printed by _check_step), see test_direction_check_catches_a_transposed_tile for what they catch
DIRECTION = {       # measured (x1 and at the BASELINE batch)            worst element      cosine
    "densenet121_320_b8": (0.6, 0.995, 0.25),  # rms 0.173 / 0.157; 0.40 / 0.36      0.9978
    "aadensenet121_320_b8": (2.2, 0.975, 0.48),  # rms 0.280 / 0.344; 0.79 / 1.55      0.9863  (in_proj_qkv of transition1: its output gradient passes the attention backward in bf16)
    "resnet152_320_b8": (1.9, 0.94, 0.65),  # rms 0.502 / 0.470; 1.51 / 1.10      0.9565  (152 layers of bf16 operands: the norms agree to 1.4 %, single elements to ~1 RMS)
    "aaresnet152_320_b8": (3.2, 0.85, 1.1),  # rms 0.773; 2.24             0.918   (the ill-conditioned fixture, see CASES)
    "densenetbc_k12_L40_32_b8": (1.1, 0.99, 0.36),  # rms 0.250; 0.73             0.9947
    "densenetbc_k12_L100_32_b8": (0.9, 0.99, 0.48),  # rms 0.339; 0.58             0.9947
    "aadensenetbc_k12_L100_32_b8": (0.95, 0.98, 0.35),  # rms 0.242; 0.63             0.9903
    "aadensenetbcv07_k12_L100_32_b8": (0.9, 0.98, 0.33),  # rms 0.227; 0.59             0.9916
    "efficientnet-b0_224_b8": (0.6, 0.997, 0.17),  # rms 0.111; 0.39             0.9992
    "efficientnet-b4_380_b8": (0.7, 0.995, 0.28),  # rms 0.195 / 0.149; 0.46             0.9981
}


YARD = 2.0      # the HIP path also rounds the MFMA operands, which the storage yardstick does not model


def _limits(tag, rec):
    if CASES[tag] is not None:
        c = CASES[tag]
        if "yard" in c:
            y = rec["bf16_storage_yardstick"]
            ys = (y["logits"], y["loss"], y["weight_grad_norm"], y["norm_grad_norm"])
            c = tuple(ys[i] if v == "yard" else v for i, v in enumerate(c))
        return c
    y = rec["bf16_storage_yardstick"]
    return (YARD * y["logits"], YARD * y["loss"], YARD * y["weight_grad_norm"], YARD * y["norm_grad_norm"])


@pytest.mark.parametrize("tag", list(CASES))
def test_train_step_matches_reference_smooth_fixture(dev, golden, tag):
    rec = golden[tag]
    model, sd = _make(tag, rec["n_classes"])
    assert sum(p.numel() for p in model.parameters()) == rec["n_params"]
    model = model.to(dev)
    _check_step(tag, rec, model, dev, 1, *_limits(tag, rec))
    after = model.state_dict()
    for k, r in rec["running"].items():                          # BatchNorm running statistics after the step
        f = after[k].detach().double().flatten().cpu()
        assert abs(float(f.norm()) - r["l2"]) <= 1e-2 * r["l2"], (k, float(f.norm()), r["l2"])
        assert (f[:8] - torch.tensor(r["head"], dtype=torch.float64)).abs().max().item() <= 1e-2 * r["l2"] / max(1.0, r["n"] ** 0.5) + 1e-3, k


# BASELINE.json configs[1..4]: per-GPU batches 256 / 128 / 128 / 64
@pytest.mark.parametrize("tag,copies", [("densenet121_320_b8", 32), ("aadensenet121_320_b8", 16), ("resnet152_320_b8", 16),
                                        ("efficientnet-b4_380_b8", 8)])
def test_baseline_batch_geometry_reproduces_the_fixture(dev, golden, tag, copies):
    rec = golden[tag]
    model, sd = _make(tag, rec["n_classes"])
    _check_step(tag, rec, model.to(dev), dev, copies, *_limits(tag, rec))


def test_direction_check_catches_a_transposed_tile(dev, golden):
    """Negative control at the BASELINE batch (densenet121, 256 images): the element-level check passes on the gradients the HIP path
    computes and FAILS when one 8 x 8 tile of ONE weight gradient is transposed (norms are blind to that), or when one tensor's sign
    is flipped."""
    tag = "densenet121_320_b8"
    rec = golden[tag]
    model, _ = _make(tag, rec["n_classes"])
    model = model.to(dev)
    _check_step(tag, rec, model, dev, 32, *_limits(tag, rec))
    lim_dir, lim_cos, lim_rms = DIRECTION[tag]
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    errs, cos, rmss = _direction(list(grads.items()), rec)
    assert errs[0][0] < lim_dir and cos > lim_cos and rmss[0][0] < lim_rms
    name = "features.denseblock3.denselayer7.conv1.weight"              # (128, 448, 1, 1): the fused 1x1 backward's weight gradient
    w = grads[name]
    tile = w[:8, :8, 0, 0].clone()
    bad = dict(grads)
    bad[name] = w.clone()
    bad[name][:8, :8, 0, 0] = tile.t()
    assert abs(float(bad[name].norm() / w.norm()) - 1.0) < 1e-12         # the norm check cannot see it
    errs_t, _, rms_t = _direction(list(bad.items()), rec)
    assert errs_t[0][1] == name and errs_t[0][0] > lim_dir, errs_t[:3]
    assert rms_t[0][1] == name and rms_t[0][0] > lim_rms, rms_t[:3]
    bad = dict(grads)
    bad["features.denseblock2.denselayer3.conv2.weight"] = -grads["features.denseblock2.denselayer3.conv2.weight"]
    errs_s, cos_s, rms_s = _direction(list(bad.items()), rec)
    assert errs_s[0][0] > lim_dir and rms_s[0][0] > lim_rms


def test_direction_check_catches_a_transposed_tile_in_resnet152(dev, golden):
    """The negative control on the kernels ResNet152 runs on (conv_mm / wgrad_mm / wgrad3), at the BASELINE batch (128 images = the
    fixture's 8 x 16): the element-level check passes on the HIP gradients and FAILS on what a fragment-layout bug in a weight-gradient
    kernel produces -- neighbouring output channels of a 1x1 weight gradient swapped (wgrad_mm), the taps of a 3x3 weight gradient mirrored
    (wgrad3) -- and on a sign-flipped tensor; the norms are blind to all three.  (ONE transposed tile moves 7 of the 16 recorded
    elements by about one RMS: 0.54 RMS error of the tensor against 0.47 for the noisiest clean tensor of this 152-layer network in
    bf16 -- not resolvable here; on DenseNet121, whose noise is 0.16, the single tile is the control above.)"""
    tag = "resnet152_320_b8"
    rec = golden[tag]
    model, _ = _make(tag, rec["n_classes"])
    model = model.to(dev)
    _check_step(tag, rec, model, dev, 16, *_limits(tag, rec))
    lim_dir, lim_cos, lim_rms = DIRECTION[tag]
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    errs, cos, rmss = _direction(list(grads.items()), rec)
    assert errs[0][0] < lim_dir and cos > lim_cos and rmss[0][0] < lim_rms
    for name in ("layer3.10.conv1.weight", "layer2.3.conv2.weight"):     # (256, 1024, 1, 1) and (128, 128, 3, 3)
        w = grads[name]
        bad = dict(grads)
        bad[name] = w.clone()
        if w.shape[-1] == 1:                                             # output channels 2k <-> 2k + 1 (a lane-pairing slip)
            O_, I_ = w.shape[:2]
            bad[name] = w.view(O_ // 2, 2, I_, 1, 1).flip(1).reshape(w.shape).contiguous()
        else:                                                            # kx mirrored
            bad[name] = w.flip(-1).contiguous()
        assert abs(float(bad[name].norm() / w.norm()) - 1.0) < 1e-6      # the norm check cannot see it
        errs_t, _, rms_t = _direction(list(bad.items()), rec)
        print("%s corrupted: RMS error of its recorded elements %s (limit %.2f; clean worst %.3f); worst single element %.3f (clean %.3f: bf16 noise reaches that)"
              % (name, [(round(a, 3), b) for a, b in rms_t[:2]], lim_rms, rmss[0][0], errs_t[0][0], errs[0][0]))
        assert rms_t[0][1] == name and rms_t[0][0] > lim_rms, rms_t[:3]
    bad = dict(grads)
    bad["layer3.20.conv3.weight"] = -grads["layer3.20.conv3.weight"]
    errs_s, cos_s, rms_s = _direction(list(bad.items()), rec)
    print("sign flip: worst element %s, worst tensor RMS error %s, cosine %.4f" % ([(round(a, 3), b) for a, b in errs_s[:1]], [(round(a, 3), b) for a, b in rms_s[:1]], cos_s))
    assert rms_s[0][1] == "layer3.20.conv3.weight" and rms_s[0][0] > lim_rms and errs_s[0][0] > lim_dir


@pytest.mark.parametrize("tag", ["densenetbc_k12_L40_32_b8", "densenetbc_k12_L100_32_b8", "aadensenetbc_k12_L100_32_b8",
                                 "aadensenetbcv07_k12_L100_32_b8"])
def test_densenet_bc_eval_and_second_step(dev, golden, tag):
    """Channel-padded twin of the CIFAR Densenet-BC: eval-mode logits against the reference (running statistics mapped real ->
    padded), two training steps in a row give the same gradients (the twin's gradient buffer is rebuilt, the real one accumulates
    only what zero_grad left), the padded channels stay exactly zero, and state_dict round-trips into a fresh model."""
    from chexpert_amd.models import DenseNet
    rec = golden[tag]
    n_cls = rec["n_classes"]
    model, sd = _make(tag, n_cls)
    model = model.to(dev)
    x = synth.xray_batch(rec["x_seed"], rec["B"], rec["S"]).to(dev)
    t = synth.targets(rec["t_seed"], rec["B"], n_cls).to(dev)
    model.eval()
    with torch.no_grad():
        le = model(x).cpu()
    e = _rel(le, torch.tensor(rec["logits_eval"]))
    print("%s eval logits rel %.3e" % (tag, e))
    assert e < 1e-2
    model.train()
    model.zero_grad()
    model.forward_backward(x, t)
    g1 = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    model.load_state_dict(sd, strict=True)                   # undo the running-statistics update
    model.zero_grad()
    model.forward_backward(x, t)
    for k, p in model.named_parameters():
        assert torch.equal(p.grad, g1[k]), k
    eng = model._eng()
    twin = eng.twin
    kp, k = twin.growth_rate, model.growth_rate
    w = twin.features.denseblock1.denselayer2.conv2.weight       # rows k..kp of a padded 3x3 convolution: never written
    assert (w[k:] == 0).all() and (w[:k] != 0).any()
    assert (eng.inner.flat_grad[eng.inner.off_of[id(w)]:][:w.numel()].view(w.shape)[k:] == 0).all()
    m2 = DenseNet(k, model.block_config, 2 * k, num_classes=n_cls,
                  attn_params=dict(ATTN, input_dims=(32, 32), v=0.7 if "v07" in tag else 0.1) if tag.startswith("aa") else None).to(dev)
    m2.load_state_dict(model.state_dict(), strict=True)
    if tag.startswith("aa"):                                   # AAConv2d.weights of the REAL modules (attn_aug_conv.py:87; --vis_attn)
        w = model.features.transition1.conv.weights
        assert w is not None and tuple(w.shape) == (rec["B"], 8, 256, 256) and abs(w[0, 0].sum(-1) - 1).max().item() < 1e-3
    m2.eval()
    model.eval()
    with torch.no_grad():
        assert torch.equal(m2(x), model(x))

