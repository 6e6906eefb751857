"""Command line of the MI355X path, keeping the flag names and defaults of /root/reference/chexpert.py:29-57
(`--train`, `--evaluate_single_model`, `--evaluate_ensemble`, `--visualize`, `--model`, `--batch_size 16`, `--lr 1e-4`,
`--n_epochs 1`, `--log_interval 50`, `--eval_interval 300`, `--lr_decay_factor 0.97`, `--resize`, `--mini_data`, `--cuda`,
`--restore`, `--load_config`, `--seed`), plus `--evaluate` (alias of --evaluate_single_model), `--synthetic N` (the
CheXpert images are not available offline: N hash-generated X-rays with U-Ones-like labels), `--n_classes` and
`--fused_optimizer`.  Logging goes to stdout / JSON (tensorboardX is not used).
"""
import argparse
import json
import os
import pprint
import time

import numpy as np
import torch
import torch.nn as nn

from . import metrics as M
from . import synth

ATTR_NAMES = ["Atelectasis", "Cardiomegaly", "Consolidation", "Edema", "Pleural Effusion"]     # dataset.py:25


def build_parser():
    p = argparse.ArgumentParser(description="CheXpert classifiers on MI355X (HIP kernels)")
    p.add_argument("--load_config", type=str)
    p.add_argument("--train", action="store_true")
    p.add_argument("--evaluate_single_model", "--evaluate", dest="evaluate_single_model", action="store_true")
    p.add_argument("--evaluate_ensemble", action="store_true")
    p.add_argument("--visualize", action="store_true")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--cuda", type=int, default=0)
    p.add_argument("--data_path", default="")
    p.add_argument("--output_dir")
    p.add_argument("--restore", type=str)
    p.add_argument("--model", default="densenet121")
    p.add_argument("--mini_data", type=int)
    p.add_argument("--resize", type=int)
    p.add_argument("--pretrained", action="store_true")
    p.add_argument("--batch_size", type=int, default=16)
    p.add_argument("--n_epochs", type=int, default=1)
    p.add_argument("--lr", type=float, default=1e-4)
    p.add_argument("--lr_warmup_steps", type=float, default=0)
    p.add_argument("--lr_decay_factor", type=float, default=0.97)
    p.add_argument("--step", type=int, default=0)
    p.add_argument("--log_interval", type=int, default=50)
    p.add_argument("--eval_interval", type=int, default=300)
    p.add_argument("--synthetic", type=int, default=0, help="number of synthetic training images (no dataset offline)")
    p.add_argument("--n_classes", type=int, default=len(ATTR_NAMES))
    p.add_argument("--fused_optimizer", action="store_true", help="one-kernel optimiser on the flat parameter buffer")
    return p


class SyntheticXrays(torch.utils.data.Dataset):
    """uint8 U{0..255} images through the reference transform chain (chexpert.py:70-72); Bernoulli(0.3) labels."""

    def __init__(self, n, size, n_classes, seed):
        self.n, self.size, self.n_classes, self.seed = n, size, n_classes, seed
        self.targets = synth.targets(seed + 1, n, n_classes)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return synth.xray_batch(self.seed * 1000003 + i, 1, self.size)[0], self.targets[i], i


def make_model(args, device):
    """Model zoo of chexpert.py:461-502 (the families built so far)."""
    from .models import densenet121
    name = args.model
    if name == "densenet121":
        model = densenet121(pretrained=args.pretrained)
        model.classifier = nn.Linear(model.classifier.in_features, args.n_classes)
        nn.init.constant_(model.classifier.bias, 0)
        model = model.to(device)
        if args.fused_optimizer:
            from .optim import FusedAdam
            opt = FusedAdam(model, lr=args.lr)
        else:
            opt = torch.optim.Adam(model.parameters(), lr=args.lr)
        return model, opt, None
    if name in ("aadensenet121", "densenet121_attn_aug"):      # chexpert.py:474-480 (README row name accepted too)
        from .models import DenseNet
        size = args.resize or 320
        model = DenseNet(32, (6, 12, 24, 16), 64, num_classes=args.n_classes,
                         attn_params={"k": 0.2, "v": 0.1, "nh": 8, "relative": True, "input_dims": (size, size)}).to(device)
        opt = torch.optim.SGD(model.parameters(), lr=args.lr, momentum=0.9, nesterov=True)
        return model, opt, torch.optim.lr_scheduler.MultiStepLR(opt, [40000, 60000])
    if name == "resnet152":                                   # chexpert.py:481-486
        from .models import resnet152
        model = resnet152(pretrained=args.pretrained)
        model.fc = nn.Linear(model.fc.in_features, args.n_classes)
        model = model.to(device)
        return model, torch.optim.Adam(model.parameters(), lr=args.lr), None
    if "efficientnet" in name:                                # chexpert.py:496-500
        from .models import construct_model
        model = construct_model(name, n_classes=args.n_classes).to(device)
        opt = torch.optim.RMSprop(model.parameters(), lr=args.lr, momentum=0.9, eps=0.001)
        return model, opt, torch.optim.lr_scheduler.ExponentialLR(opt, args.lr_decay_factor)
    if name == "aaresnet152":                                 # chexpert.py:486-494
        from .models import Bottleneck, ResNet
        size = args.resize or 320
        model = ResNet(Bottleneck, [3, 8, 36, 3], num_classes=args.n_classes,
                       attn_params={"k": 0.2, "v": 0.1, "nh": 8, "relative": True, "input_dims": (size, size)}).to(device)
        return model, torch.optim.Adam(model.parameters(), lr=args.lr), None
    raise RuntimeError("Model architecture not supported.")


@torch.no_grad()
def evaluate(model, loader, device):
    model.eval()
    outs, tgts, losses = [], [], []
    loss_fn = nn.BCEWithLogitsLoss(reduction="none")
    for x, t, _ in loader:
        o = model(x.to(device))
        losses.append(loss_fn(o, t.to(device)).cpu())
        outs.append(o.cpu())
        tgts.append(t)
    return torch.cat(outs), torch.cat(tgts), torch.cat(losses)


def save_checkpoint(ckpt, optim_state, sched_state, args, max_records=10):
    """Latest + the `max_records` best checkpoints by mean AUROC with a tracker file
    (behaviour of chexpert.py:90-123: evict the lowest-AUROC record and re-use its file id)."""
    d = args.output_dir
    os.makedirs(os.path.join(d, "best_checkpoints"), exist_ok=True)
    torch.save(ckpt, os.path.join(d, "checkpoint_latest.pt"))
    torch.save(optim_state, os.path.join(d, "optim_checkpoint_latest.pt"))
    if sched_state:
        torch.save(sched_state, os.path.join(d, "sched_checkpoint_latest.pt"))
    path = os.path.join(d, "checkpoints_tracker.csv")
    recs = []
    if os.path.exists(path):
        recs = [list(r) for r in np.atleast_2d(np.loadtxt(path, skiprows=1))]
    file_id, floor = len(recs), float("-inf")
    if len(recs) == max_records:
        worst = min(range(len(recs)), key=lambda i: recs[i][3])
        floor, file_id = recs[worst][3], int(recs[worst][0])
        recs.pop(worst)
    recs.append([file_id, args.step, float(ckpt["eval_loss"]), float(ckpt["avg_auc"])])
    recs.sort(key=lambda r: -r[3])
    if ckpt["avg_auc"] > floor:
        np.savetxt(path, np.array(recs), delimiter=" ", header="CheckpointId Step Loss AvgAUC")
        torch.save(ckpt, os.path.join(d, "best_checkpoints", "checkpoint_%d.pt" % file_id))


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.load_config:
        args.__dict__.update(json.load(open(args.load_config)))
    if not args.output_dir:
        if args.restore:
            raise RuntimeError("Must specify `output_dir` argument")
        args.output_dir = os.path.join("results", time.strftime("%Y-%m-%d_%H-%M-%S", time.gmtime()))
    os.makedirs(args.output_dir, exist_ok=True)
    cfg_path = os.path.join(args.output_dir, "config.json")
    if not os.path.exists(cfg_path):
        json.dump(args.__dict__, open(cfg_path, "w"), indent=4)
    if not torch.cuda.is_available():
        raise RuntimeError("chexpert_amd needs an MI355X (no CPU fallback)")
    device = torch.device("cuda:%d" % (args.cuda or 0))
    if args.seed:
        torch.manual_seed(args.seed)
        np.random.seed(args.seed)
    model, optimizer, scheduler = make_model(args, device)
    if args.restore and os.path.isfile(args.restore):
        ck = torch.load(args.restore, map_location=device)
        model.load_state_dict(ck["state_dict"])
        args.step = ck["global_step"]
    size = args.resize or 320
    if not args.synthetic:
        raise RuntimeError("the CheXpert-small dataset is not available offline; pass --synthetic N")
    n_valid = max(args.batch_size, args.synthetic // 5)
    train = torch.utils.data.DataLoader(SyntheticXrays(args.mini_data or args.synthetic, size, args.n_classes, 7),
                                        args.batch_size, shuffle=True, drop_last=True)
    valid = torch.utils.data.DataLoader(SyntheticXrays(n_valid, size, args.n_classes, 11), args.batch_size)
    loss_fn = nn.BCEWithLogitsLoss(reduction="none")
    print("Loaded %s (number of parameters: %s; weights trained to step %d)" % (
        model._get_name(), format(sum(p.numel() for p in model.parameters()), ","), args.step))

    def run_eval(tag):
        res = M.compute_metrics(*evaluate(model, valid, device))
        print("Evaluate metrics @ step %d:\nAUC:\n%s\nLoss:\n%s" % (args.step, pprint.pformat(res["aucs"]), pprint.pformat(res["loss"])))
        json.dump(res, open(os.path.join(args.output_dir, tag + ".json"), "w"), indent=4)
        return res

    if args.train:
        for epoch in range(args.n_epochs):
            model.train()
            for x, t, _ in train:
                args.step += 1
                out = model(x.to(device))
                loss = loss_fn(out, t.to(device)).sum(1).mean(0)          # chexpert.py:160
                optimizer.zero_grad()
                loss.backward()
                optimizer.step()
                if scheduler and args.step >= args.lr_warmup_steps:
                    scheduler.step()
                if args.step % args.log_interval == 0:
                    print(json.dumps({"step": args.step, "train_loss": round(loss.item(), 5)}), flush=True)
                if args.step % args.eval_interval == 0:
                    res = M.compute_metrics(*evaluate(model, valid, device))
                    save_checkpoint({"global_step": args.step, "eval_loss": float(np.sum(list(res["loss"].values()))),
                                     "avg_auc": M.mean_auc(res), "state_dict": model.state_dict()},
                                    optimizer.state_dict() if hasattr(optimizer, "state_dict") else {}, None, args)
                    model.train()
            run_eval("eval_results_step_%d" % args.step)
    if args.evaluate_single_model:
        run_eval("eval_results_step_%d" % args.step)
    if args.evaluate_ensemble:
        assert args.restore and os.path.isdir(args.restore), "Restore argument must be directory with saved checkpoints"
        outs, losses = [], []
        for c in sorted(f for f in os.listdir(args.restore) if f.startswith("checkpoint") and f.endswith(".pt")):
            model.load_state_dict(torch.load(os.path.join(args.restore, c), map_location=device)["state_dict"])
            o, tg, l = evaluate(model, valid, device)
            outs.append(o)
            losses.append(l)
        res = M.compute_metrics(torch.stack(outs, 2).mean(2), tg, torch.stack(losses, 2).mean(2))   # mean of logits, chexpert.py:233
        json.dump(res, open(os.path.join(args.output_dir, "eval_results_ensemble.json"), "w"), indent=4)
        print("AUC:\n", pprint.pformat(res["aucs"]))
    if args.visualize:
        from .gradcam import grad_cam
        x, _, _ = next(iter(valid))
        cam = grad_cam(model, x.to(device))
        os.makedirs(os.path.join(args.output_dir, "vis"), exist_ok=True)
        np.save(os.path.join(args.output_dir, "vis", "grad_cam.npy"), cam.cpu().numpy())
        print("grad-cam maps:", tuple(cam.shape))


if __name__ == "__main__":
    main()
