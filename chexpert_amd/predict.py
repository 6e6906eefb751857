"""Prediction over a csv of image paths, as the reference's predict.py: probabilities of the five competition findings per study
(max over a study's views), from one checkpoint or the mean over the checkpoints of a folder, written as csv.  The forward pass is
the eval-mode HIP path of chexpert_amd.models; reading / cropping is chexpert_amd.data (mode 'test').

  python predict.py <data.csv> <predictions.csv> --restore_path <checkpoint.pt | folder> [--model densenet121|resnet152]
                    [--batch_size 16] [--resize N] [--mini_data N] [--cuda 0]
"""
import argparse
import os

import numpy as np
import torch
import torch.nn as nn


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("data_path", type=str, help="csv with a Path column")
    p.add_argument("output_path", type=str, help="csv to write")
    p.add_argument("--restore_path", type=str, required=True, help="one checkpoint, or a folder of checkpoint*.pt to ensemble")
    p.add_argument("--model", default="densenet121", choices=["densenet121", "resnet152"])
    p.add_argument("--cuda", type=int, default=0)
    p.add_argument("--batch_size", type=int, default=16)
    p.add_argument("--resize", type=int)
    p.add_argument("--mini_data", type=int)
    return p


@torch.no_grad()
def predict(model, dataset, batch_size, device):
    """DataFrame indexed by study ('.../patient64541/study1') with one probability column per finding (predict.py:33-52)."""
    import pandas as pd
    from .data import extract_patient_ids
    model.eval()
    probs, studies = [], []
    for k in range(0, len(dataset), batch_size):
        items = [dataset[i] for i in range(k, min(k + batch_size, len(dataset)))]
        x = torch.stack([it[0] for it in items]).to(device)
        probs.append(torch.sigmoid(model(x).float()).cpu())
        studies += list(extract_patient_ids(dataset, [it[2] for it in items]))
    df = pd.DataFrame(torch.cat(probs).numpy(), index=studies, columns=list(dataset.attr_names))
    df.index.name = "Study"
    return df.groupby("Study").max()


def main(argv=None):
    import pandas as pd
    from .data import ChexpertCSV
    from .models import densenet121, resnet152
    args = build_parser().parse_args(argv)
    if not torch.cuda.is_available():
        raise RuntimeError("predict runs on the GPU only (there is no CPU path)")
    device = torch.device("cuda:%d" % args.cuda)
    ds = ChexpertCSV(args.data_path, "test", args.resize, mini_data=args.mini_data)
    n = len(ds.attr_names)
    if args.model == "densenet121":
        model = densenet121(pretrained=False)
        model.classifier = nn.Linear(model.classifier.in_features, n)
    else:
        model = resnet152(pretrained=False)
        model.fc = nn.Linear(model.fc.in_features, n)
    model = model.to(device)
    if os.path.isdir(args.restore_path):
        files = sorted(os.path.join(args.restore_path, f) for f in os.listdir(args.restore_path)
                       if f.startswith("checkpoint") and f.endswith(".pt"))
        print("Running ensemble prediction using %d checkpoints." % len(files))
    else:
        files = [args.restore_path]
    frames = []
    for f in files:
        model.load_state_dict(torch.load(f, map_location=device)["state_dict"])
        frames.append(predict(model, ds, args.batch_size, device))
    df = frames[0] if len(frames) == 1 else sum(frames[1:], frames[0]) / float(len(frames))      # mean over checkpoints
    df.to_csv(args.output_path)
    return df


if __name__ == "__main__":
    main()
