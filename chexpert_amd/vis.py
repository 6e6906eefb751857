"""Visualisation outputs of the reference's `--visualize` mode, host side (matplotlib, Agg): the Grad-CAM grids of
chexpert.py:305-361 (`visualize`, `visualize_one`) and the attention-map grids of chexpert.py:363-397 (`vis_attn`), with the
subset selection of dataset.py:50-68 (mode 'vis').  The maps themselves come from the HIP path (gradcam.grad_cam,
AAConv2d.weights); this module only arranges them into the same figures and file names.
"""
import os

import numpy as np
import torch

MEAN, STD = 0.5330, 0.0349        # chexpert.py:71 (Normalize) -- undone before display


def select_vis_subset(targets, attr_names, per_group=3):
    """dataset.py:50-68: up to three examples with exactly one finding for every attribute, then three with no finding, three with
    two findings and three with more; returns (group names, list of index lists)."""
    t = torch.as_tensor(targets).float()
    total = t.sum(1)
    groups = []
    for a in range(len(attr_names)):
        groups.append(torch.nonzero((t[:, a] == 1) & (total == 1)).flatten()[:per_group].tolist())
    groups.append(torch.nonzero(total == 0).flatten()[:per_group].tolist())
    groups.append(torch.nonzero(total == 2).flatten()[:per_group].tolist())
    groups.append(torch.nonzero(total > 2).flatten()[:per_group].tolist())
    return list(attr_names) + ["No findings", "2 conditions", "Multiple conditions"], groups


def _plt():
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    return plt


def _row(plt, axs, img, mask, label, prob, ident, attr_names):
    """One example: [ground truth / predicted probability table | image | Grad-CAM over the image] (chexpert.py:339-361)."""
    order = np.argsort(-prob)
    lab, pr = label[order], prob[order]
    names = [attr_names[i] for i in order]
    cells = np.stack([lab, pr.round(3)], 1)
    axs[0].set_title(str(ident))
    axs[0].table(cellText=cells, rowLabels=names, colLabels=["Ground truth", "Pred. prob"], rowColours=plt.cm.Greens(0.5 * lab),
                 cellColours=plt.cm.Greens(0.5 * cells), cellLoc="center", loc="center")
    axs[0].axis("tight")
    axs[1].set_title("Original image", fontsize=10)
    axs[1].imshow(img, cmap="gray")
    axs[2].set_title("Top class activation \n%s: %.4f" % (names[0], pr[0]), fontsize=10)
    axs[2].imshow(img, cmap="gray")
    axs[2].imshow(mask, cmap="jet", alpha=0.5)
    for ax in axs:
        ax.axis("off")


def visualize(imgs, labels, logits, masks, idents, attr_names, groups, out_dir, step):
    """imgs (N,H,W) in [0,1], labels (N,C), logits (N,C), masks (N,H,W) Grad-CAM in [0,1]; groups = (names, index lists into the N
    examples).  One figure of up to 3 rows x 3 columns per group, saved as vis/vis_<group>_step_<step>.png (chexpert.py:326-337)."""
    plt = _plt()
    os.makedirs(os.path.join(out_dir, "vis"), exist_ok=True)
    probs = 1.0 / (1.0 + np.exp(-np.asarray(logits, dtype=np.float64)))
    written = []
    for name, idxs in zip(*groups):
        fig, axs = plt.subplots(3, 3, figsize=(4 * imgs.shape[1] / 100, 3.3 * imgs.shape[2] / 100), dpi=100, frameon=False)
        fig.suptitle(name)
        for r in range(3):
            if r < len(idxs):
                i = idxs[r]
                _row(plt, axs[r], imgs[i], masks[i], np.asarray(labels[i], dtype=np.float64), probs[i], idents[i], attr_names)
            else:
                for ax in axs[r]:
                    ax.axis("off")
        path = os.path.join(out_dir, "vis", "vis_%s_step_%d.png" % (name.replace(" ", "_"), step))
        plt.savefig(path, dpi=100)
        plt.close(fig)
        written.append(path)
    return written


def vis_attn(x, idents, idxs, attn_layers, out_dir, batch_element=0, window=30):
    """chexpert.py:363-397: for every attention layer a grid with one column per probed pixel (the corners of the centre third of the
    image) and one row per head below a row of images with the probed window marked; the maps are the layer's softmax weights
    (AAConv2d.weights, (B, nh, HW, HW)) averaged over the window.  x: (B,C,H,W) normalised images on the CPU."""
    plt = _plt()
    os.makedirs(os.path.join(out_dir, "vis"), exist_ok=True)
    H, W = x.shape[2:]
    corners = lambda h, w: [(h // 3, w // 3), (h // 3, int(2 * w / 3)), (int(2 * h / 3), w // 3), (int(2 * h / 3), int(2 * w / 3))]
    written = []
    for j, layer in enumerate(attn_layers):
        nh = layer.nh
        attn = layer.weights.detach()[batch_element].float().cpu()
        side = int(round(float(np.sqrt(attn.shape[-1]))))
        attn = attn.reshape(nh, side, side, side, side)
        ws = max(1, int(window * side / H))
        fig, axs = plt.subplots(nh + 1, 4, figsize=(3, 3 / 4 * (1 + nh)), frameon=False)
        fig.suptitle(str(idents[batch_element]), fontsize=8)
        base = (x[batch_element].detach().float().cpu() * STD + MEAN).clamp(0, 1)
        if base.shape[0] == 1:
            base = base.expand(3, -1, -1)
        for ax, (ph, pw) in zip(axs[0], corners(H, W)):
            im = base.clone()
            im[:, max(ph - window, 0):ph + window, max(pw - window, 0):pw + window] = torch.tensor([1.0, 215 / 255, 0.0]).view(3, 1, 1)
            ax.imshow(im.permute(1, 2, 0).numpy())
            ax.axis("off")
        for i, (ph, pw) in enumerate(corners(side, side)):
            for h in range(nh):
                patch = attn[h, max(ph - ws, 0):ph + ws, max(pw - ws, 0):pw + ws]
                axs[h + 1, i].imshow(patch.mean((0, 1)).numpy())
                axs[h + 1, i].axis("off")
        path = os.path.join(out_dir, "vis", "attn_image_idx_%d_%d_layer_%d.png" % (int(idxs[batch_element]), batch_element, j))
        fig.subplots_adjust(0, 0, 1, 0.95, 0.05, 0.05)
        plt.savefig(path)
        plt.close(fig)
        written.append(path)
    return written
