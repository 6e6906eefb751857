"""The CheXpert-small input pipeline of the reference, host side: what `ChexpertSmall` (dataset.py:17-153) and the transform chain of
`fetch_dataloader` (chexpert.py:64-79) do up to the decoded, resized and centre-cropped grey image.  The image leaves as uint8
(1, S, S): the rest of the chain -- `float().div(255)`, `Normalize(0.5330, 0.0349)`, `expand(3,-1,-1)` -- runs on the GPU inside
the models' input kernels (cx_u8_to_nhwc4 / cx_u8_to_nhwc8), so a batch crosses PCIe at one byte per pixel.

Labels follow dataset.py:134-153 (U-Ones): training rows get blanks -> 0 and uncertain (-1) -> 1 on the five competition findings;
the validation file is used as it is; `test` mode reads a bare csv of paths with zero labels (dataset.py:33-37).  Nothing is
downloaded here (no network): a missing data folder raises.
"""
import os

import numpy as np
import torch
from PIL import Image

ATTR_NAMES = ["Atelectasis", "Cardiomegaly", "Consolidation", "Edema", "Pleural Effusion"]      # dataset.py:26
DIR_NAME = "CheXpert-v1.0-small"                                                                  # dataset.py:18-19


def resize_center_crop(img, resize, crop):
    """T.Resize(resize) (shorter side to `resize`, bilinear, aspect kept; skipped when `resize` is falsy) then T.CenterCrop(crop)
    (chexpert.py:67-69), on a PIL image; returns the grey bytes (crop, crop) uint8."""
    img = img.convert("L")
    if resize:
        w, h = img.size
        if w <= h:
            nw, nh = resize, int(resize * h / w)
        else:
            nw, nh = int(resize * w / h), resize
        if (nw, nh) != (w, h):
            img = img.resize((nw, nh), Image.BILINEAR)
    w, h = img.size
    if w < crop or h < crop:                      # CenterCrop pads with zeros when the image is smaller than the crop
        canvas = Image.new("L", (max(w, crop), max(h, crop)), 0)
        canvas.paste(img, ((canvas.size[0] - w) // 2, (canvas.size[1] - h) // 2))
        img, (w, h) = canvas, canvas.size
    left, top = int(round((w - crop) / 2.0)), int(round((h - crop) / 2.0))
    return np.asarray(img.crop((left, top, left + crop, top + crop)), dtype=np.uint8)


class ChexpertCSV(torch.utils.data.Dataset):
    """mode 'train' / 'valid' / 'vis': `root` holds the extracted CheXpert-v1.0-small folder; 'test': `root` is a csv of image paths.
    Items are (uint8 (1,S,S), float32 labels (5,), row index in the source table) like the reference's (img, attr, idx)."""
    attr_names = ATTR_NAMES

    def __init__(self, root, mode="train", resize=None, data_filter=None, mini_data=None):
        import pandas as pd
        assert mode in ("train", "valid", "test", "vis")
        self.mode, self.resize, self.crop = mode, resize, (resize or 320)
        root = os.path.expanduser(root)
        if mode == "test":
            df = pd.read_csv(root, keep_default_na=True)
            self.root, self.csv_path = ".", root
            for a in self.attr_names:
                df[a] = 0.0
        else:
            folder = os.path.join(root, DIR_NAME)
            if not os.path.isdir(folder):
                raise FileNotFoundError("%s not found (the dataset is not downloaded here; pass --synthetic N)" % folder)
            self.root = root
            self.csv_path = os.path.join(folder, "train.csv" if mode == "train" else "valid.csv")      # (also part of the decoded-cache key)
            df = pd.read_csv(self.csv_path, keep_default_na=True)
            if mode == "train":
                df[self.attr_names] = df[self.attr_names].fillna(0).replace(-1, 1)            # U-Ones
                for k, v in (data_filter or {}).items():
                    df = df[df[k] == v]
        if mini_data is not None:
            df = df[:mini_data]
        self.vis_attrs = self.vis_idxs = None
        if mode == "vis":                          # dataset.py:50-68
            from .vis import select_vis_subset
            lab = torch.tensor(df[self.attr_names].fillna(0).values.astype(np.float32))
            self.vis_attrs, groups = select_vis_subset(lab, self.attr_names)
            self.vis_idxs = [[int(df.index[i]) for i in g] for g in groups]
            df = df.iloc[[i for g in groups for i in g]]
        self.data = df
        self.targets = torch.tensor(df[self.attr_names].fillna(0).values.astype(np.float32))

    def __len__(self):
        return len(self.data)

    def enable_decoded_cache(self, max_bytes=None, node_shared=False):
        """Keep every decoded crop in ONE shared-memory uint8 table (N, 1, S, S): the reference's transform chain has no random
        step (Resize + CenterCrop, chexpert.py:67-69), so an image decodes to the same bytes every epoch -- from the second epoch
        on an item is a 100 KB copy instead of a JPEG decode + bilinear resize, and the loader's worker processes (which share the
        table: fork, or torch's shared-memory handles under spawn) stop bounding the GPU.  CheXpert-small at 320x320 is 22.9 GB.
        Returns False (cache off) when the table would exceed `max_bytes`.

        node_shared: the data-parallel ranks of one node (one process per GPU) map ONE table -- two files under /dev/shm named after
        the dataset (folder, mode, size, row count), created by whichever rank comes first and unlinked when the creating process
        exits -- instead of one 22.9 GB table per rank (8 ranks: 183 GB of /dev/shm); a row decoded by any rank's workers serves
        all of them, so `max_bytes` is a per-node budget."""
        n, c = len(self), self.crop
        need = n * c * c
        if max_bytes is not None and need > max_bytes:
            return False
        if node_shared and os.path.isdir("/dev/shm"):
            import atexit
            import hashlib
            # the key names the DATA, not only the folder: the index file's size and mtime go in, so a dataset regenerated in the same
            # place does not map the previous run's pixels
            csv = getattr(self, "csv_path", None)
            try:
                st_ = os.stat(csv) if csv else None
                stamp = "%d|%d" % (st_.st_size, st_.st_mtime_ns) if st_ else "-"
            except OSError:
                stamp = "-"
            key = hashlib.sha1(("%s|%s|%s|%d|%d|%s|%s" % (os.path.abspath(self.root), self.mode, self.resize, c, n, stamp,
                                                            "|".join(map(str, self.data.index[:16])))).encode()).hexdigest()[:16]
            base = "/dev/shm/chexpert_amd_cache_%d_%s" % (os.getuid(), key)
            owner = base + ".owner"                # pid of the creating process: a table whose creator is dead is stale
            try:
                pid = int(open(owner).read().strip() or 0)
                os.kill(pid, 0)                    # raises when that process is gone
            except FileNotFoundError:
                pass
            except (ValueError, ProcessLookupError):
                # left behind by a creator that was killed (no atexit): nobody may trust its `.have` flags -- start over
                for f in (base + ".have", base + ".rows", owner):
                    try:
                        os.unlink(f)
                    except FileNotFoundError:
                        pass
            except PermissionError:
                pass                               # alive, another user's: not ours to judge (the uid is in the name anyway)
            created = False
            try:                                   # O_EXCL: exactly one process of the node creates (and later unlinks) the files
                os.close(os.open(base + ".have", os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600))
                created = True
                os.close(os.open(base + ".rows", os.O_CREAT | os.O_RDWR, 0o600))       # (torch.from_file would create it with the umask's mode)
                with open(os.open(owner, os.O_CREAT | os.O_WRONLY | os.O_TRUNC, 0o600), "w") as f_:
                    f_.write(str(os.getpid()))
            except FileExistsError:
                pass
            # torch.from_file(shared=True) sizes the file (ftruncate: new bytes read as zero) and maps it
            self._have = torch.from_file(base + ".have", shared=True, size=n, dtype=torch.uint8)
            self._cache = torch.from_file(base + ".rows", shared=True, size=need, dtype=torch.uint8).view(n, 1, c, c)
            if created:
                atexit.register(lambda: [os.unlink(f) for f in (base + ".have", base + ".rows", owner) if os.path.exists(f)])
            return True
        self._cache = torch.empty((n, 1, c, c), dtype=torch.uint8).share_memory_()
        self._have = torch.zeros(n, dtype=torch.uint8).share_memory_()      # 1 once row i of the table is complete
        return True

    def cache_fill(self):
        """fraction of the table filled so far"""
        return 0.0 if getattr(self, "_have", None) is None else float(self._have.float().mean())

    def __getitem__(self, i):
        have = getattr(self, "_have", None)
        if have is not None and have[i]:
            return self._cache[i], self.targets[i], int(self.data.index[i])
        path = self.data.iloc[i, 0]                                     # 'Path' is the first column
        with Image.open(os.path.join(self.root, path)) as img:
            px = resize_center_crop(img, self.resize, self.crop)
        x = torch.from_numpy(px.copy()).unsqueeze(0)
        if have is not None:
            self._cache[i].copy_(x)                                     # (two workers decoding the same row write the same bytes)
            have[i] = 1
        return x, self.targets[i], int(self.data.index[i])


def extract_patient_ids(dataset, idxs):
    """dataset.py:156-160: '<...>/patient64541/study1' for each source row index."""
    return dataset.data["Path"].loc[list(idxs)].str.rsplit("/", expand=True, n=1)[0].values
