"""Input pipeline that can feed the GPU: what `DataLoader(dataset, batch_size, shuffle, num_workers=16)` does for the reference
(/root/reference/chexpert.py:76-79), re-designed around the one-byte-per-pixel hand-over of `chexpert_amd/data.py`.

  * W worker PROCESSES (default 16, the reference's count) decode / resize / centre-crop (PIL) -- or, for `--synthetic`, hash --
    straight into slots of ONE shared-memory uint8 ring `(K, B, 1, S, S)`; labels and row indices ride in two small rings.  Work
    is dealt per chunk of a batch (16 images), so all workers fill the SAME batch: the latency of a batch is its decode time / W.
  * Workers never touch the GPU: they are forked before the first GPU call when the loader is built first (cli.py does), otherwise
    spawned.
  * The consumer registers the ring as pinned memory (hipHostRegister) once the GPU is up and uploads a finished slot with ONE
    non-blocking copy on its own copy stream into one of three device buffers; the compute stream only waits on that copy's
    event, so decode, upload and the training step of three consecutive batches overlap.
  * The rest of the transform chain (`/255`, `Normalize`, `expand(3)`) runs inside the models' first kernel (cx_u8_to_nhwc4/8).

`python -m chexpert_amd.loader --bench` measures decode throughput on a generated folder of JPEGs of CheXpert-small's geometry
(390x320 grey), the figure to set beside the GPU's images/s.
"""
import os
import queue
import time
import traceback

import numpy as np
import torch
import torch.multiprocessing as mp

CHUNK = 16


def _worker(ds, ring_x, ring_t, ring_i, tasks, done, wid):
    torch.set_num_threads(1)
    try:
        while True:
            job = tasks.get()
            if job is None:
                return
            slot, start, idxs = job
            for j, i in enumerate(idxs):
                x, t, src = ds[i]
                ring_x[slot, start + j].copy_(x)
                ring_t[slot, start + j].copy_(t)
                ring_i[slot, start + j] = int(src)
            done.put((slot, len(idxs)))
    except Exception:                                   # surfaces in the consumer, with the worker's traceback
        done.put(("error", "loader worker %d:\n%s" % (wid, traceback.format_exc())))


class RingLoader:
    """Iterate minibatches `(x uint8 (n,1,S,S), target float32 (n,C), idx int64 (n,))` of `dataset` over given index lists.

    dataset[i] -> (uint8 (1,S,S) tensor, float32 (C,) tensor, int).  `device` None: CPU tensors (copies of the ring slot).
    With a CUDA device the tensors handed out live in device buffers that are re-used three batches later; the consumer must
    enqueue its use of a batch on the current stream before asking for the next one (a training loop does)."""

    def __init__(self, dataset, batch_size, num_workers=16, slots=4, device=None):
        x0, t0, _ = dataset[0]
        assert x0.dtype == torch.uint8 and x0.dim() == 3
        self.ds, self.B, self.W, self.K = dataset, int(batch_size), int(num_workers), max(2, int(slots))
        self.device = device if (device is not None and torch.device(device).type == "cuda") else None
        self.ring_x = torch.empty((self.K, self.B) + tuple(x0.shape), dtype=torch.uint8).share_memory_()
        self.ring_t = torch.zeros((self.K, self.B) + tuple(t0.shape), dtype=torch.float32).share_memory_()
        self.ring_i = torch.zeros((self.K, self.B), dtype=torch.int64).share_memory_()
        # fork is only safe while this process has not initialised the GPU runtime; afterwards spawn (slower start, same behaviour)
        method = "spawn" if torch.cuda.is_initialized() else "fork"
        ctx = mp.get_context(method)
        self.tasks, self.done = ctx.Queue(), ctx.Queue()
        self.procs = []
        for w in range(self.W):
            p = ctx.Process(target=_worker, args=(dataset, self.ring_x, self.ring_t, self.ring_i, self.tasks, self.done, w), daemon=True)
            p.start()
            self.procs.append(p)
        self.start_method = method
        self._pinned = False
        self._dev = None
        self._pending = 0               # chunks handed to the workers and not yet reported done
        self.closed = False

    # ---- device side
    def _device_setup(self):
        if self._dev is not None or self.device is None:
            return
        dev = torch.device(self.device)
        for t in (self.ring_x, self.ring_t):            # page-lock the shared rings in place: the copies below become true DMA
            try:
                rc = torch.cuda.cudart().cudaHostRegister(t.data_ptr(), t.numel() * t.element_size(), 0)
                self._pinned = self._pinned or int(rc) == 0
            except Exception:
                pass
        self._dev = dict(x=[torch.empty_like(self.ring_x[0], device=dev) for _ in range(3)],
                         t=[torch.empty_like(self.ring_t[0], device=dev) for _ in range(3)],
                         used=[None, None, None], stream=torch.cuda.Stream(device=dev), k=0)

    def _submit(self, slot, idxs):
        for s in range(0, len(idxs), CHUNK):
            self.tasks.put((slot, s, idxs[s:s + CHUNK]))
            self._pending += 1

    def _next_done(self):
        try:
            msg = self.done.get(timeout=120)
        except queue.Empty:
            dead = [p.pid for p in self.procs if not p.is_alive()]
            raise RuntimeError("input pipeline stalled for 120 s (dead workers: %s)" % dead)
        if msg[0] == "error":
            raise RuntimeError(msg[1])
        self._pending -= 1
        return msg

    def _wait_slot(self, slot, need, counts):
        while counts[slot] < need:
            msg = self._next_done()
            counts[msg[0]] += msg[1]

    def batches(self, indices, drop_last=False):
        """One pass over `indices` in order, `batch_size` at a time."""
        if self.closed:
            raise RuntimeError("loader is closed")
        indices = [int(i) for i in indices]
        chunks = [indices[k:k + self.B] for k in range(0, len(indices), self.B)]
        if drop_last and chunks and len(chunks[-1]) < self.B:
            chunks.pop()
        if self.W == 0:                                 # in-process path (the reference's num_workers=0 validation loader)
            for idx in chunks:
                items = [self.ds[i] for i in idx]
                x, t = torch.stack([it[0] for it in items]), torch.stack([it[1] for it in items])
                ii = torch.tensor([int(it[2]) for it in items])
                yield (x.to(self.device), t.to(self.device), ii) if self.device is not None else (x, t, ii)
            return
        self._device_setup()
        while self._pending > 0:                        # a previous pass was abandoned half-way: let its chunks finish first
            self._next_done()
        counts = [0] * self.K
        copy_ev = [None] * self.K                       # upload of the slot's previous content (it may be refilled once this is done)
        nxt = 0
        for nxt in range(min(self.K, len(chunks))):
            self._submit(nxt, chunks[nxt])
        nxt = min(self.K, len(chunks))
        for b, idx in enumerate(chunks):
            slot, n = b % self.K, len(idx)
            self._wait_slot(slot, n, counts)
            counts[slot] = 0
            ii = self.ring_i[slot, :n].clone()
            if self.device is None:
                out = (self.ring_x[slot, :n].clone(), self.ring_t[slot, :n].clone(), ii)
            else:
                d = self._dev
                k = d["k"]
                d["k"] = (k + 1) % 3
                cur = torch.cuda.current_stream()
                for j in range(3):                       # whatever was handed out earlier has been enqueued for use by now
                    if d["used"][j] == "pending":
                        ev = torch.cuda.Event()
                        ev.record(cur)
                        d["used"][j] = ev
                if isinstance(d["used"][k], torch.cuda.Event):
                    d["stream"].wait_event(d["used"][k])       # the step that read this device buffer three batches ago
                with torch.cuda.stream(d["stream"]):
                    d["x"][k][:n].copy_(self.ring_x[slot, :n], non_blocking=True)
                    d["t"][k][:n].copy_(self.ring_t[slot, :n], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(d["stream"])
                copy_ev[slot] = ev
                cur.wait_event(ev)
                d["used"][k] = "pending"
                out = (d["x"][k][:n], d["t"][k][:n], ii)
            if nxt < len(chunks):                        # refill this slot with the batch K ahead
                if copy_ev[slot] is not None:
                    copy_ev[slot].synchronize()          # (K >= 2 batches old by the time it is needed again: already complete)
                self._submit(slot, chunks[nxt])
                nxt += 1
            yield out

    def close(self):
        if self.closed:
            return
        self.closed = True
        for _ in self.procs:
            self.tasks.put(None)
        for p in self.procs:
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()
        if self._pinned:
            for t in (self.ring_x, self.ring_t):
                try:
                    torch.cuda.cudart().cudaHostUnregister(t.data_ptr())
                except Exception:
                    pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- throughput of the decode path on this host ---------------------------------------------------------------------------------
def make_jpeg_folder(root, n=256, w=390, h=320, seed=0):
    """A stand-in CheXpert-v1.0-small folder (no network here): n grey JPEGs of the real files' geometry with smooth content (so
    that they compress like radiographs, ~20-30 KB), plus train.csv / valid.csv with the reference's columns."""
    from PIL import Image
    from .data import ATTR_NAMES, DIR_NAME
    rng = np.random.RandomState(seed)
    folder = os.path.join(root, DIR_NAME)
    rows = []
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    for i in range(n):
        d = os.path.join(folder, "train", "patient%05d" % i, "study1")
        os.makedirs(d, exist_ok=True)
        cx, cy, s = rng.uniform(0.3, 0.7) * w, rng.uniform(0.3, 0.7) * h, rng.uniform(60, 140)
        img = 200.0 * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s)) + rng.normal(0, 6, (h, w))
        Image.fromarray(np.clip(img, 0, 255).astype(np.uint8), "L").save(os.path.join(d, "view1_frontal.jpg"), quality=90)
        lab = rng.choice([1.0, 0.0, -1.0, np.nan], size=len(ATTR_NAMES), p=[0.25, 0.45, 0.1, 0.2])
        rows.append(["%s/train/patient%05d/study1/view1_frontal.jpg" % (DIR_NAME, i), "Male", 50, "Frontal", "AP"] + list(lab))
    import pandas as pd
    df = pd.DataFrame(rows, columns=["Path", "Sex", "Age", "Frontal/Lateral", "AP/PA"] + ATTR_NAMES)
    df.to_csv(os.path.join(folder, "train.csv"), index=False)
    df[:max(8, n // 8)].to_csv(os.path.join(folder, "valid.csv"), index=False)
    return folder


def bench(workers=(1, 4, 8, 16), n=512, batch=256, resize=320, seconds=6.0):
    import tempfile
    from .data import ChexpertCSV
    out = {}
    with tempfile.TemporaryDirectory() as root:
        make_jpeg_folder(root, n=n)
        ds = ChexpertCSV(root, "train", resize=resize)
        for w in workers:
            ld = RingLoader(ds, batch, num_workers=w, slots=4)
            idx = list(range(len(ds))) * 64
            t0, seen = time.perf_counter(), 0
            for x, t, i in ld.batches(idx):
                seen += x.shape[0]
                if time.perf_counter() - t0 > seconds:
                    break
            out[w] = seen / (time.perf_counter() - t0)
            ld.close()
        # the same loader over the decoded-image cache (ChexpertCSV.enable_decoded_cache): one pass fills the table, then timed
        ds = ChexpertCSV(root, "train", resize=resize)
        ds.enable_decoded_cache()
        for w in (workers[-1], 4):
            ld = RingLoader(ds, batch, num_workers=w, slots=4)
            for _ in ld.batches(list(range(len(ds)))):
                pass
            idx = list(range(len(ds))) * 256
            t0, seen = time.perf_counter(), 0
            for x, t, i in ld.batches(idx):
                seen += x.shape[0]
                if time.perf_counter() - t0 > seconds:
                    break
            out["cached_%d" % w] = seen / (time.perf_counter() - t0)
            ld.close()
    return out


if __name__ == "__main__":
    import argparse
    import json
    ap = argparse.ArgumentParser()
    ap.add_argument("--bench", action="store_true")
    ap.add_argument("--workers", default="1,4,8,16")
    a = ap.parse_args()
    if a.bench:
        r = bench(tuple(int(w) for w in a.workers.split(",")))
        print(json.dumps({"metric": "decoded+resized+cropped images/sec (PIL, 390x320 JPEG -> 320x320 uint8)", "host_cores": os.cpu_count(),
                          "by_workers": {str(k): round(v, 1) for k, v in r.items() if not str(k).startswith("cached")},
                          "decoded_cache_by_workers": {str(k)[7:]: round(v, 1) for k, v in r.items() if str(k).startswith("cached")}}))
