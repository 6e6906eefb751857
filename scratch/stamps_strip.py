"""Phase shares of the batched 3x3 weight-gradient kernel (diagnostic build -DCX_STRIP_STAMPS, scratch/libstamp.so):
python scratch/stamps_strip.py   (the script copies the diagnostic library over the product one in ITS snapshot only)"""
import ctypes, os, shutil, sys, numpy as np, torch
sys.path.insert(0, '.')
from chexpert_amd import _lib
shutil.copy("scratch/libstamp.so", _lib.LIB_PATH)
from chexpert_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16; B = 256
ops.set_det_wgrad(True)
names = ["restart", "wait+stage", "barrier1", "issue", "multiply", "barrier2"]
for hw, n in ((40, 12), (20, 24), (10, 16)):
    items = []
    for i in range(n):
        g = torch.randn(B, hw, hw, 32, device=dev).to(bf)
        x = torch.randn(B, hw, hw, 128, device=dev).to(bf)
        items.append((g, x, torch.rand(128, device=dev) + 0.5, torch.rand(128, device=dev) - 0.5, torch.zeros(32, 128, 3, 3, device=dev)))
    for _ in range(3):
        ops.wgrad_defer_begin(dev)
        assert ops.conv3x3_wgrad_batch(items)
        ops.wgrad_defer_flush(dev)
    torch.cuda.synchronize()
    host = (ctypes.c_ulonglong * (1024 * 8))()
    ctypes.CDLL(_lib.LIB_PATH).dbg_strip_stamps(host, 1024 * 8)
    a = np.frombuffer(host, dtype=np.uint64).reshape(1024, 8).astype(np.float64)
    a = a[a[:, 6] > 0]
    per = a[:, :6] / a[:, 6:7]
    med = np.median(per, 0)
    print("%dx%d: %d workgroups seen, steps/wg %.0f, core clocks (s_memtime) per step %.1f: " % (hw, hw, len(a), np.median(a[:, 6]), med.sum()) +
          ", ".join("%s %.1f" % (nm, v) for nm, v in zip(names, med)), flush=True)
