"""Phase shares of conv3x3_pc_wgrad_kernel (diagnostic build: make -C chexpert_amd/csrc stamps-pc), wave 0 (consumer) / wave 4 (producer)."""
import ctypes, os, shutil, sys, numpy as np, torch
sys.path.insert(0, '.')
from chexpert_amd import _lib
shutil.copy("scratch/libstamp_pc.so", _lib.LIB_PATH)
from chexpert_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16; B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
raw = ctypes.CDLL(_lib.LIB_PATH)
for hw in (80, 40):
    y1 = (torch.randn(B, hw, hw, 128, device=dev) * 0.5).to(bf)
    gs = (torch.randn(B, hw, hw, 32, device=dev) * 0.5).to(bf)
    one, zero = torch.ones(128, device=dev), torch.zeros(128, device=dev)
    dw = torch.zeros(32, 128, 3, 3, device=dev)
    f = lambda: ops.conv_wgrad(gs, y1, dw, kh=3, kw=3, pad=1, x_prologue=ops.PRO_AFFINE_RELU, pa=one, pb=zero, hint=ops.kernel_hint(-1, 8))
    for _ in range(3): f()
    torch.cuda.synchronize()
    raw.dbg_pc_stamps(None, 0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    host = (ctypes.c_ulonglong * (1024 * 16))()
    raw.dbg_pc_stamps(host, 1024 * 16)
    a = np.frombuffer(host, dtype=np.uint64).reshape(1024, 16).astype(np.float64)[:256]
    a = a[a[:, 7] > 0]
    c = np.median(a[:, :3] / a[:, 7:8], 0)
    p = np.median(a[:, 8:13] / a[:, 15:16], 0)
    print("%dx%d (%.1f us incl. the slab reduce): steps/wg %.0f | consumer per step %.0f: loop %.0f, barrier wait %.0f, k-steps %.0f | "
          "producer per step %.0f: loop %.0f, wait loads %.0f, stage %.0f, issue %.0f, barrier %.0f" % (
              hw, hw, e0.elapsed_time(e1) * 1e3, np.median(a[:, 7]), c.sum(), c[0], c[1], c[2], p.sum(), p[0], p[1], p[2], p[3], p[4]), flush=True)
