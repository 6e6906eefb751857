"""Phase shares of conv3x3_pc_fwd_kernel (diagnostic build -DCX_PC_STAMPS: make -C chexpert_amd/csrc stamps-pc), wave 0 (consumer) and
wave 4 (producer) of every workgroup; copies the diagnostic library over the product one in ITS snapshot only."""
import ctypes, os, shutil, sys, numpy as np, torch
sys.path.insert(0, '.')
from chexpert_amd import _lib
shutil.copy("scratch/libstamp_pc.so", _lib.LIB_PATH)
from chexpert_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16; B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
raw = ctypes.CDLL(_lib.LIB_PATH)
for hw, ctot in ((80, 256), (40, 512), (20, 1024), (10, 1024)):
    z1 = (torch.randn(B, hw, hw, 128, device=dev) * 0.5).to(bf)
    buf = (torch.randn(B, hw, hw, ctot, device=dev) * 0.5).to(bf)
    w = (torch.randn(9 * 32 * 128, device=dev) * 0.05).to(bf)
    one, zero = torch.ones(128, device=dev), torch.zeros(128, device=dev)
    cap = 4096
    st = torch.zeros(2, cap * 128, device=dev)
    ys = buf[..., 64:96]
    f = lambda: ops.conv_gemm(z1, w, ys, N=32, kh=3, kw=3, pad=1, prologue=ops.PRO_AFFINE_RELU, pa=one, pb=zero, stat_sum=st[0], stat_sq=st[1],
                              stat_det=True, stat_replicas=cap, stat_rstride=32, hint=ops.kernel_hint(-1, 8))
    for _ in range(3): f()
    torch.cuda.synchronize()
    raw.dbg_pc_stamps(None, 0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    host = (ctypes.c_ulonglong * (1024 * 16))()
    raw.dbg_pc_stamps(host, 1024 * 16)
    a = np.frombuffer(host, dtype=np.uint64).reshape(1024, 16).astype(np.float64)[:256]
    a = a[a[:, 7] > 0]
    c = np.median(a[:, :6] / a[:, 7:8], 0)
    p = np.median(a[:, 8:14] / a[:, 15:16], 0)
    print("%dx%d (%s, %.1f us): steps/wg %.0f | consumer per step %.0f: loop %.0f, barrier A %.0f, multiply %.0f, partials out %.0f, barrier B %.0f, sum + epilogue %.0f | "
          "producer per step %.0f: loop %.0f, stage 1st half %.0f, barrier B %.0f, stage 2nd half %.0f, issue %.0f, barrier A %.0f" % (
              hw, hw, _lib.lib().cx_last_kernel().decode(), e0.elapsed_time(e1) * 1e3, np.median(a[:, 7]), c.sum(), c[0], c[1], c[2], c[3], c[4], c[5],
              p.sum(), p[0], p[1], p[2], p[3], p[4], p[5]), flush=True)
