"""Phase shares of the 3x3 input-gradient ring kernel (diagnostic build -DCX_RING_STAMPS, scratch/libstamp_ring.so):
python scratch/stamps_ring.py   (copies the diagnostic library over the product one in ITS snapshot only)"""
import ctypes, os, shutil, sys, numpy as np, torch
sys.path.insert(0, '.')
from chexpert_amd import _lib
shutil.copy("scratch/libstamp_ring.so", _lib.LIB_PATH)
from chexpert_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16; B = 256
names = ["restart", "wait+stage", "barrier1", "issue", "items:multiply", "items:epilogue", "barrier2"]
for hw, ctot in ((80, 256), (40, 512), (20, 1024)):
    z1 = (torch.randn(B, hw, hw, 128, device=dev) * 0.5).to(bf)
    buf = (torch.randn(B, hw, hw, ctot, device=dev) * 0.5).to(bf)
    gbuf = (torch.randn(B, hw, hw, ctot, device=dev) * 0.5).to(bf)
    dz = torch.empty(B, hw, hw, 128, device=dev, dtype=bf)
    w = (torch.randn(9 * 32 * 128, device=dev) * 0.05).to(bf)
    one, zero = torch.ones(128, device=dev), torch.zeros(128, device=dev)
    cap = 4096
    st = torch.zeros(2, cap * 128, device=dev)
    ys, gs = buf[..., 64:96], gbuf[..., 64:96]
    d = lambda: ops.conv_gemm(gs, w, dz, N=128, kh=3, kw=3, pad=1, prologue=ops.PRO_AFFINE2, x2=ys, pa=one[:32], pb=zero[:32], pc=zero[:32],
                              epilogue=ops.EPI_MASK, ex=z1, e_sc=one, e_sh=zero, e_mu=zero, e_r=one, e_scale=one, stat_sum=st[0], stat_sq=st[1],
                              stat_det=True, stat_replicas=cap, stat_rstride=128)
    one128 = torch.ones(128, device=dev); zero128 = torch.zeros(128, device=dev)
    f = lambda: ops.conv_gemm(z1, w, ys, N=32, kh=3, kw=3, pad=1, prologue=ops.PRO_AFFINE_RELU, pa=one128, pb=zero128, stat_sum=st[0], stat_sq=st[1],
                              stat_det=True, stat_replicas=cap, stat_rstride=32)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    kname = _lib.lib().cx_last_kernel().decode()
    host = (ctypes.c_ulonglong * (1024 * 8))()
    ctypes.CDLL(_lib.LIB_PATH).dbg_ring_fwd_stamps(host, 1024 * 8)
    a = np.frombuffer(host, dtype=np.uint64).reshape(1024, 8).astype(np.float64)[:256]
    a = a[a[:, 7] > 0]
    med = np.median(a[:, :7] / a[:, 7:8], 0)
    fn = ["between steps", "wait+stage", "barrier1", "issue", "sub-tiles:multiply", "sub-tiles:epilogue", "barrier2"]
    print("%dx%d FORWARD (%s, %.1f us): steps/wg %.0f, core clocks per step %.0f: " % (hw, hw, kname, e0.elapsed_time(e1) * 1e3, np.median(a[:, 7]), med.sum()) +
          ", ".join("%s %.0f" % (nm, v) for nm, v in zip(fn, med)), flush=True)
    for _ in range(3): d()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); d(); e1.record(); torch.cuda.synchronize()
    host = (ctypes.c_ulonglong * (1024 * 8))()
    ctypes.CDLL(_lib.LIB_PATH).dbg_ring_stamps(host, 1024 * 8)
    a = np.frombuffer(host, dtype=np.uint64).reshape(1024, 8).astype(np.float64)
    a = a[:256]
    a = a[a[:, 7] > 0]
    per = a[:, :7] / a[:, 7:8]
    med = np.median(per, 0)
    print("%dx%d (%s, %.1f us): steps/wg %.0f, core clocks per step %.0f: " % (hw, hw, _lib.lib().cx_last_kernel().decode(), e0.elapsed_time(e1) * 1e3, np.median(a[:, 7]), med.sum()) +
          ", ".join("%s %.0f" % (nm, v) for nm, v in zip(names, med)), flush=True)
