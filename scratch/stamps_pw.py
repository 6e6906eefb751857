"""Phase shares of pw_bwd2 (diagnostic build CX_PW_BWD_DBG=256): python scratch/stamps_pw.py"""
import ctypes, os, sys, numpy as np, torch
os.environ["CX_PW_BWD_DBG"] = "256"
sys.path.insert(0, '.')
from chexpert_amd import ops, _lib
dev = torch.device('cuda:0')
bf = torch.bfloat16
B = 256
for hw, ct, cin in [(80, 256, 128), (40, 512, 256), (20, 1024, 512), (10, 1024, 992)]:
    M = B * hw * hw
    buf = (torch.randn(B, hw, hw, ct, device=dev) * 0.5).to(bf)
    gbuf = torch.zeros(B, hw, hw, ct, device=dev, dtype=bf)
    y1 = torch.randn(B, hw, hw, 128, device=dev).to(bf); dz2 = torch.randn(B, hw, hw, 128, device=dev).to(bf)
    ones = torch.ones(1024, device=dev); zeros = torch.zeros(1024, device=dev)
    st = torch.zeros(2, 16 * 1024, device=dev)
    wf = torch.randn(128 * cin, device=dev).to(bf); dw = torch.zeros(128, cin, 1, 1, device=dev)
    for _ in range(3):
        ops.conv_gemm(dz2, wf, gbuf[..., :cin], N=cin, prologue=ops.PRO_AFFINE2, x2=y1, pa=ones, pb=zeros, pc=zeros, epilogue=ops.EPI_MASK,
                      ex=buf[..., :cin], e_sc=ones, e_sh=zeros, e_mu=zeros, e_r=ones, e_scale=ones, stat_sum=st[0], stat_sq=st[1],
                      accumulate=True, stat_replicas=16, stat_rstride=1024, fused_dw=dw)
    torch.cuda.synchronize()
    host = (ctypes.c_ulonglong * (1024 * 8))()
    rc = _lib.lib().__getattr__("dbg_pw_bwd2_stamps") if False else ctypes.CDLL(_lib.LIB_PATH).dbg_pw_bwd2_stamps(host, 1024 * 8)
    a = np.frombuffer(host, dtype=np.uint64).reshape(1024, 8).astype(np.float64)
    a = a[a[:, 6] > 0]
    per = a[:, :6] / a[:, 6:7]
    names = ["stage->LDS", "barrier1", "dgrad+loads", "epilogue+loads", "barrier2", "wgrad"]
    med = np.median(per, 0)
    print("hw=%d cin=%d: tiles/wg %.0f, cycles per tile %.0f: " % (hw, cin, np.median(a[:, 6]), med.sum()) +
          ", ".join("%s %.0f" % (n, v) for n, v in zip(names, med)), flush=True)
    del buf, gbuf, y1, dz2
