"""Timing ablations of conv3x3_pc_fwd_kernel (make -C chexpert_amd/csrc abl-pc ABL=<bits>): each library in a child process."""
import os, shutil, subprocess, sys
sys.path.insert(0, ".")
child = r'''
import sys, torch
sys.path.insert(0, '.')
from chexpert_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16; B = 256
def timeit(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
out = []
for hw, ctot in ((80, 256), (40, 512)):
    z1 = (torch.randn(B, hw, hw, 128, device=dev) * 0.5).to(bf)
    buf = (torch.randn(B, hw, hw, ctot, device=dev) * 0.5).to(bf)
    w = (torch.randn(9 * 32 * 128, device=dev) * 0.05).to(bf)
    one, zero = torch.ones(128, device=dev), torch.zeros(128, device=dev)
    cap = 4096
    st = torch.zeros(2, cap * 128, device=dev)
    ys = buf[..., 64:96]
    f = lambda: ops.conv_gemm(z1, w, ys, N=32, kh=3, kw=3, pad=1, prologue=ops.PRO_AFFINE_RELU, pa=one, pb=zero, stat_sum=st[0], stat_sq=st[1],
                              stat_det=True, stat_replicas=cap, stat_rstride=32, hint=ops.kernel_hint(-1, 8))
    out.append("%dx%d %.1f us" % (hw, hw, min(timeit(f), timeit(f))))
print(sys.argv[1], " | ".join(out), flush=True)
'''
from chexpert_amd import _lib
shutil.copy(_lib.LIB_PATH, "/tmp/lib_product.so")
libs = [("product", "/tmp/lib_product.so")] + [(os.path.basename(f), os.path.join("scratch", f)) for f in sorted(os.listdir("scratch")) if f.startswith("libpc_abl")]
for name, path in libs:
    shutil.copy(path, _lib.LIB_PATH)
    subprocess.run([sys.executable, "-c", child, name], check=False)
shutil.copy("/tmp/lib_product.so", _lib.LIB_PATH)
