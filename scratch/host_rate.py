import sys, time, torch
sys.path.insert(0, '.')
from chexpert_amd.models import densenet121
from chexpert_amd import synth
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = densenet121(num_classes=14).to(dev)
x = torch.rand(B, 3, 320, 320, device=dev)
t = (torch.rand(B, 14, device=dev) > 0.5).float()
for _ in range(3):
    m.zero_grad(); m.forward_backward(x, t)
torch.cuda.synchronize()
for bs in (B, 8):
    xx, tt = x[:bs].contiguous(), t[:bs].contiguous()
    for _ in range(2):
        m.zero_grad(); m.forward_backward(xx, tt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        m.zero_grad(); m.forward_backward(xx, tt)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("bs=%d host enqueue %.1f ms/step, total %.1f ms/step" % (bs, (t1 - t0) / 5 * 1e3, (t2 - t0) / 5 * 1e3), flush=True)
