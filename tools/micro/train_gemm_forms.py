"""Time the neighborhood training step's layer products in their two forms (exact-fp32 gemm_multi against the bf16x6
gemm_split_desc) on a real-size batch's shapes: python tools/micro/train_gemm_forms.py [rows]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from desco_amd import ops

m = int(sys.argv[1]) if len(sys.argv) > 1 else 52000
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
agg = torch.randn(m, 256, generator=g).to(dev)
x = torch.randn(m, 64, generator=g).to(dev)
wt = (torch.randn(320, 64, generator=g) / 18).to(dev)
bias = torch.randn(64, generator=g).to(dev)
dz = torch.randn(m, 64, generator=g).to(dev)
wtT = wt.t().contiguous()
out, out2 = torch.empty(m, 64, device=dev), torch.empty(m, 64, device=dev)
D, D2 = torch.empty(m, 320, device=dev), torch.empty(m, 320, device=dev)


def timeit(f, n=50):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


fwd32 = lambda: ops.gemm_multi([dict(a1=agg, a2=x, wt=wt, bias=bias, act=ops.ACT_RELU, out=out)])
fwdx6 = lambda: ops.gemm_split_desc(dict(a1=agg, a2=x, bias=bias, act=ops.ACT_RELU, out=out2), ops.split_bf16_planes_t(wt))
pl = ops.split_bf16_planes_t(wt)
fwdx6_np = lambda: ops.gemm_split_desc(dict(a1=agg, a2=x, bias=bias, act=ops.ACT_RELU, out=out2), pl)
bwd32 = lambda: ops.gemm_multi([dict(a1=dz, wt=wtT, out=D)])
plb = ops.split_bf16_planes(wt)          # [n = 320][k = 64] n-major of dA = dZ Wt^T
bwdx6 = lambda: ops.gemm_split_desc(dict(a1=dz, out=D2), plb)
print(f"rows {m}: forward K=320 N=64  fp32 {timeit(fwd32):.1f} us   bf16x6 {timeit(fwdx6_np):.1f} us (+ split {timeit(fwdx6) - timeit(fwdx6_np):.1f})")
print(f"rows {m}: input grad K=64 N=320  fp32 {timeit(bwd32):.1f} us   bf16x6 {timeit(bwdx6):.1f} us")
fwd32(); fwdx6_np(); bwd32(); bwdx6()
torch.cuda.synchronize()
print("max |d| fwd", float((out - out2).abs().max()), "bwd", float((D - D2).abs().max()))
