"""A/B timing of the fused field kernels on random inputs (uses N2M_HIP_LIB): python tools/mlp_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf2mesh_amd import _lib as L
from nerf2mesh_amd.network import NeRFNetwork
from nerf2mesh_amd.options import make_options
torch.manual_seed(0)
net = NeRFNetwork(make_options(O=True, bound=1, dt_gamma=0, fused_mlp=True)).cuda()
M = int(os.environ.get('M', 2 ** 18))
x = torch.rand(M, 3, device="cuda") * 1.9 - 0.95
d = torch.randn(M, 3, device="cuda")
for shading in ("diffuse", "full"):
    for _ in range(3):
        net.zero_grad(set_to_none=True)
        s, c, p = net(x, d, None, shading, raw_dirs=True)
        (s.sum() + c.sum()).backward()
    torch.cuda.synchronize()
    L.prof_reset(); L.prof_enable(1)
    for _ in range(20):
        net.zero_grad(set_to_none=True)
        s, c, p = net(x, d, None, shading, raw_dirs=True)
        (s.sum() + c.sum()).backward()
    torch.cuda.synchronize(); L.prof_enable(0)
    nf, msf, _ = L.prof_read("mlp_forward"); nb, msb, _ = L.prof_read("mlp_backward")
    print(f"M={M} {shading:8s} forward {1e3*msf/nf:7.1f} us  backward {1e3*msb/nb:7.1f} us   lib={os.environ.get('N2M_HIP_LIB','default')[-28:]}")
