"""Split-bf16 conv against the fp32 MFMA direct conv and torch CPU fp32 on inputs that punish a lossy product (the cases of
tests/test_gpu_kernels.py::test_conv2d_split_bf16_accuracy_where_fp32_struggles): max |error| against float64.
python tools/conv_bf3_accuracy.py"""
import sys; sys.path.insert(0,'.')
import torch, torch.nn.functional as F
import ivln_ce_amd
from ivln_ce_amd import ops
DEV='cuda:0'
g = torch.Generator().manual_seed(7)
N, Cin, H, Cout, k = 4, 64, 32, 64, 3
for case in ["offset", "range", "tiny"]:
    if case == "offset":
        x = 1000.0 + torch.randn(N, Cin, H, H, generator=g); w = torch.randn(Cout, Cin, k, k, generator=g); w = w - w.mean(dim=(1,2,3), keepdim=True)
    elif case == "range":
        x = torch.randn(N, Cin, H, H, generator=g) * 10.0 ** torch.randint(-6, 7, (N, Cin, H, H), generator=g).float()
        w = torch.randn(Cout, Cin, k, k, generator=g) * 10.0 ** torch.randint(-6, 7, (Cout, Cin, k, k), generator=g).float()
    else:
        x = torch.randn(N, Cin, H, H, generator=g) * 1e-30; w = torch.randn(Cout, Cin, k, k, generator=g) * 1e-3
    ref = F.conv2d(x.double(), w.double(), padding=1)
    ops.TILE_OVERRIDE = 9; got = ops.conv2d(x.to(DEV), w.to(DEV), pad=1, splitk=False)
    ops.TILE_OVERRIDE = 6; fp32 = ops.conv2d(x.to(DEV), w.to(DEV), pad=1, splitk=False); ops.TILE_OVERRIDE = 0
    t = F.conv2d(x, w, padding=1)
    sc = float(ref.abs().max())
    print(case, 'max|ref| %.3e' % sc, 'split %.3e' % float((got.double().cpu()-ref).abs().max()), 'fp32 mfma %.3e' % float((fp32.double().cpu()-ref).abs().max()), 'torch cpu f32 %.3e' % float((t.double()-ref).abs().max()))
