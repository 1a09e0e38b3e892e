import sys, torch
sys.path.insert(0, '/root/repo')
from vimo_clip_amd import ops
from vimo_clip_amd._lib import lib
# bit-exactness of variant 2 vs variant 1 on integer data, ragged shapes and epilogues
g = torch.Generator().manual_seed(1)
for (M, N, K) in [(65792, 1024, 128), (3333, 4100, 256), (16448, 1024, 1024), (3900, 4096, 256), (65792, 3072, 1024)]:
    a = torch.randint(-3, 4, (M, K), generator=g).float().cuda().to(torch.bfloat16)
    w = torch.randint(-2, 3, (N, K), generator=g).float().cuda().to(torch.bfloat16)
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    outs = []
    for var in (1, 2):
        lib.vmc_set_gemm_variant(var)
        o1 = ops.linear(a, w, out_dtype=torch.float32)
        o2 = ops.linear(a, w, bias=bias, act=1)
        o3 = ops.linear(a, w, bias=bias, res=res, out_dtype=torch.float32)
        torch.cuda.synchronize()
        outs.append((o1, o2, o3))
    ok = all(torch.equal(x, y) for x, y in zip(*outs))
    ref = (a.float() @ w.float().t())
    print((M, N, K), "v2 == v1:", ok, " v2 exact vs fp32 matmul:", torch.equal(outs[1][0], ref))
