import sys, torch
sys.path.insert(0, '.')
from dupl_amd import ops
dev = torch.device('cuda:0')
H, hd = 12, 64
D = H*hd
for (B, N) in [(2,197),(1,785),(2,64),(1,130),(3,50),(2,1765),(1,33)]:
    g = torch.Generator().manual_seed(B*1000+N)
    qkv = (torch.randn(B*N, 3*D, generator=g)*1.5).to(dev)
    scale = hd ** -0.5
    q, k, v = (qkv.double().view(B, N, 3, H, hd).permute(2,0,3,1,4)[i] for i in range(3))
    att = (q @ k.transpose(-1,-2))*scale
    ref = (att.softmax(-1) @ v).transpose(1,2).reshape(B*N, D)
    ref_lse = torch.logsumexp(att, dim=-1)
    qkv16 = ops.split16(qkv)
    res = []
    for impl in (0, 1):
        ops.L().dupl_set_attention_fwd16_impl(impl)
        out = torch.empty(B*N, D, device=dev)
        lse = ops.attention_fwd16(qkv16, B, N, H, hd, scale, need_lse=True, out=out)
        sc = float(ref.abs().max())
        res.append((float((out.double()-ref).abs().max())/sc, float((lse.double()-ref_lse).abs().max())))
    print(B, N, "impl0 out %.2e lse %.2e | impl1 out %.2e lse %.2e" % (res[0]+res[1]))
ops.L().dupl_set_attention_fwd16_impl(0)
