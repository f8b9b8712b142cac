"""Times dm3d_attention_group (two passes, B=32, L=512, units 256: one CrossAttentionBlock's attention) fused vs the three-launch form.
usage: python tools/attn_bench.py [B]      (DM3D_ATTN_FUSED=0 forces the three-launch form inside the entry)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dm3d_amd import _lib, ops
from dm3d_amd._lib import lib, check

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L, u = 512, 256
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
enc = lambda t: ops.split_h2(t.to(dev).contiguous()).view(*t.shape[:-1], -1)
q, k, v = (torch.randn(B, L, u, generator=g) for _ in range(3))
qh, kh, vth = enc(q), enc(k), enc(v.transpose(1, 2).contiguous())
res = torch.randn(B, L, u, generator=g).to(dev)
outs = [torch.empty(B, L, u, device=dev) for _ in range(2)]
descs = (_lib.AttentionDesc * 2)()
for d, o in zip(descs, outs):
    d.q, d.ldq = qh.data_ptr(), u
    d.k, d.ldk, d.stride_k = kh.data_ptr(), u, L * u
    d.vt, d.ldv, d.stride_vt = vth.data_ptr(), L, u * L
    d.out, d.ldo, d.res = o.data_ptr(), u, res.data_ptr()
    d.batch, d.lq, d.lk, d.c, d.scale, d.precision, d.fmt = B, L, L, u, u ** -0.5, _lib.PREC_H3, _lib.FMT_H2
scratch = torch.empty(lib().dm3d_attention_workspace_bytes(B, L, L) // 4, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    check(lib().dm3d_attention_group(descs, 2, scratch.data_ptr(), st), "attn")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n):
    check(lib().dm3d_attention_group(descs, 2, scratch.data_ptr(), st), "attn")
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
fl = 2 * 2 * 2.0 * B * L * L * u
print(f"attention x2 passes B={B}: {ms * 1e3:.1f} us per launch group, {fl / ms / 1e9:.1f} TF/s algorithmic ({3 * fl / ms / 1e9:.0f} executed)")
