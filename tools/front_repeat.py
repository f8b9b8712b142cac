"""dm3d_attn_front run to run: the five outputs of repeated launches on one input must be bit-identical (a race inside the kernel shows here).
Launches ALTERNATE between two inputs (and a conv launch scribbles over the LDS in between): repeated launches of one kernel on one input leave
every CU's LDS holding exactly the data the next launch will write, which hides a read-before-write.
usage: python tools/front_repeat.py [m] [repeats]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dm3d_amd import ops
dev = torch.device("cuda:0")
m = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
u = 256
g = torch.Generator().manual_seed(7)
c = lambda t: t.to(dev).contiguous()
x = c(torch.randn(m, u, generator=g) * 2.0)
tile = lambda w: ops.pack_front_weights(ops.split_h2(c(w)), w.shape[0])
w_in, w_qk, w_v = tile(torch.randn(u, u, generator=g) / 16.0), tile(torch.randn(2 * u, u, generator=g) / 16.0), tile(torch.randn(u, u, generator=g) / 16.0)
b_in, b_qk, b_v = c(torch.randn(u, generator=g) * 0.1), c(torch.randn(2 * u, generator=g) * 0.1), c(torch.randn(u, generator=g) * 0.1)
norms = [(c(torch.rand(u, generator=g) + 0.5), c(torch.randn(u, generator=g) * 0.2)) for _ in range(3)]
names = ("y", "qk", "v_t", "q2", "n3")
xs = [x, c(torch.randn(m, u, generator=g) * 3.0 + 1.0)]
firsts = [None, None]
bad = {n: 0 for n in names}
cx = torch.randn(2, 16, 16, 16, 64, device=dev)
ck = torch.randn(3, 3, 3, 64, 64, device=dev) * 0.05
cw = ops.pack_weights(ck) if hasattr(ops, "pack_weights") else None
for r in range(reps):
    outs = ops.attn_front(xs[r & 1], w_in, b_in, w_qk, b_qk, w_v, b_v, norms)
    if cw is not None and os.environ.get("SCRIBBLE", "1") == "1":
        ops.conv3d(cx, cw, 64, 3)
    torch.cuda.synchronize()
    first = firsts[r & 1]
    if first is None:
        firsts[r & 1] = [o.clone() for o in outs]
        continue
    for n, a, b in zip(names, first, outs):
        if not torch.equal(a.view(torch.int32), b.view(torch.int32)):
            bad[n] += 1
            d = (a.view(torch.int32) != b.view(torch.int32)).nonzero()
            if bad[n] == 1:
                print(f"rep {r}: {n} differs in {d.shape[0]} words; first at {d[0].tolist()}, rows {sorted(set(d[:, 0].tolist()))[:8]} cols {sorted(set(d[:, 1].tolist()))[:16]}")
print(f"m={m} reps={reps}: mismatching launches per output: {bad}")
