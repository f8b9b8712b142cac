"""The floor of a small-batch step under the CURRENT launch structure: why config 2 (32^3 x 4ch, B = 4) does not reach
2.7 ms per step without fewer, fatter launches.

For every launch of the B = 4 plan two lower bounds are measured on this box:
  latency bound     the same layer's launch at B = 1 — one round of workgroups, nothing to amortise: what the launch costs when only its
                    critical path (launch + one workgroup's prologue, chunks, epilogue + the split's hand-over) counts;
  throughput bound  the same layer's launch at B = 32 scaled by 4 / 32 — the rate the kernel reaches when the chip is full.
A launch cannot beat max(latency bound, throughput bound) without changing the kernel or the launch structure; the sum over the plan is the
floor of the eager step, and (floor - kernels overlapping nothing) is what a graph replay of it can reach.  Also measured: the replay floor of
an EMPTY graph with as many kernel nodes (the pure launch cost).

usage: python tools/small_batch_floor.py [channels=4] > profiles/r05_small_batch_floor.log"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dm3d_amd
from dm3d_amd.unet import UNet

C = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = dm3d_amd.UNetConfig(img_size=32, img_channels=C)
W = dm3d_amd.synthetic_weights(cfg, 0)


def profile(B):
    net = UNet(cfg, weights=W, precision="h3")
    plan = net.plan(B, 1000, False)
    net.fill_time_table(list(range(1000)), plan.vec)
    plan.set_context([1])
    plan.x.normal_()
    plan.t_idx.fill_(500)
    plan.run_timed()
    runs = [plan.run_timed() for _ in range(5)]
    rows = []
    for i, (kind, meta, _) in enumerate(runs[0]):
        rows.append((kind, meta.get("desc", kind), meta.get("flops", 0.0), min(r[i][2] for r in runs)))
    del net, plan
    torch.cuda.empty_cache()
    return rows


p1, p4, p32 = profile(1), profile(4), profile(32)


def key(desc):          # a layer's identity across batch sizes: its description without kind names / batch-dependent GEMM shapes
    import re
    d = re.sub(r"^(conv_[a-z0-9_]+|gemm_h3|attn_[a-z]+|mlp_fused|layernorm|softmax|range)\s*", "", desc)
    return re.sub(r"\b(m|b|batch)=\d+", "", d)


# the three plans do not have the same launches (the attention block is 3 fused launches at B = 32, 8 at B <= 8): match convs one to one by
# position among convs, everything else by kind totals
def convs(rows): return [r for r in rows if r[0].startswith("conv")]
def others(rows): return [r for r in rows if not r[0].startswith("conv")]


c1, c4, c32 = convs(p1), convs(p4), convs(p32)
assert len(c1) == len(c4) == len(c32), (len(c1), len(c4), len(c32))
print(f"# config 2-like plan: 32^3 x {C}ch; per conv launch: measured at B = 4 | latency bound (B = 1) | throughput bound (B = 32 x 4/32) | floor = max of the bounds   [ms]")
tot4 = totf = 0.0
for (k1, d1, f1, t1), (k4, d4, f4, t4), (k32, d32, f32_, t32) in zip(c1, c4, c32):
    thr = t32 * 4 / 32
    fl = max(t1, thr)
    tot4 += t4
    totf += fl
    print(f"{d4[:64]:64s} {t4:7.4f} | {t1:7.4f} | {thr:7.4f} | {fl:7.4f}  {'latency' if t1 >= thr else 'throughput'}")
o1, o4, o32 = sum(r[3] for r in others(p1)), sum(r[3] for r in others(p4)), sum(r[3] for r in others(p32))
print(f"convs: {len(c4)} launches, measured {tot4:.3f} ms, floor {totf:.3f} ms")
print(f"attention blocks + norms + range check: B = 4 {len(others(p4))} launches {o4:.3f} ms | B = 1 {len(others(p1))} launches {o1:.3f} ms (latency bound) | "
      f"B = 32 {len(others(p32))} launches {o32:.3f} ms x 4/32 = {o32 * 4 / 32:.3f} ms (throughput bound, the three-launch fused form)")
floor = totf + max(o1, o32 * 4 / 32)
print(f"eager step at B = 4: measured {tot4 + o4:.3f} ms; floor of this launch structure {floor:.3f} ms")

# the pure launch cost: an empty kernel per node, as many nodes as the B = 4 step has
n_nodes = len(p4) + 1
x = torch.zeros(1, device="cuda")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        x.add_(0)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(n_nodes):
            x.add_(0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
e0.record()
for _ in range(50):
    g.replay()
e1.record()
torch.cuda.synchronize()
per = e0.elapsed_time(e1) / 50
print(f"graph replay of {n_nodes} dependent one-element kernels: {per:.3f} ms = {per / n_nodes * 1e3:.2f} us per node (the launch floor of the step's graph)")
print(f"=> a graph-replayed step of this structure cannot go below ~{max(floor - 0.0, per):.2f} ms; the target of 2.7 ms needs fewer launches or kernels "
      f"whose one-round latency is shorter, not a faster steady state")
