"""Experiment: the bench batch as ONE sampler of B volumes against TWO samplers of B/2 volumes stepping concurrently on two HIP
streams (their kernels interleave on the chip, so the per-kernel prologue / epilogue bursts of one overlap the MFMA phases of the
other).  usage: python tools/two_stream_exp.py [B] [steps] [precision]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
import dm3d_amd
from dm3d_amd.networks import conditional_dm3d as cdm

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
prec = sys.argv[3] if len(sys.argv) > 3 else "h3"
dev = torch.device("cuda:0")
cfg = dm3d_amd.UNetConfig(img_size=32, img_channels=8)
W = dm3d_amd.synthetic_weights(cfg, 0)
margs = SimpleNamespace(timesteps=1000, num_gpus=1, kernel_resize=False, bs=B)


def sampler(b, seed):
    m = cdm.DiffusionModel(32, 1024, 8, None, margs, device=dev, weights=W, precision=prec)
    s = m.sampler((b, 32, 32, 32, 8), context_value=1, seed=seed)
    s.prepare(); s.reset()
    return m, s


def timed(fn):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


m1, s1 = sampler(B, 1)
print(f"one sampler,  B={B}: {timed(s1.step):.2f} ms/step")
del s1, m1
torch.cuda.empty_cache()
ma, sa = sampler(B // 2, 1)
mb, sb = sampler(B // 2, 2)
print(f"one sampler,  B={B // 2}: {timed(sa.step):.2f} ms/step")
st_a, st_b = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    with torch.cuda.stream(st_a): sa.step()
    with torch.cuda.stream(st_b): sb.step()
print(f"two samplers, B={B // 2} each, two streams: {timed(both):.2f} ms per step of both")
