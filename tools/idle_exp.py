"""Is the whole step bound by the chip's power cap averaged over milliseconds?  Time K denoising steps (B=32, 32^3x8ch, HIP-graph replays)
with an idle gap of G microseconds (one spinning thread: torch.cuda._sleep) inserted after every step.  If the chip were limited by its
instruction schedule, the step would grow by G; if it is limited by average power, part of the gap comes back as a higher clock.
usage: python tools/idle_exp.py [gap_us ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch, dm3d_amd
from dm3d_amd.networks import conditional_dm3d as cdm
gaps = [int(a) for a in sys.argv[1:]] or [0, 500, 1000, 2000, 4000]
cfg = dm3d_amd.UNetConfig(img_size=32, img_channels=8)
m = cdm.DiffusionModel(32, 1024, 8, None, SimpleNamespace(timesteps=1000, num_gpus=1, kernel_resize=False, bs=32), weights=dm3d_amd.synthetic_weights(cfg, seed=0))
smp = m.sampler((32, 32, 32, 32, 8), context_value=1, seed=1).prepare()
smp.reset()
# calibrate the spin kernel: cycles per microsecond
torch.cuda._sleep(1000); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(20_000_000); e1.record(); torch.cuda.synchronize()
cyc_per_us = 20_000_000 / (e0.elapsed_time(e1) * 1e3)
print(f"spin kernel: {cyc_per_us:.1f} cycles per us")
K = 150
for rnd in range(2):
    for g in gaps:
        for _ in range(10): smp.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            smp.step()
            if g: torch.cuda._sleep(int(g * cyc_per_us))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / K * 1e3
        print(f"round {rnd} gap {g:5d} us: {ms:7.3f} ms per step+gap  -> step alone {ms - g / 1e3:7.3f} ms", flush=True)
