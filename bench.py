#!/usr/bin/env python3
"""bench.py — latent volumes/sec of the conditional 3D U-Net DDPM sampler at 32^3 x 8ch, T=1000 on MI355X.

Contract (one JSON line from rank 0):  python bench.py --gpus N --steps K --warmup W
  * workload: BASELINE.json configs[2] per GPU — conditional_dm3d U-Net (widths 64/128/256), 32^3 x 8ch latents,
    B=32 volumes per GPU, T=1000 DDPM.  A "step" is one denoising step of the whole per-GPU batch: U-Net eps
    prediction + posterior update (+ Philox noise) — every one of the T steps of a chain does exactly this work, so
    value = volumes/sec for full T=1000 chains = N*B / (T * seconds_per_step); K steps of real chains are timed
    (x_T ~ N(0,1), t = T-1, T-2, ...), nothing is skipped inside a step.
  * N>1: one process per GPU, rank 0's weights broadcast once over RCCL, batch sharded, no per-step collective ("weak"
    scaling: B per GPU fixed).  Under torchrun (RANK/LOCAL_RANK/WORLD_SIZE set) this process is one rank; started plainly as
    `python bench.py --gpus N` it launches the N ranks itself as fresh child processes — before this process has touched the
    GPU — relays rank 0's JSON line and exits non-zero if any rank fails.
  * roofline: the dominant kernel is the k3/stride-1 Conv3d with the norm+SiLU prologue in its Winograd F(2,3)-along-x form,
    conv3d_igemm_h3w<1> (23 launches/step at B = 32: every Cin >= 32 conv of the 32^3 / 16^3 levels that reads a float32 tensor, and the
    Cin >= 256 convs of the 8^3 level with their two-way Cin split (the halves meet inside the launch), with or without a fused 1x1 skip tail; ~52 % of the step; persistent
    workgroups, one per CU); its MODE-2 twin behind a DM3D_FMT_H2 hand-off (10 launches, ~28 %), the direct kernel conv3d_igemm_h3v3
    (conv_in / conv_out, the UpSample parity convs, the first 8^3 conv, every k3 conv of small batches) and the stride-2 convs are listed
    under per_kernel_kind;
    achieved = algorithmic FLOPs (2*27*Cin*Cout*B*Dout^3 per launch, plus those of a fused 1x1 skip conv) /
    HIP-event time of those launches, measured live on the launch stream; peak = the dense MFMA peak of the datatype the
    kernel multiplies in (MI355X_MICROARCH.md): float16 2500 TFLOP/s in the default h3 mode (three v_mfma_f32_16x16x32_f16
    passes per product: executed_mfma_tflops counts what is issued — 40/54 of the direct form's k-steps in the Winograd-x form, a tenth of
    them against a zero pad tap; useful_mfma_tflops leaves the pad steps out), float32 157.3 TFLOP/s with
    --precision fp32 (v_mfma_f32_32x32x2_f32).  traffic = HBM bytes per launch from the committed rocprofv3 --pmc summary.
  * cpu_baseline: the CPU oracle (PyTorch-CPU restatement of the reference path; TensorFlow is not installed) on the
    host cores, a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_F16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA ~2.5 PF dense"
T_FULL = 1000
_T0 = time.time()


def log(msg):
    """Progress on stderr (stdout carries exactly one JSON line)."""
    print(f"[bench +{time.time() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def host_cores() -> int:
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (this parent never initialises the GPU),
    relay rank 0's stdout (the JSON line) and every rank's stderr.  All children are polled: when one exits non-zero its siblings
    are terminated at once (a rank that dies before the rendezvous would otherwise leave the others in init_process_group until
    the collective timeout) and the parent returns non-zero.  A port that another process grabbed between the probe and the
    children's bind makes the ranks exit with RC_RENDEZVOUS: one retry on a fresh port, for that failure only."""
    import socket
    import subprocess
    import threading

    def attempt():
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        procs = []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
        out0 = []
        reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
        reader.start()
        bad = []
        while True:
            rcs = [p.poll() for p in procs]
            bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
            if bad or all(rc is not None for rc in rcs):
                break
            time.sleep(0.2)
        if bad:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
        reader.join(timeout=10)
        if bad:
            print(f"[bench] ranks failed (rank, exit code): {bad}; siblings terminated", file=sys.stderr)
            return 1, [rc for _, rc in bad]
        sys.stdout.write(out0[0] if out0 else "")
        sys.stdout.flush()
        return 0, []

    # One retry, and only for the failure it exists for: a rank that could not bind / reach the rendezvous port (another process grabbed
    # it between the probe and the children's bind) exits with RC_RENDEZVOUS.  Any other failure — a signal, a GPU fault, an assertion —
    # is reported as it is: a second attempt would run on the box again and could publish a clean line over the first one's evidence.
    rc, codes = attempt()
    if rc != 0 and codes and all(c == RC_RENDEZVOUS or c == 0 for c in codes) and os.environ.get("DM3D_BENCH_NO_RETRY") != "1":
        print("[bench] rendezvous port was taken: one retry on a fresh port", file=sys.stderr)
        rc, codes = attempt()
    return rc


RC_RENDEZVOUS = 98      # exit code of a rank whose init_process_group failed to bind / connect (EADDRINUSE and kin)


def csrc_digest() -> str:
    """sha256 over the kernel sources: a PMC summary records the digest it was measured with (stale evidence is not attached)."""
    import hashlib
    d = os.path.join(ROOT, "3d-condtional-stable-diffusion_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def _restarting(smp):
    """step() of a sampler whose finished chain (t = 0 done) restarts from x_T, as the headline loop does: --steps may exceed T."""
    left = [T_FULL]

    def step():
        if left[0] == 0:
            smp.reset()
            left[0] = T_FULL
        smp.step()
        left[0] -= 1
    return step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="volumes per GPU (BASELINE configs[2]: 32)")
    ap.add_argument("--channels", type=int, default=8)
    ap.add_argument("--size", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--norm", choices=["batch", "group"], default="batch",
                    help="batch = the reference's inference BatchNormalization (folded); group = its commented-out "
                         "GroupNormalization(8) variant (per-sample statistics computed on the device every step)")
    ap.add_argument("--precision", choices=["h3", "fp32"], default="h3",
                    help="Conv3d arithmetic: h3 = float16 hi+lo split, 3 MFMA passes, fp32 accumulate (default); "
                         "fp32 = exact float32 MFMA")
    ap.add_argument("--no-fp32-mode", action="store_true", help="skip the extra exact-float32 timing (N=1, h3 runs only)")
    ap.add_argument("--no-full-chain", action="store_true", help="skip the wall-clock timing of one whole T=1000 generate() (N=1 only)")
    ap.add_argument("--print-csrc-digest", action="store_true")
    args = ap.parse_args()
    if args.print_csrc_digest:
        print(csrc_digest())
        return
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))              # nothing above has imported torch or touched the GPU

    import numpy as np
    import torch
    import torch.distributed as dist
    import dm3d_amd
    from dm3d_amd import _lib, parallel
    from dm3d_amd.networks import conditional_dm3d as cdm

    rank, local_rank, world = parallel.env_rank()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match the launcher's WORLD_SIZE={world}")
    # DM3D_BENCH_REHEARSAL=1: several ranks share GPU 0 and talk over gloo — only to rehearse the multi-process logic
    # on a one-GPU box (RCCL refuses two ranks on one device); the driver's real runs use one GPU per rank over RCCL.
    # DM3D_BENCH_REHEARSAL=cpu: the launch path only, at any width, without a GPU — self-launch, rendezvous, weight broadcast, digests,
    # gathers, barriers, max-over-ranks over gloo; the ranks build no model and time sleeps.  (The GPU boxes allow six GPU processes, so
    # the eight-rank width of the driver's scaling run can only be rehearsed this way; the line says so and is not a measurement.)
    rehearsal_mode = os.environ.get("DM3D_BENCH_REHEARSAL", "")
    rehearsal, dry = rehearsal_mode in ("1", "cpu"), rehearsal_mode == "cpu"
    dev_index = 0 if rehearsal else local_rank
    dev = None
    if not dry:
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
        _lib.require_device()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        tmo = datetime.timedelta(seconds=int(os.environ.get("DM3D_BENCH_INIT_TIMEOUT", "180")))   # a dead sibling must not hold the box for the default 10 min
        try:
            if rehearsal:
                dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
        except Exception as e:                  # (the parent retries on a fresh port for this exit code only)
            msg = str(e)
            if any(k in msg for k in ("EADDRINUSE", "Address already in use", "address already in use", "failed to bind", "Connection refused")):
                print(f"[bench] rank {rank}: rendezvous failed: {msg[:300]}", file=sys.stderr)
                sys.exit(RC_RENDEZVOUS)
            raise
    comm_dev = torch.device("cpu") if rehearsal else dev

    B, S, Cc = args.batch, args.size, args.channels
    log(f"rank {rank}/{world}: building weights and model (B={B}, {S}^3x{Cc})")
    cfg = dm3d_amd.UNetConfig(img_size=S, img_channels=Cc)
    spec = dm3d_amd.param_spec(cfg)
    # every rank builds the seeded weights itself (no GPU idles on rank 0's NumPy) and rank 0's copy is broadcast over them anyway:
    # the path a real deployment takes (rank 0 loads the checkpoint), and the digests below prove what each rank ended up with
    W = dm3d_amd.synthetic_weights(cfg, seed=0)
    W = parallel.broadcast_state(W if rank == 0 else None, spec, src=0, device=comm_dev)   # RCCL broadcast over xGMI (no-op at N=1)
    # every rank's identity: device, Philox seed, digest of the weights it holds after the broadcast (must all agree)
    # (PCI bus id / UUID of the device this rank computes on and the world size its process group reports: an N-rank line must show N
    # distinct devices)
    ident = {"pci_bus_id": None, "uuid": None}
    if not dry:
        pr = torch.cuda.get_device_properties(dev_index)
        if getattr(pr, "pci_bus_id", None) is not None:
            ident["pci_bus_id"] = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{getattr(pr, 'pci_device_id', 0):02x}"
        ident["uuid"] = str(getattr(pr, "uuid", "")) or None
    me = json.dumps({"rank": rank, "device": "none (cpu rehearsal)" if dry else torch.cuda.get_device_name(dev_index), "device_index": dev_index,
                     "pci_bus_id": ident["pci_bus_id"], "uuid": ident["uuid"], "pid": os.getpid(),
                     "world_size_seen": dist.get_world_size() if world > 1 else 1,
                     "seed": parallel.rank_seed(1234, rank), "weights_sha": parallel.state_digest(W)})
    ranks_info = {"world_size": dist.get_world_size() if world > 1 else 1,
                  "backend": dist.get_backend() if world > 1 else None,
                  "per_rank": [json.loads(x) for x in parallel.gather_strings(me)]}
    if len({r["weights_sha"] for r in ranks_info["per_rank"]}) != 1:
        raise SystemExit("ranks hold different weights after the broadcast")
    ids = [r["uuid"] or r["pci_bus_id"] for r in ranks_info["per_rank"]]
    ranks_info["distinct_devices"] = len(set(ids)) if all(ids) else None
    if not rehearsal and world > 1 and ranks_info["distinct_devices"] not in (None, world):
        raise SystemExit(f"{world} ranks but {ranks_info['distinct_devices']} distinct devices: {ids}")
    if dry:
        K = args.steps
        dist.barrier() if world > 1 else None
        t0 = time.perf_counter()
        for _ in range(K):
            time.sleep(0.002 * (1 + rank % 3))                 # ranks differ: the maximum over ranks is what gets reported
        if world > 1:
            dist.barrier()
        my_elapsed = time.perf_counter() - t0
        elapsed = parallel.max_over_ranks(my_elapsed, comm_dev)
        for r, ms_r in enumerate(parallel.gather_strings(f"{my_elapsed / K * 1e3:.4f}")):
            ranks_info["per_rank"][r]["ms_per_step"] = float(ms_r)
        if rank == 0:
            print(json.dumps({"metric": "NOT A MEASUREMENT: cpu rehearsal of the multi-rank launch path", "value": 0.0, "unit": "volumes/s",
                              "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
                              "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "none",
                              "config": {"workload": "no kernels run: self-launch, rendezvous, broadcast, gathers and barriers only",
                                         "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"batch-shard x{world} (CPU REHEARSAL, gloo)"},
                              "roofline": None, "cpu_baseline": None, "ranks": ranks_info}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    margs = SimpleNamespace(timesteps=T_FULL, num_gpus=world, kernel_resize=False, bs=B * world)
    model = cdm.DiffusionModel(S, 1024, Cc, None, margs, device=dev, weights=W, precision=args.precision, norm=args.norm)
    smp = model.sampler((B, S, S, S, Cc), context_value=1, seed=parallel.rank_seed(1234, rank),
                        use_graph=not args.no_graph)
    smp.prepare()                          # graph capture is setup, not a step: --warmup 0 then times replays only
    smp.reset()
    torch.cuda.synchronize()
    log("model prepared; warm-up")
    K, Wm = args.steps, args.warmup
    left = [T_FULL]                       # steps left in the current chain; a finished chain (t = 0 done) restarts from x_T

    def one_step():
        if left[0] == 0:
            smp.reset()
            left[0] = T_FULL
        smp.step()
        left[0] -= 1

    for _ in range(Wm):
        one_step()
        torch.cuda.synchronize()
        log("warm-up step done")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    my_elapsed = time.perf_counter() - t0
    elapsed = parallel.max_over_ranks(my_elapsed, comm_dev)
    s_per_step = elapsed / K
    for r, ms_r in enumerate(parallel.gather_strings(f"{my_elapsed / K * 1e3:.4f}")):       # the spread across GPUs
        ranks_info["per_rank"][r]["ms_per_step"] = float(ms_r)
    log(f"timed {K} steps: {s_per_step * 1e3:.2f} ms/step")
    value = world * B / (T_FULL * s_per_step)

    # ---- roofline of the dominant kernel, measured live with HIP events on the launch stream (rank 0) --------------
    roofline = None
    per_kind = {}
    if rank == 0:
        plan = smp.plan
        plan.run_timed()                                                   # warm
        reps = 2
        acc = {}
        for _ in range(reps):
            for kind, meta, ms in plan.run_timed():
                a = acc.setdefault(kind, [0, 0.0, 0.0, 0.0, 0.0, 0.0])
                a[0] += 1
                a[1] += ms
                a[2] += meta.get("flops", 0.0)
                a[3] += meta.get("bytes", 0.0)
                a[4] += meta.get("exec_flops", 0.0)
                a[5] += meta.get("useful_flops", meta.get("exec_flops", 0.0))
        for kind, (n, ms, fl, by, ex, us) in acc.items():
            per_kind[kind] = {"launches_per_step": n // reps, "ms_per_step": round(ms / reps, 4),
                              "tflops": round(fl / (ms * 1e-3) / 1e12, 2) if fl else None}
        # the dominant kernel: the 8-slice form where the grid is large enough for it (B = 32), else the 4-slice form (small batches)
        # (the kind with the most time per step among the k3 / stride-1 forms with the fused prologue; the float8 form wherever it runs)
        dom = max((k for k in ("conv_wino", "conv_k3s1", "conv_k3s1_td4") if k in acc), key=lambda k: acc[k][1])
        wide = dom == "conv_k3s1"
        n, ms, fl, by, ex, us = acc[dom]
        achieved = fl / (ms * 1e-3) / 1e12
        if dom == "conv_wino":
            kname = ("conv3d_igemm_h3w<1> (k3 stride-1 Conv3d with the fused norm+SiLU prologue as Winograd F(2,3) along x: 8x8x8 bricks, one wave "
                     "per SIMD with 256 accumulator registers, persistent workgroups (one per CU walks a list of bricks; the next brick's first image "
                     "and weight steps are staged during the last chunk of the current one), 40 k-steps per output pair instead of 54; float16 hi+lo "
                     "split, 3 x v_mfma_f32_16x16x32_f16 per transformed product, fp32 accumulate; a fused 1x1 skip conv runs as a register-direct tail on the "
                     "transformed tiles; channel-quad column mapping: a lane's four column tiles are four consecutive channels, 16-byte stores "
                     "without a transpose; launches with Cin < 32, partial bricks or grids too small even with the two-way Cin split stay on the direct "
                     "kernel conv3d_igemm_h3v3, listed as conv_k3s1_td4 / conv_k3s1_n32 / conv_up)")
            peak, passes = PEAK_F16_MFMA_TFLOPS, round(3 * 40 / 54, 3)
        elif args.precision == "h3":
            kname = ("conv3d_igemm_h3v3<3, 1, 8, 4> (k3 stride-1 Conv3d with the fused norm+SiLU prologue, 8-slice bricks, free-running "
                     "software-pipelined waves; float16 hi+lo split, 3 x v_mfma_f32_16x16x32_f16 per algorithmic product, fp32 accumulate; "
                     "its 4-slice form <3, 1, 4, 4> is listed as conv_k3s1_td4, the MODE-2 twin that reads pre-activated DM3D_FMT_H2 input as "
                     "conv_k3s1_h2in)") if wide else (
                     "conv3d_igemm_h3v3<3, 1, 4, 4> (k3 stride-1 Conv3d with the fused norm+SiLU prologue, 4-slice bricks: this batch is "
                     "too small for the 8-slice form; float16 hi+lo split, 3 x v_mfma_f32_16x16x32_f16, fp32 accumulate)")
            peak, passes = PEAK_F16_MFMA_TFLOPS, 3
        else:
            kname, peak, passes = "conv3d_igemm_f32<4, 8, 8, 1, 3, 4, 1> (k3 stride-1 Conv3d, v_mfma_f32_32x32x2_f32)", PEAK_FP32_MFMA_TFLOPS, 1
        # the WHOLE conv set of the step (every kind that is a Conv3D launch): sum of algorithmic FLOPs / sum of time / peak
        conv_kinds = [k for k in acc if k.startswith("conv")]
        conv_fl, conv_ms = sum(acc[k][2] for k in conv_kinds), sum(acc[k][1] for k in conv_kinds)
        conv_ex, conv_us = sum(acc[k][4] for k in conv_kinds), sum(acc[k][5] for k in conv_kinds)
        conv_all = {"kinds": sorted(conv_kinds), "ms_per_step": round(conv_ms / reps, 4),
                    "algorithmic_tflops": round(conv_fl / (conv_ms * 1e-3) / 1e12, 2),
                    "algorithmic_frac_of_peak": round(conv_fl / (conv_ms * 1e-3) / 1e12 / peak, 4),
                    "executed_mfma_frac_of_peak": round(conv_ex / (conv_ms * 1e-3) / 1e12 / peak, 4),
                    "useful_mfma_frac_of_peak": round(conv_us / (conv_ms * 1e-3) / 1e12 / peak, 4)}
        # the second kernel of the step (MODE-2 twin behind a DM3D_FMT_H2 hand-off, with its fused skip tails): its own algorithmic figures
        secondary = None
        if dom == "conv_wino" and "conv_wino_h2in" in acc:
            n2, ms2, fl2, by2, ex2, us2 = acc["conv_wino_h2in"]
            secondary = {"kernel": "conv3d_igemm_h3w<2> (the same kernel reading a pre-activated, pre-split DM3D_FMT_H2 tensor: ResidualBlock conv2 "
                                   "behind the conv1 hand-off, incl. its fused 1x1 skip tails)",
                         "achieved": round(fl2 / (ms2 * 1e-3) / 1e12, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(fl2 / (ms2 * 1e-3) / 1e12 / peak, 4), "avg_launch_ms": round(ms2 / n2, 4), "launches_per_step": n2 // reps,
                         "algorithmic_gflop_per_launch": round(fl2 / n2 / 1e9, 2), "algorithmic_mb_per_launch": round(by2 / n2 / 1e6, 2),
                         "useful_mfma_frac_of_peak": round(us2 / (ms2 * 1e-3) / 1e12 / peak, 4), "traffic": None}
        roofline = {"bound": "mfma", "kernel": kname,
                    "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": None,
                    "mfma_passes_per_product": passes,
                    "executed_mfma_tflops": round(ex / (ms * 1e-3) / 1e12, 2),
                    "executed_mfma_frac_of_peak": round(ex / (ms * 1e-3) / 1e12 / peak, 4),
                    "useful_mfma_tflops": round(us / (ms * 1e-3) / 1e12, 2),
                    "useful_mfma_frac_of_peak": round(us / (ms * 1e-3) / 1e12 / peak, 4),
                    "note": "achieved = algorithmic FLOPs (2*27*Cin*Cout per output voxel) / time; executed counts the MFMA "
                            "work issued: x3 passes in h3, 40/54 of the k-steps in the Winograd-x form (36 + the zero pad tap; useful leaves "
                            "the pad steps out: 36/54), and the two UpSample convs run as 8-tap parity convs (8/27 of the taps)",
                    "achieved_vs_fp32_mfma_peak": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                    "avg_launch_ms": round(ms / n, 4), "launches_per_step": n // reps,
                    "algorithmic_gflop_per_launch": round(fl / n / 1e9, 2),
                    "algorithmic_mb_per_launch": round(by / n / 1e6, 2),
                    "all_convs": conv_all, "secondary": secondary}

    # HBM traffic of the dominant kernel: PMC counters need their own rocprofv3 passes (MI355X_MICROARCH.md), so the figure
    # comes from the committed summary of those passes over this same command (profiles/summarize_pmc.py), not from this run.
    # It is attached only when the summary was measured on THIS workload with THESE kernel sources (its header records both);
    # otherwise traffic stays null and the reason is reported.
    if roofline is not None:
        import csv
        import glob
        want = ("conv3d_igemm_h3w<1>" if roofline["kernel"].startswith("conv3d_igemm_h3w") else
             "conv3d_igemm_h3v3<3, 1, 8, 4>" if "conv_k3s1" in per_kind else "conv3d_igemm_h3v3<3, 1, 4, 4>") if args.precision != "fp32" \
            else "conv3d_igemm_f32<4, 8, 8, 1, 3, 4, 1>"
        sig = f"batch={B} size={S} channels={Cc} norm={args.norm} precision={args.precision} csrc={csrc_digest()}"
        reason = "no profiles/*_pmc_hbm.csv records this workload and these kernel sources: " + sig
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm.csv")), reverse=True):
            lines = open(path).read().splitlines()
            if not any(l.startswith("# workload: ") and l[len("# workload: "):].strip() == sig for l in lines):
                continue
            rows = [r for r in csv.reader(l for l in lines if not l.startswith("#")) if r and r[0] == want]
            if not rows:
                reason = f"{os.path.basename(path)} matches the workload but has no row for {want}"
                log(reason)
                continue
            roofline["traffic"] = float(rows[0][4]) * 1e6
            if roofline.get("secondary"):
                rows2 = [r for r in csv.reader(l for l in lines if not l.startswith("#")) if r and r[0] == "conv3d_igemm_h3w<2>"]
                if rows2:
                    sec = roofline["secondary"]
                    sec["traffic"] = float(rows2[0][4]) * 1e6
                    sec["traffic_over_algorithmic"] = round(sec["traffic"] / (sec["algorithmic_mb_per_launch"] * 1e6), 3) if sec["algorithmic_mb_per_launch"] else None
            roofline["traffic_over_algorithmic"] = round(roofline["traffic"] / (roofline["algorithmic_mb_per_launch"] * 1e6), 3) if roofline["algorithmic_mb_per_launch"] else None
            roofline["traffic_unit"] = "bytes/launch (2*FETCH_SIZE+WRITE_SIZE, rocprofv3 --pmc, " + os.path.basename(path) + ")"
            roofline["hbm_GBps_of_kernel"] = round(roofline["traffic"] / (roofline["avg_launch_ms"] * 1e-3) / 1e9, 1)
            roofline["hbm_frac_of_8TBps"] = round(roofline["hbm_GBps_of_kernel"] / 8000.0, 4)
            reason = None
            break
        if reason:
            roofline["traffic_note"] = reason
            log("roofline.traffic = null: " + reason)
        # matrix-pipe occupancy (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), rocprofv3 --pmc passes over this same command,
        # profiles/summarize_sq.py) and the in-kernel shader clock (tools/kernel_clock.py): attached, like traffic, only from committed
        # summaries whose "# workload:" line records this workload / these kernel sources
        roofline["mfma_busy_frac"], roofline["clock_ghz"] = None, None
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq.csv")), reverse=True):
            lines = open(path).read().splitlines()
            if not any(l.startswith("# workload: ") and l[len("# workload: "):].strip() == sig for l in lines):
                continue
            body = [l for l in lines if not l.startswith("#")]
            rows = list(csv.DictReader(body))
            hit = [r for r in rows if r["kernel"] == want]
            if hit:
                roofline["mfma_busy_frac"] = float(hit[0]["mfma_busy_frac"])
                roofline["mfma_busy_note"] = ("SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs) of " + want + ", whole launch incl. prologue / "
                                              "epilogue, profiled passes (" + os.path.basename(path) + "); valu_per_mfma " + hit[0]["valu_per_mfma"] +
                                              ", lds_conflict " + hit[0]["lds_conflict"] + ", wait_frac " + hit[0]["wait_frac"] + ", stall_frac " + hit[0]["stall_frac"])
                break
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_clock.csv")), reverse=True):
            lines = open(path).read().splitlines()
            if not any(l.startswith("# workload: ") and l.strip().endswith("csrc=" + csrc_digest()) for l in lines):
                continue
            is_wino = any(l.startswith("# workload: kernel_clock conv wino") for l in lines)
            if is_wino != roofline["kernel"].startswith("conv3d_igemm_h3w"):
                continue                                # the clock of another kernel (profile_round.sh writes one file per conv form)
            rows = list(csv.DictReader(l for l in lines if not l.startswith("#")))
            if rows and is_wino:
                vals = sorted(float(r["clock_ghz_median"]) for r in rows)
                roofline["clock_ghz"] = vals[len(vals) // 2]
                roofline["clock_note"] = ("in-kernel shader clock of the Winograd-x conv kernel under sustained load on random data (median of " +
                                          ", ".join(f'{r["shape"]}: {r["clock_ghz_median"]}' for r in rows) + " GHz; matrix-pipe duty inside the chunk loop " +
                                          ", ".join(r["mfma_duty_in_loop"] for r in rows) + "; " + os.path.basename(path) +
                                          "); one wave per SIMD: in cycles this kernel is bound by what a single wave can issue beside its MFMAs (the same "
                                          "duty on all-zero operands at 2.38 GHz); the clock those cycles run at is set by the chip's power management "
                                          "(socket power 1.3-1.35 kW under the sustained chain: profiles/r03_clocks_under_load.log; DESIGN.md section 4)")
                break
            if rows:
                vals = sorted(float(r["clock_ghz_median"]) for r in rows)
                roofline["clock_ghz"] = vals[len(vals) // 2]
                roofline["clock_note"] = ("in-kernel shader clock of the k3 conv kernel under sustained load on random data (median of " +
                                          ", ".join(f'{r["shape"]}: {r["clock_ghz_median"]}' for r in rows) + " GHz; matrix-pipe duty inside the chunk loop " +
                                          ", ".join(r["mfma_duty_in_loop"] for r in rows) + "; " + os.path.basename(path) +
                                          "); the same instruction stream on all-zero operands holds ~2.39 GHz: the kernel is bound by the chip's power "
                                          "management, not by its instruction schedule (DESIGN.md section 4)")
                break

    # ---- the same K steps in exact-float32 arithmetic (v_mfma_f32_32x32x2_f32), so that number is timed in this run too ------
    fp32_mode = None
    if rank == 0 and world == 1 and args.precision == "h3" and not args.no_fp32_mode:
        log("fp32 mode: building the exact-float32 model")
        m32 = cdm.DiffusionModel(S, 1024, Cc, None, margs, device=dev, weights=W, precision="fp32", norm=args.norm)
        s32 = m32.sampler((B, S, S, S, Cc), context_value=1, seed=parallel.rank_seed(1234, rank), use_graph=not args.no_graph)
        s32.prepare()
        s32.reset()
        step32 = _restarting(s32)
        for _ in range(max(1, Wm)):
            step32()
        torch.cuda.synchronize()
        t32 = time.perf_counter()
        for _ in range(K):
            step32()
        torch.cuda.synchronize()
        sp32 = (time.perf_counter() - t32) / K
        acc32 = [0, 0.0, 0.0]
        s32.plan.run_timed()
        for kind, meta, ms in s32.plan.run_timed():
            if kind == "conv_k3s1":
                acc32[0] += 1
                acc32[1] += ms
                acc32[2] += meta.get("flops", 0.0)
        a32 = acc32[2] / (acc32[1] * 1e-3) / 1e12
        fp32_mode = {"ms_per_step": sp32 * 1e3, "value": B / (T_FULL * sp32), "unit": "volumes/s", "steps": K,
                     "roofline": {"bound": "mfma", "kernel": "conv3d_igemm_f32<4, 8, 8, 1, 3, 4, 1> (v_mfma_f32_32x32x2_f32)",
                                  "achieved": round(a32, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                  "frac": round(a32 / PEAK_FP32_MFMA_TFLOPS, 4), "avg_launch_ms": round(acc32[1] / acc32[0], 4),
                                  "launches_per_step": acc32[0]}}
        log(f"fp32 mode: {sp32 * 1e3:.2f} ms/step")
        del s32, m32
        torch.cuda.empty_cache()

    # ---- one whole chain, wall clock: generate() of B volumes through all T steps (the K-step figure extrapolates to this) ------
    full_chain = None
    if rank == 0 and world == 1 and not args.no_full_chain:
        torch.cuda.synchronize()
        f0 = time.perf_counter()
        out = model.generate((B, S, S, S, Cc), context_value=1, seed=parallel.rank_seed(1234, rank), use_graph=not args.no_graph)
        torch.cuda.synchronize()
        fs = time.perf_counter() - f0
        full_chain = {"full_chain_s": fs, "volumes": B, "timesteps": T_FULL, "volumes_per_s": B / fs,
                      "finite": bool(torch.isfinite(out).all().item()),
                      "ratio_to_extrapolated": fs / (T_FULL * s_per_step)}
        log(f"full T={T_FULL} chain of B={B}: {fs:.2f} s ({B / fs:.3f} volumes/s; {T_FULL}*ms_per_step = {T_FULL * s_per_step:.2f} s)")
        del out

    # ---- CPU baseline: the oracle on the host cores, bounded sample (rank 0, N=1 only) ----------------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ref_torch as rt
        cores = host_cores()
        log(f"cpu baseline on {cores} threads (os.cpu_count()={os.cpu_count()})")
        torch.set_num_threads(cores)
        ocfg = rt.UNetConfig(img_size=S, img_channels=Cc)
        Wt = {k: torch.from_numpy(v) for k, v in W.items()}
        cb, csteps = 8, 10                 # ~10-20 s of host work: a bounded sample of the same workload
        g = torch.Generator().manual_seed(1234)
        x = torch.randn(cb, S, S, S, Cc, generator=g)
        noises = {i: torch.randn(cb, S, S, S, Cc, generator=g) for i in range(T_FULL - csteps - 1, T_FULL)}
        betas = rt.Betas(T_FULL)
        ctx = torch.tensor([[[1]]])
        tt = torch.full((cb,), T_FULL - 1, dtype=torch.int64)
        rt.unet_forward(Wt, ocfg, x, tt, ctx)                             # warm-up (oneDNN primitive creation)
        c0 = time.perf_counter()
        for i in range(T_FULL - 1, T_FULL - 1 - csteps, -1):
            tt = torch.full((cb,), i, dtype=torch.int64)
            x = rt.ddpm_step(betas, x, rt.unet_forward(Wt, ocfg, x, tt, ctx), tt, noises[i])
            log(f"cpu step t={i} done")
        c_step = (time.perf_counter() - c0) / csteps
        cpu = {"value": cb / (T_FULL * c_step), "unit": "volumes/s", "cores": cores, "kind": "port",
               "sample": f"oracle/ref_torch.py (PyTorch-CPU fp32, oneDNN conv3d), B={cb}, {csteps} denoising steps of the "
                         f"T=1000 chain at {S}^3x{Cc}ch, extrapolated x{T_FULL}/{csteps} (steps are identical work); "
                         f"{c_step:.2f} s/step"}

    if rank == 0:
        line = {
            "metric": "latent volumes/sec at 32^3x8ch T=1000 DDPM", "value": value, "unit": "volumes/s",
            "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": s_per_step * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "f32 (f16 hi+lo x3 MFMA, f32 accumulate)", "data": "synthetic",
            "config": {"workload": f"conditional_dm3d U-Net (widths 64/128/256) DDPM sampling, {S}^3x{Cc}ch latents, "
                                   f"B={B} volumes per GPU, T={T_FULL}; step = one denoising step of the batch "
                                   f"(U-Net eps + posterior update + Philox noise, HIP-graph replay); "
                                   f"value = n_gpus*B/(T*s_per_step)",
                       "batch_per_gpu": B, "global_batch": B * world, "timesteps": T_FULL,
                       "weights": "seeded synthetic (seed 0), rank-0 broadcast", "parallelism": f"batch-shard x{world}" + (" (REHEARSAL: ranks share one GPU, gloo)" if rehearsal else "")},
            "roofline": roofline, "cpu_baseline": cpu, "fp32_mode": fp32_mode, "full_chain": full_chain,
            "per_kernel_kind": per_kind,
            "per_kernel_kind_note": "eager launches with a HIP-event pair around each (event overhead included; the timed step is "
                                    "a HIP-graph replay, so these rows sum to slightly more than ms_per_step)",
            "ranks": ranks_info,
            # MFMA utilisation of the Conv3d set, three accountings (all against the dense peak of the MFMA datatype used):
            #   executed  = every MFMA issued (h3: three passes per product; the Winograd form's zero pad steps included)
            #   useful    = executed minus the pad steps
            #   algorithmic = SURVEY 8(d): 2*27*Cin*Cout FLOPs per output voxel / time — what BASELINE's "MFMA util %" is judged on
            # "dominant" = the roofline kernel alone, "all" = every conv launch of the step
            "conv_mfma_executed_pct": None if roofline is None else {"dominant": round(100 * roofline["executed_mfma_frac_of_peak"], 2),
                                                                     "all": round(100 * roofline["all_convs"]["executed_mfma_frac_of_peak"], 2)},
            "conv_mfma_useful_pct": None if roofline is None else {"dominant": round(100 * roofline["useful_mfma_frac_of_peak"], 2),
                                                                   "all": round(100 * roofline["all_convs"]["useful_mfma_frac_of_peak"], 2)},
            "conv_algorithmic_pct_of_peak": None if roofline is None else {"dominant": round(100 * roofline["frac"], 2),
                                                                           "all": round(100 * roofline["all_convs"]["algorithmic_frac_of_peak"], 2)},
            "conv_algorithmic_frac_of_peak": None if roofline is None else roofline["all_convs"]["algorithmic_frac_of_peak"],
            "precision": args.precision, "norm": args.norm,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()                      # ranks != 0 wait here while rank 0 finishes its roofline leg: nobody tears the group down early
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
