"""CPU tier: host logic of the product and the C-ABI surface (no kernel is launched without a GPU)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(built_library):
    header = open(os.path.join(ROOT, "include", "dm3d.h")).read()
    declared = set(re.findall(r"\b(dm3d_[a-z0-9_]+)\s*\(", header))
    declared -= {"dm3d_pack_weights output"}
    handle = ctypes.CDLL(built_library)
    for name in sorted(declared):
        assert hasattr(handle, name), f"{name} declared in include/dm3d.h but not exported"
    from dm3d_amd import _lib
    assert set(_lib.SIGNATURES) == declared
    assert _lib.lib().dm3d_version() == _lib.ABI_VERSION == 111
    assert _lib.lib().dm3d_packed_weight_elems(27, 96, 64) == 27 * 64 * 96
    assert _lib.lib().dm3d_packed_weight_elems(1, 8, 8) == 64 * 16


def test_struct_layouts_match_the_header(built_library, tmp_path):
    """sizeof() of the descriptor structs as the C compiler sees them == the ctypes mirrors."""
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "dm3d.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(dm3d_conv_desc),'
                   ' sizeof(dm3d_gemm_desc), sizeof(dm3d_ddpm_desc), sizeof(dm3d_attention_desc), sizeof(dm3d_mlp_desc), sizeof(dm3d_attn_front_desc));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    sizes = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    from dm3d_amd import _lib
    assert sizes == [ctypes.sizeof(_lib.ConvDesc), ctypes.sizeof(_lib.GemmDesc), ctypes.sizeof(_lib.DdpmDesc),
                     ctypes.sizeof(_lib.AttentionDesc), ctypes.sizeof(_lib.MlpDesc), ctypes.sizeof(_lib.AttnFrontDesc)]


def test_plain_c_program_links_against_the_abi(built_library, tmp_path):
    """The boundary is a C ABI: a C99 translation unit including only include/dm3d.h compiles with gcc, links against the shared
    library and calls the entries that need no device (version, size queries, layout query, argument validation)."""
    src = tmp_path / "abi.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "dm3d.h"
int main(void) {
    dm3d_conv_desc d;
    memset(&d, 0, sizeof d);
    int rc = dm3d_conv3d_ndhwc(&d, NULL);                 /* all-null descriptor: refused before any device call */
    printf("%d %d %lld %lld %d %s\n", dm3d_version() > 0, dm3d_conv_weight_layout(3, 2, 0, 0, 64),
           (long long)dm3d_packed_weight_h3_bytes(27, 64, 64), (long long)dm3d_attention_workspace_bytes(2, 64, 48), rc != 0,
           dm3d_last_error());
    return 0;
}
''')
    exe = tmp_path / "abi"
    libdir = os.path.dirname(built_library)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", libdir, "-ldm3d_hip", f"-Wl,-rpath,{libdir}"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    f = out.split(None, 5)
    assert f[0] == "1" and f[1] == "0" and int(f[2]) == 27 * 64 * 4 * 64 * 2 // 2 * 1 and int(f[3]) == 2 * 64 * 48 * 4 and f[4] == "1"
    assert "conv" in f[5]


def test_stamp_variant_tool_still_patches_the_kernels(built_library):
    """tools/mk_stamp_variants.py builds the diagnostic libraries (conv: -DDM3D_CLOCK_STAMPS; GEMM: -DDM3D_GEMM_STAMPS; no source patching)."""
    import shutil
    import sys
    vdir = os.path.join(os.path.dirname(built_library), "variants")
    try:
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mk_stamp_variants.py")], check=True, capture_output=True, cwd=ROOT)
        assert os.path.exists(os.path.join(vdir, "cck.so")) and os.path.exists(os.path.join(vdir, "gst.so"))
    finally:
        shutil.rmtree(vdir, ignore_errors=True)


def test_argument_validation_needs_no_gpu(built_library):
    """Bad descriptors are rejected before any launch, with a readable message."""
    from dm3d_amd import _lib
    lib = _lib.lib()
    d = _lib.ConvDesc()
    assert lib.dm3d_conv3d_ndhwc(ctypes.byref(d), None) == -1
    assert b"non-null" in lib.dm3d_last_error()
    g = _lib.GemmDesc()
    g.a = g.b = g.out = 16
    g.m, g.n, g.k, g.batch, g.lda, g.ldb, g.ldo = 4, 4, 6, 1, 6, 6, 4
    assert lib.dm3d_gemm_tn(ctypes.byref(g), None) == -1 and b"multiples of 4" in lib.dm3d_last_error()
    assert lib.dm3d_softmax_rows(None, 1, 1, 1, None) == -1
    with pytest.raises(_lib.Dm3dError):
        _lib.check(-1, "x")


def test_product_fails_loudly_without_a_device(built_library):
    from dm3d_amd import _lib
    if _lib.lib().dm3d_device_ok():
        pytest.skip("a gfx950 device is present")
    import dm3d_amd
    from dm3d_amd.unet import UNet
    cfg = dm3d_amd.UNetConfig(img_size=4, img_channels=4, widths=(16, 32), has_attention=(False, True), first_conv_channels=16)
    net = UNet(cfg, device="cpu")
    with pytest.raises(_lib.Dm3dError):
        net([torch.zeros(1, 4, 4, 4, 4), torch.tensor([0]), torch.tensor([[[0]]])])


def test_spec_and_synthetic_weights_equal_the_oracles():
    import dm3d_amd
    from oracle import ref_torch as rt
    for cond, S, Cc in ((True, 8, 4), (False, 16, 4), (True, 32, 8)):
        cfg = dm3d_amd.UNetConfig(img_size=S, img_channels=Cc, conditional=cond)
        ocfg = rt.UNetConfig(img_size=S, img_channels=Cc, conditional=cond)
        spec, ospec = dm3d_amd.param_spec(cfg), rt.param_spec(ocfg)
        assert list(spec.items()) == list(ospec.items())
    cfg = dm3d_amd.UNetConfig(img_size=8, img_channels=4)
    W, OW = dm3d_amd.synthetic_weights(cfg, 0), rt.synthetic_weights(rt.UNetConfig(img_size=8, img_channels=4), 0)
    assert all(np.array_equal(W[k], OW[k].numpy()) for k in W)


def test_kernel_init_matches_variance_scaling():
    import dm3d_amd
    rng = np.random.default_rng(0)
    w = dm3d_amd.kernel_init(1.0)((3, 3, 3, 64, 128), rng)
    lim = np.sqrt(3.0 / ((27 * 64 + 27 * 128) / 2))
    assert w.dtype == np.float32 and abs(np.abs(w).max() - lim) / lim < 1e-3
    assert abs(w.var() - lim ** 2 / 3) / (lim ** 2 / 3) < 0.02
    z = dm3d_amd.kernel_init(0.0)((64, 64), rng)
    assert np.abs(z).max() < 1e-5 and np.abs(z).max() > 0           # scale clamps to 1e-10, not 0
    W = dm3d_amd.keras_init_weights(dm3d_amd.UNetConfig(img_size=4, img_channels=4), 0)
    assert np.abs(W["out.conv.kernel"]).max() < 1e-5 and np.abs(W["down0.res0.conv2.kernel"]).max() < 1e-5
    assert np.all(W["out.norm.gamma"] == 1) and np.all(W["out.conv.bias"] == 0)


def test_betas_and_time_table_equal_the_oracles():
    import dm3d_amd
    from oracle import ref_torch as rt
    for T in (5, 50, 1000):
        b, ob = dm3d_amd.Betas(T), rt.Betas(T)
        for n in ob.NAMES:
            assert np.array_equal(getattr(b, n), getattr(ob, n).numpy())
    t = np.array([0, 1, 7, 500, 999])
    a, o = dm3d_amd.time_embedding_table(t, 128), rt.time_embedding(torch.from_numpy(t), 128).numpy()
    assert np.abs(a - o).max() < 2e-6                                # same fp32 op order; libm vs sleef sin/cos ulps


def test_drop_in_signatures():
    import inspect
    from dm3d_amd.networks import conditional_dm3d, dm3d
    p = list(inspect.signature(conditional_dm3d.build_model).parameters)
    assert p[:10] == ["img_size", "img_channels", "widths", "has_attention", "has_cross_attention", "num_res_blocks",
                      "norm_groups", "interpolation", "activation_fn", "context_dim"]
    assert inspect.signature(conditional_dm3d.build_model).parameters["context_dim"].default == 1
    assert inspect.signature(dm3d.build_model).parameters["context_dim"].default is None
    p = list(inspect.signature(conditional_dm3d.DiffusionModel.__init__).parameters)
    assert p[1:6] == ["latent_size", "num_embed", "latent_channels", "vqvae_load_ckpt", "args"]
    g = inspect.signature(conditional_dm3d.DiffusionModel.generate).parameters
    assert list(g)[1:4] == ["shape", "last_step", "context_value"] and g["shape"].default == (1, 16, 16, 16, 16)
    assert list(inspect.signature(conditional_dm3d.DiffusionModel.sample).parameters)[1:] == \
        ["x_t", "pred_noise", "curr_time_step", "shape"]
    ts = inspect.signature(conditional_dm3d.DiffusionModel.train_step).parameters
    assert [n for n, q in ts.items() if q.kind == q.POSITIONAL_OR_KEYWORD][1:] == ["inputs"]
    assert {n for n, q in ts.items() if q.kind == q.KEYWORD_ONLY} == {"t", "noise", "latents"}          # extensions are keyword-only
    assert conditional_dm3d.first_conv_channels == 32 and dm3d.first_conv_channels == 64
    with pytest.raises(ValueError):
        conditional_dm3d.build_model(8, 4, [64, 128, 256], [False, False, True], has_cross_attention=[True], context_dim=0)
    m = conditional_dm3d.DiffusionModel(8, 1024, 4, None, type("A", (), dict(timesteps=7, num_gpus=1, kernel_resize=False, bs=3))(),
                                        device="cpu")
    assert m.timesteps == 7 and m.lc == 4 and m.global_bs == 3 and m.b.beta.shape == (7,) and m.metrics[0].name == "loss"
    assert m.network.cfg.widths == (64, 128, 256) and m._vqvae is None      # the VQ-VAE bracket is built lazily
    with pytest.raises(ValueError, match="images"):
        m.train_step((None, None, None))
    from dm3d_amd import _lib
    with pytest.raises(_lib.Dm3dError):                       # no device here: the training path has no CPU fallback either
        m.train_step((None, None, [[[1]]]), latents=np.zeros((1, 8, 8, 8, 4), np.float32), t=[0], noise=np.zeros((1, 8, 8, 8, 4), np.float32))


def test_walk_block_order_and_plan_shapes():
    import dm3d_amd
    from dm3d_amd.weights import walk
    blocks, _ = walk(dm3d_amd.UNetConfig(img_size=32, img_channels=8))
    res = [b for b in blocks if b.kind == "res"]
    assert len(res) == 17 and [b.cin + b.cskip for b in res[8:]] == [512, 512, 384, 384, 256, 192, 192, 128, 96]
    assert [b.name for b in blocks if b.kind == "attn"] == ["down2.attn0", "down2.attn1", "mid.attn", "up2.attn0",
                                                            "up2.attn1", "up2.attn2"]
    assert sum(b.kind == "push" for b in blocks) == 8 and [b.edge for b in blocks if b.kind in ("down", "up")] == [16, 8, 16, 32]


def test_shard_and_seed_helpers():
    from dm3d_amd import parallel
    for total, world in ((256, 8), (10, 4), (3, 8)):
        got = [parallel.shard_range(total, r, world) for r in range(world)]
        assert got[0][0] == 0 and got[-1][1] == total and all(a[1] == b[0] for a, b in zip(got, got[1:]))
        assert max(h - l for l, h in got) - min(h - l for l, h in got) <= 1
    assert parallel.rank_seed(1234, 3) == 1237
    with pytest.raises(ValueError):
        parallel.shard_range(4, 4, 4)


_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
import dm3d_amd
from dm3d_amd import parallel
rank, world = int(sys.argv[2]), 2
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[3], RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank))
dist.init_process_group("gloo", rank=rank, world_size=world)
cfg = dm3d_amd.UNetConfig(img_size=4, img_channels=4, widths=(16, 32), has_attention=(False, True), first_conv_channels=16)
spec = dm3d_amd.param_spec(cfg)
W = dm3d_amd.synthetic_weights(cfg, seed=42) if rank == 0 else None
got = parallel.broadcast_state(W, spec, src=0)
ref = dm3d_amd.synthetic_weights(cfg, seed=42)
assert list(got) == list(spec) and all(np.array_equal(got[k], ref[k]) for k in ref)
assert parallel.env_rank() == (rank, rank, 2)
lo, hi = parallel.shard_range(7, rank, world)
assert (lo, hi) == ((0, 4) if rank == 0 else (4, 7))
assert parallel.max_over_ranks(1.0 + rank) == 2.0
assert parallel.gather_strings(f"r{rank}") == ["r0", "r1"]
assert parallel.state_digest(got) == parallel.state_digest(ref) != parallel.state_digest({**ref, "x": np.ones(1, np.float32)})

class FakeModel:                       # records what generate_sharded asks of DiffusionModel.generate (no GPU here)
    device = torch.device("cpu")
    def generate(self, shape, last_step=0, context_value=None, *, seed=None, **kw):
        self.call = (tuple(shape), last_step, None if context_value is None else np.asarray(context_value).reshape(-1).tolist(), seed, kw)
        out = torch.empty(shape, dtype=torch.float32)
        for i in range(shape[0]):
            out[i] = 1000 * seed + i + 0.5 * float(np.asarray(context_value).reshape(-1)[i if np.asarray(context_value).size > 1 else 0])
        return out

m = FakeModel()
full = parallel.generate_sharded(m, (5, 2, 2, 2, 4), 0, [0, 1, 0, 1, 1], seed=7, use_graph=False)
lo, hi = parallel.shard_range(5, rank, world)
assert m.call == ((hi - lo, 2, 2, 2, 4), 0, [0, 1, 0, 1, 1][lo:hi], 7 + rank, {"use_graph": False})
want = torch.tensor([7000.0, 7001.5, 7002.0, 8000.5, 8001.5])          # rank 0: volumes 0-2 (key 7), rank 1: volumes 3-4 (key 8)
assert full.shape == (5, 2, 2, 2, 4) and torch.equal(full[:, 0, 0, 0, 0], want)
local = parallel.generate_sharded(m, (5, 2, 2, 2, 4), 0, 1, seed=7, gather=False)
assert local.shape[0] == hi - lo and m.call[2] == [1]
one = parallel.generate_sharded(m, (1, 2, 2, 2, 4), 0, 0, seed=3)               # fewer volumes than ranks: rank 1's shard is empty
assert one.shape[0] == 1 and float(one[0, 0, 0, 0, 0]) == 3000.0
try:
    parallel.generate_sharded(m, (5, 2, 2, 2, 4), 0, [0, 1], seed=7)
    raise SystemExit("a context list of the wrong length was accepted")
except ValueError:
    pass
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
"""


def test_weight_broadcast_and_sharded_generate_world_size_2_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    port = str(29500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), port], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "ok 0" in outs[0] and "ok 1" in outs[1]


def test_bench_self_launch_fails_fast_when_a_rank_dies(built_library):
    """`python bench.py --gpus 2` started plainly supervises its ranks: on a box without a GPU every rank dies in require_device(); the
    parent must notice, terminate what is left and return non-zero at once instead of waiting out a collective timeout."""
    import time
    if torch.cuda.is_available():
        pytest.skip("needs a box where the ranks fail: no GPU")
    env = dict(os.environ, DM3D_BENCH_NO_RETRY="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "ranks failed" in r.stderr, r.stderr[-2000:]
    assert r.stdout.strip() == "" and time.time() - t0 < 120


def test_bench_launch_path_at_eight_ranks_without_a_gpu():
    """The width the driver's scaling run uses: `python bench.py --gpus 8` self-launches eight rank processes, which rendezvous over gloo,
    broadcast rank 0's weights, compare digests, gather per-rank records, time between barriers and reduce the maximum — in
    DM3D_BENCH_REHEARSAL=cpu mode, where no rank touches a GPU (the GPU boxes admit six GPU processes; tests/test_gpu_round4.py rehearses four
    ranks on the card).  The line says what it is; a rank that dies takes its siblings down and the parent fails."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, DM3D_BENCH_REHEARSAL="cpu")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--batch", "2",
                        "--size", "8", "--channels", "4"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and line["metric"].startswith("NOT A MEASUREMENT") and line["value"] == 0.0
    ranks = line["ranks"]
    assert ranks["world_size"] == 8 and ranks["backend"] == "gloo"
    assert [p["seed"] for p in ranks["per_rank"]] == list(range(1234, 1242))
    assert len({p["weights_sha"] for p in ranks["per_rank"]}) == 1
    per = [p["ms_per_step"] for p in ranks["per_rank"]]
    assert max(per) <= line["ms_per_step"] * 1.5 and line["ms_per_step"] >= 5.9       # the slowest ranks sleep 6 ms per step: the maximum is reported
