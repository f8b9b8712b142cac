"""What round 5 found in the kernels' ISA and fixed, kept fixed (CPU tier: hipcc cross-compiles gfx950 without a GPU; nothing is launched):
  * the direct conv kernel (dm3d_conv_h3v3.hip) compiles without a spilled register and without a vmcnt(0) inside a chunk loop — both came
    from the FLAT-encoded form of the weight LDS-DMA (a 64-bit lane address, and hipcc's wait-count pass treating vmcnt as out of order);
  * no loop kernel issues the FLAT form (__builtin_amdgcn_global_load_lds) any more;
  * the Winograd-x kernel's step loop holds no scratch traffic and no vmcnt(0);
  * the fused attention kernel's score loop waits for no vector-memory counter, and its epilogue requests every residual piece before
    the first store (a load under `if (res)` inside the store loop cost a vmcnt(0) — i.e. the previous store's completion — per row)."""
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "3d-condtional-stable-diffusion_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-Wno-unused-function", "-ffp-contract=off", "-S", "--cuda-device-only"]
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


def _compile(name, tmp_path):
    out = str(tmp_path / (name + ".s"))
    subprocess.run([HIPCC, *FLAGS, os.path.join(CSRC, name + ".hip"), "-o", out], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def _blocks(path):
    """[(kernel, label, mfma, scratch, vmcnt0)] per basic block (tools/isa_blocks.py's census)"""
    out, fn, cur = [], None, None
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            fn = m.group(1); cur = [fn, "entry", 0, 0, 0]; out.append(cur); continue
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m and fn:
            cur = [fn, m.group(1), 0, 0, 0]; out.append(cur); continue
        if cur is None:
            continue
        t = line.strip()
        if t.startswith("v_mfma"): cur[2] += 1
        elif t.startswith("scratch_"): cur[3] += 1
        elif t.startswith("s_waitcnt") and "vmcnt(0)" in t: cur[4] += 1
    return out


@pytest.fixture(scope="module")
def listings(tmp_path_factory):
    d = tmp_path_factory.mktemp("isa")
    names = ["dm3d_conv_h3v3", "dm3d_conv_h3w", "dm3d_attn_h3"]
    with ThreadPoolExecutor(len(names)) as ex:
        return dict(zip(names, ex.map(lambda n: _compile(n, d), names)))


def test_no_flat_form_lds_dma_in_the_loop_kernels():
    for name in ("dm3d_conv_h3v3.hip", "dm3d_conv_h3w.hip", "dm3d_conv_h3v2_parts.h", "dm3d_attn_h3.hip"):
        src = open(os.path.join(CSRC, name)).read()
        code = "\n".join(l.split("//")[0] for l in src.splitlines())
        assert "__builtin_amdgcn_global_load_lds" not in code, f"{name}: the FLAT form of the LDS-DMA is back (use raw_ptr_buffer_load_lds)"


def test_direct_conv_kernel_is_spill_free_and_its_chunk_loops_wait_counted(listings):
    text = open(listings["dm3d_conv_h3v3"]).read()
    spills = re.findall(r"\.name:\s+(\S*conv3d_igemm_h3v3\S*)[\s\S]*?\.vgpr_spill_count:\s+(\d+)", text)
    assert len(spills) >= 16
    assert all(int(n) == 0 for _, n in spills), [s for s in spills if int(s[1])]
    loops = [b for b in _blocks(listings["dm3d_conv_h3v3"]) if b[2] >= 128]          # the chunk loops: 160 (k2) ... 672 (k3) MFMAs per block
    assert len(loops) >= 16
    assert all(b[3] == 0 and b[4] == 0 for b in loops), [b for b in loops if b[3] or b[4]]


def test_winograd_step_loop_has_no_scratch_and_no_drained_wait(listings):
    loops = [b for b in _blocks(listings["dm3d_conv_h3w"]) if b[2] >= 128]
    assert len(loops) >= 3
    assert all(b[3] == 0 and b[4] == 0 for b in loops), [b for b in loops if b[3] or b[4]]


def test_attention_score_loop_and_epilogue(listings):
    lines = [l.strip() for l in open(listings["dm3d_attn_h3"])]
    k0 = next(i for i, l in enumerate(lines) if re.match(r"^_ZN\S*attn_fused_h3\S*:", l))
    k1 = next(i for i in range(k0, len(lines)) if lines[i].startswith("s_endpgm"))
    body = lines[k0:k1]
    mf = [i for i, l in enumerate(body) if l.startswith("v_mfma")]
    # score phase of the first unrolled tile body: from the first MFMA to the first v_exp (the softmax); the only vector-memory waits
    # allowed in between are none at all (the DMA pieces issued there are waited for at the next tile head)
    first_exp = next(i for i, l in enumerate(body) if l.startswith("v_exp_f32") and i > mf[0])
    waits = [l for l in body[mf[0]:first_exp] if l.startswith("s_waitcnt") and "vmcnt" in l]
    assert not waits, waits
    # epilogue: behind the last MFMA every global load precedes the first global store
    tail = body[mf[-1]:]
    st = next(i for i, l in enumerate(tail) if l.startswith("global_store"))
    assert not [l for l in tail[st:] if l.startswith("global_load")], "a residual load sits behind a store again"
    assert sum(l.startswith("global_store") for l in tail) >= 32
