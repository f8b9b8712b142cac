"""CPU tier: the static hazard check for inline-asm MFMAs (tools/isa_hazard.py) — its rules on hand-written listings, green on every kernel
file of the library that issues an MFMA from `asm`, red on the pre-fix form of the attention front kernel (asm MFMAs).
hipcc cross-compiles gfx950 ISA without a GPU; nothing here launches a kernel."""
import glob
import io
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "3d-condtional-stable-diffusion_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_hazard  # noqa: E402

HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}", "-Wno-unused-function",
         "-ffp-contract=off", "-S", "--cuda-device-only"]


def _listing(tmp_path, body):
    p = tmp_path / "k.s"
    p.write_text("_Z1kv:\n" + body + "\ts_endpgm\n")
    return str(p)


def _kinds(path, **kw):
    ks, (ins, labels) = next(iter(isa_hazard.parse(path).items()))
    return sorted(f[0] for f in isa_hazard.check_kernel(ks, ins, labels, **kw))


ASM = "\t;;#ASMSTART\n\t{}\n\t;;#ASMEND\n"
M16 = "v_mfma_f32_16x16x32_f16 a[0:3], v[8:11], v[12:15], a[0:3]"
M32 = "v_mfma_f32_32x32x16_f16 v[16:31], v[8:11], v[12:15], v[16:31]"


def test_rules_on_handwritten_listings(tmp_path):
    # RAW: a VALU write of A / B / C fewer than two wait states in front of an asm MFMA (measured: profiles/r05_mfma_hazards.log)
    assert _kinds(_listing(tmp_path, "\tv_cvt_pk_f16_f32 v11, v0, v1\n\ts_nop 0\n" + ASM.format(M16))) == ["RAW-hazard"]
    assert _kinds(_listing(tmp_path, "\tv_cvt_pk_f16_f32 v11, v0, v1\n\ts_nop 1\n" + ASM.format(M16))) == []
    assert _kinds(_listing(tmp_path, "\tv_mov_b64_e32 v[16:17], v[0:1]\n\ts_waitcnt lgkmcnt(2)\n" + ASM.format(M32))) == ["RAW-hazard"]
    assert _kinds(_listing(tmp_path, "\tv_accvgpr_write_b32 a2, v5\n" + ASM.format(M16))) == ["RAW-hazard"]
    # ... a load's destination is not a VALU write (its s_waitcnt orders it), and the compiler's own MFMAs are its business unless --all
    assert _kinds(_listing(tmp_path, "\tds_read_b128 v[8:11], v40\n\ts_waitcnt lgkmcnt(0)\n" + ASM.format(M16))) == []
    assert _kinds(_listing(tmp_path, "\tv_mov_b32_e32 v8, v1\n\t" + M16 + "\n")) == []
    assert _kinds(_listing(tmp_path, "\tv_mov_b32_e32 v8, v1\n\t" + M16 + "\n"), all_mfma=True) == ["RAW-hazard"]
    # ... through a branch: the write sits at the end of the predecessor block
    assert _kinds(_listing(tmp_path, "\tv_mov_b32_e32 v12, v1\n\ts_cbranch_scc1 .LBB0_2\n\ts_nop 7\n.LBB0_2:\n" + ASM.format(M16))) == ["RAW-hazard"]
    # D: P + 4 wait states before anything but the accumulate chain touches the result (4 passes: 8, 8 passes: 12)
    assert _kinds(_listing(tmp_path, ASM.format(M16) + "\ts_nop 6\n\tv_accvgpr_read_b32 v1, a3\n")) == ["D-hazard"]
    assert _kinds(_listing(tmp_path, ASM.format(M16) + "\ts_nop 7\n\tv_accvgpr_read_b32 v1, a3\n")) == []
    assert _kinds(_listing(tmp_path, ASM.format(M32) + "\ts_nop 7\n\ts_nop 2\n\tv_add_f32_e32 v1, v31, v2\n")) == ["D-hazard"]
    assert _kinds(_listing(tmp_path, ASM.format(M32) + "\ts_nop 7\n\ts_nop 3\n\tv_add_f32_e32 v1, v31, v2\n")) == []
    assert _kinds(_listing(tmp_path, ASM.format(M32) + "\tglobal_store_dwordx4 v[40:41], v[16:19], off\n")) == ["D-hazard"]
    # ... the chain itself needs nothing; an MFMA reading D as an operand does
    assert _kinds(_listing(tmp_path, ASM.format(M16) + ASM.format(M16))) == []
    assert _kinds(_listing(tmp_path, ASM.format(M16) + ASM.format("v_mfma_f32_16x16x32_f16 a[4:7], a[0:3], v[12:15], a[4:7]"))) == ["D-hazard"]
    # ... and the wait states run through a loop's back edge
    assert _kinds(_listing(tmp_path, ".LBB0_1:\n\tv_accvgpr_read_b32 v1, a0\n" + ASM.format(M16) + "\ts_cbranch_scc1 .LBB0_1\n")) == ["D-hazard"]
    # A / B overwritten BEHIND the MFMA: measured harmless (operands are read at issue) — listed with war_ab only
    body = ASM.format(M16) + "\tv_mov_b32_e32 v8, v1\n"
    assert _kinds(_listing(tmp_path, body)) == [] and _kinds(_listing(tmp_path, body), war_ab=True) == ["AB-write"]
    assert isa_hazard.mfma_passes("v_mfma_f32_16x16x32_f16") == 4 and isa_hazard.mfma_passes("v_mfma_f32_32x32x16_f16") == 8
    assert isa_hazard.mfma_passes("v_mfma_f32_32x32x2_f32") == 16


def _asm_mfma_sources():
    """every kernel file that issues an MFMA from an asm statement"""
    out = []
    for f in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        if re.search(r'asm\s+volatile\s*\(\s*"[^"]*v_mfma', open(f).read()):
            out.append(f)
    return out


def _compile(src, out):
    subprocess.run([HIPCC, *FLAGS, src, "-o", out], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_asm_mfma_sites_of_the_library_are_hazard_free(tmp_path):
    srcs = _asm_mfma_sources()
    assert [os.path.basename(s) for s in srcs] == ["dm3d_conv_h3w.hip", "dm3d_mlp_h3.hip"], "a new file issues asm MFMAs: it is checked below; list it here"
    with ThreadPoolExecutor(len(srcs)) as ex:
        lst = list(ex.map(lambda s: _compile(s, str(tmp_path / (os.path.basename(s) + ".s"))), srcs))
    for path in lst:
        buf = io.StringIO()
        sites, hazards = isa_hazard.run(path, out=buf)
        print(buf.getvalue())
        assert sites >= 384, f"{path}: the checker found no asm MFMA ({sites})"
        assert hazards == 0, buf.getvalue()


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_checker_fires_on_the_asm_form_of_the_front_kernel(tmp_path):
    """Round 4's silent corruption: dm3d_attn_front_h3.hip with its MFMAs as inline asm (the form that was never committed: the fix —
    the compiler builtin — went in with the file).  Rebuilt here from the macro it differs in: the checker must report the
    read-after-write it was (hipcc's v_mov_b64 copies of the zeroed accumulator directly in front of the first asm MFMA of a block)."""
    src = open(os.path.join(CSRC, "dm3d_attn_front_h3.hip")).read()
    builtin = "#define DM3D_MFMA_VV(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0)"
    drain = "#define DM3D_MFMA_DRAIN() do { } while (0)"
    assert builtin in src and drain in src
    src = src.replace(builtin, '#define DM3D_MFMA_VV(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))')
    src = src.replace(drain, '#define DM3D_MFMA_DRAIN() asm volatile("s_nop 7\\n\\ts_nop 7" ::: "memory")')
    p = tmp_path / "front_asm.hip"
    p.write_text(src)
    lst = _compile(str(p), str(tmp_path / "front_asm.s"))
    buf = io.StringIO()
    sites, hazards = isa_hazard.run(lst, out=buf)
    print(buf.getvalue())
    assert sites == 1440 and hazards >= 1 and "RAW-hazard" in buf.getvalue()
    # and the committed form (builtin MFMAs: hipcc pads them) is clean even with every MFMA checked
    good = _compile(os.path.join(CSRC, "dm3d_attn_front_h3.hip"), str(tmp_path / "front.s"))
    assert isa_hazard.run(good, all_mfma=True, quiet=True) == (1440, 0)
