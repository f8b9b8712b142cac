"""Builds the stamped variant libraries read by tools/kernel_clock.py and tools/gemm_stamps.py:
    python tools/mk_stamp_variants.py        ->  <package>/csrc/variants/{cck,gst}.so
A stamped kernel writes __builtin_readcyclecounter() at phase boundaries (thread 0 of the first workgroups) into a buffer set
through dm3d_debug_set_stamps[_conv]; the product library carries none of this.  Run the tools with DM3D_LIB=<variant>."""
import os, subprocess
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3d-condtional-stable-diffusion_amd", "csrc")
os.chdir(CSRC)
subprocess.check_call(["make"])
os.makedirs("variants", exist_ok=True)
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I../../include", "-Wno-unused-function", "-ffp-contract=off"]
OBJS = ["dm3d_api.o", "dm3d_conv.o", "dm3d_conv_h3.o", "dm3d_conv_h3v2.o", "dm3d_conv_h3v3.o", "dm3d_gemm.o", "dm3d_gemm_h3.o", "dm3d_elem.o", "dm3d_train.o", "dm3d_attn_h3.o"]


def build(src_text, tmp_name, replaces, out):
    open(tmp_name, "w").write(src_text)
    try:
        subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "-c", tmp_name, "-o", tmp_name + ".o"])
    finally:
        os.remove(tmp_name)
    objs = [tmp_name + ".o" if o == replaces else o for o in OBJS]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", out, *objs])
    os.remove(tmp_name + ".o")


def must(s, old, new, count=-1):
    assert old in s, old[:60]
    return s.replace(old, new, count) if count > 0 else s.replace(old, new)

# ---- conv3d_igemm_h3v3
# the free-running kernel carries its own (compiled-out) clock stamps: -DDM3D_CLOCK_STAMPS
subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "-DDM3D_CLOCK_STAMPS", "-c", "dm3d_conv_h3v3.hip", "-o", "_cck3.o"])
subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", "variants/cck.so", *["_cck3.o" if o == "dm3d_conv_h3v3.o" else o for o in OBJS]])
os.remove("_cck3.o")

# ---- gemm_tn_h3
s = open('dm3d_gemm_h3.hip').read()
s=must(s, """namespace {

struct GemmH3Args {""","""__device__ unsigned long long* g_dbg_stamps = nullptr;
extern "C" int dm3d_debug_set_stamps(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_stamps), &p, sizeof(p)); }
#define STAMP(i) do { if (g_dbg_stamps && threadIdx.x == 0 && blockIdx.x < 2048) { g_dbg_stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); if ((i) == 1) g_dbg_stamps[blockIdx.x * 16 + 14] = __builtin_amdgcn_s_memrealtime(); if ((i) == 10) g_dbg_stamps[blockIdx.x * 16 + 15] = __builtin_amdgcn_s_memrealtime(); } } while (0)

namespace {

struct GemmH3Args {""",1)
s=must(s, """    int which = 0;
#pragma unroll
    for (int i = 1; i < MAX_GROUP; ++i) which +=""","""    STAMP(0);
    int which = 0;
#pragma unroll
    for (int i = 1; i < MAX_GROUP; ++i) which +=""",1)
s=must(s, """    fetch(S0, 0);
    fetch(S1, clampk(1));""","""    STAMP(1);
    fetch(S0, 0);
    fetch(S1, clampk(1));""")
s=must(s, """        __syncthreads();                        // chunk `it` visible; everyone has left chunk it-1 (other buffer)""","""        __syncthreads();                        // chunk `it` visible; everyone has left chunk it-1 (other buffer)
        if (it < 8) STAMP(2 + it);""")
s=must(s, """        publish(S1, 1, (it + 1) * KC + 16 < p.k);
        __syncthreads();""","""        publish(S1, 1, (it + 1) * KC + 16 < p.k);
        __syncthreads();
        if (it < 8) STAMP(3 + it);""")
s=must(s, """    const bool full = m0 + TM <= p.m && n0 + NT <= p.n;
    char* O = static_cast<char*>(p.out)""","""    STAMP(10);
    const bool full = m0 + TM <= p.m && n0 + NT <= p.n;
    char* O = static_cast<char*>(p.out)""")
s=must(s, """    } else {
        if (p.out_h2) epilogue(no, yes); else epilogue(no, no);
    }
    if (p.range_flag && amax > p.range_limit) *p.range_flag = 1;
}""","""    } else {
        if (p.out_h2) epilogue(no, yes); else epilogue(no, no);
    }
    if (p.range_flag && amax > p.range_limit) *p.range_flag = 1;
    STAMP(11);
}""")
build(s, '_gst.hip', 'dm3d_gemm_h3.o', 'variants/gst.so')
