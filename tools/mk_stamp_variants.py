"""Builds the stamped variant libraries read by tools/conv_stamps.py and tools/gemm_stamps.py:
    python tools/mk_stamp_variants.py        ->  <package>/csrc/variants/{cst,gst}.so
A stamped kernel writes __builtin_readcyclecounter() at phase boundaries (thread 0 of the first workgroups) into a buffer set
through dm3d_debug_set_stamps[_conv]; the product library carries none of this.  Run the tools with DM3D_LIB=<variant>."""
import os, subprocess
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3d-condtional-stable-diffusion_amd", "csrc")
os.chdir(CSRC)
subprocess.check_call(["make"])
os.makedirs("variants", exist_ok=True)
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I../../include", "-Wno-unused-function", "-ffp-contract=off"]
OBJS = ["dm3d_api.o", "dm3d_conv.o", "dm3d_conv_h3.o", "dm3d_conv_h3v2.o", "dm3d_gemm.o", "dm3d_gemm_h3.o", "dm3d_elem.o", "dm3d_train.o", "dm3d_attn_h3.o"]


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

# ---- conv3d_igemm_h3v2
s = open('dm3d_conv_h3v2.hip').read()
s=must(s, """namespace {

constexpr int REC = DM3D_REC;""","""__device__ unsigned long long* g_dbg_stamps_c = nullptr;
extern "C" int dm3d_debug_set_stamps_conv(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_stamps_c), &p, sizeof(p)); }
#define STAMP(i) do { if (g_dbg_stamps_c && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 4096) { g_dbg_stamps_c[blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memtime(); if ((i) == 1) g_dbg_stamps_c[blockIdx.x * 32 + 30] = __builtin_amdgcn_s_memrealtime(); if ((i) == 28) g_dbg_stamps_c[blockIdx.x * 32 + 31] = __builtin_amdgcn_s_memrealtime(); } } while (0)

namespace {

constexpr int REC = DM3D_REC;""",1)
s=must(s, """    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;""","""    STAMP(0);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;""",1)
s=must(s, """    if (NBUF >= 3) __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0), expcnt / lgkmcnt untouched
""","""    if (NBUF >= 3) __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0), expcnt / lgkmcnt untouched
    STAMP(1);
""")
s=must(s, """        if (!SPREAD) load_halo(ch_next);
""","""        if (ch - c_lo < 12) STAMP(2 + 2 * (ch - c_lo));
        if (!SPREAD) load_halo(ch_next);
""")
s=must(s, """            wb = wb + 1 == NBUF ? 0 : wb + 1;
            const bool last_group = g + 1 == NG;
            if (!last_group) {""","""            wb = wb + 1 == NBUF ? 0 : wb + 1;
            const bool last_group = g + 1 == NG;
            if (last_group && ch - c_lo < 12) STAMP(3 + 2 * (ch - c_lo));
            if (!last_group) {""")
s=must(s, """    // ---- epilogue.  Accumulator register r of tile""","""    STAMP(28);
    // ---- epilogue.  Accumulator register r of tile""")
s=must(s, """                    } else {
                        outz[o] = v;
                    }
                }
            }
        }
        if (p.range_flag && amax > rlim) *p.range_flag = 1;
        return;
    }""","""                    } else {
                        outz[o] = v;
                    }
                }
            }
        }
        if (p.range_flag && amax > rlim) *p.range_flag = 1;
        STAMP(29);
        return;
    }""")
# ping-pong loop: segment boundaries of the chunk loop's second chunk, waves 0 (half 0 -> entries 0..15 of the block's second row) and 4
s=must(s, """    constexpr bool PP = DM3D_PINGPONG""", """#define STAMPW(i) do { if (g_dbg_stamps_c && (threadIdx.x == 0 || threadIdx.x == 256) && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 2048 && (i) < 16) g_dbg_stamps_c[(2048 + blockIdx.x) * 32 + (threadIdx.x ? 16 : 0) + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
    constexpr bool PP = DM3D_PINGPONG""")
s=must(s, """            seg_barrier();
            // ---- X
            seg_barrier();""", """            if (ch == c_lo + 1) STAMPW(0);
            seg_barrier();
            if (ch == c_lo + 1) STAMPW(1);
            // ---- X
            seg_barrier();
            if (ch == c_lo + 1) STAMPW(2);""")
s=must(s, """                __builtin_amdgcn_sched_barrier(0);
                seg_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- C(pp)""", """                __builtin_amdgcn_sched_barrier(0);
                if (ch == c_lo + 1 && pp < 6) STAMPW(3 + 2 * pp);
                seg_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- C(pp)""")
s=must(s, """                __builtin_amdgcn_sched_barrier(0);
                seg_barrier();
                if (pr == 1) wb""", """                __builtin_amdgcn_sched_barrier(0);
                if (ch == c_lo + 1 && pp < 6) STAMPW(4 + 2 * pp);
                seg_barrier();
                if (pr == 1) wb""")
s=must(s, """            const int ch_next = ch + 1 < c_hi ? ch + 1 : ch;
            // ---- ST""", """            const int ch_next = ch + 1 < c_hi ? ch + 1 : ch;
            if (ch - c_lo < 12) STAMP(2 + 2 * (ch - c_lo));
            // ---- ST""")
s=must(s, """                if (pr == 1) wb = wb + 1 == NBUF ? 0 : wb + 1;
            }
        }
        if (!hsel) seg_barrier();""", """                if (pr == 1) wb = wb + 1 == NBUF ? 0 : wb + 1;
            }
            if (ch - c_lo < 12) STAMP(3 + 2 * (ch - c_lo));
        }
        if (!hsel) seg_barrier();""")
build(s, '_cst.hip', 'dm3d_conv_h3v2.o', 'variants/cst.so')
build(s.replace('#define STAMPW(i) do {', '#define STAMPW(i) do { break;').replace('#define STAMP(i) do {', '#define STAMP(i) do { if ((i) != 1 && (i) != 28) break;'), '_cck.hip', 'dm3d_conv_h3v2.o', 'variants/cck.so')

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
