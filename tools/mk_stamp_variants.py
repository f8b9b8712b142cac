"""Builds the stamped variant libraries read by tools/kernel_clock.py and tools/gemm_stamps.py:
    python tools/mk_stamp_variants.py        ->  <package>/csrc/variants/{cck,gst}.so
The kernels carry compiled-out stamp macros (-DDM3D_CLOCK_STAMPS, -DDM3D_GEMM_STAMPS): thread 0 of the first workgroups writes s_memtime /
s_memrealtime at phase boundaries into a buffer set through dm3d_debug_set_stamps[_conv]; the product library carries none of this.
Run the tools with DM3D_LIB=<variant>."""
import os, subprocess
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3d-condtional-stable-diffusion_amd", "csrc")
os.chdir(CSRC)
subprocess.check_call(["make"])
os.makedirs("variants", exist_ok=True)
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I../../include", "-Wno-unused-function", "-ffp-contract=off"]
OBJS = ["dm3d_api.o", "dm3d_conv.o", "dm3d_conv_h3.o", "dm3d_conv_h3_host.o", "dm3d_conv_h3v3.o", "dm3d_conv_h3w.o", "dm3d_gemm.o", "dm3d_gemm_h3.o", "dm3d_mlp_h3.o", "dm3d_attn_front_h3.o", "dm3d_elem.o", "dm3d_train.o", "dm3d_attn_h3.o"]


# ---- conv3d_igemm_h3v3
# the free-running kernel carries its own (compiled-out) clock stamps: -DDM3D_CLOCK_STAMPS
subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "-DDM3D_CLOCK_STAMPS", "-c", "dm3d_conv_h3v3.hip", "-o", "_cck3.o"])
subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "-DDM3D_CLOCK_STAMPS", "-c", "dm3d_conv_h3w.hip", "-o", "_cckw.o"])     # (the Winograd-x form: dm3d_debug_set_stamps_wino)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", "variants/cck.so",
                       *["_cck3.o" if o == "dm3d_conv_h3v3.o" else ("_cckw.o" if o == "dm3d_conv_h3w.o" else o) for o in OBJS]])
os.remove("_cck3.o"); os.remove("_cckw.o")

# ---- gemm_tn_h3: compiled-out stamps of its own: -DDM3D_GEMM_STAMPS
subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "-DDM3D_GEMM_STAMPS", "-c", "dm3d_gemm_h3.hip", "-o", "_gst.o"])
subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", "variants/gst.so", *["_gst.o" if o == "dm3d_gemm_h3.o" else o for o in OBJS]])
os.remove("_gst.o")
