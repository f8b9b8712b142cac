"""In-kernel cycle stamps of the fused attention kernel (wave 0 of the first workgroups).  Build the variant first:
    tools/mk_variant.sh attnst -DDM3D_ATTN_STAMPS
then   DM3D_LIB=$PWD/3d-condtional-stable-diffusion_amd/csrc/variants/attnst.so python tools/attn_stamps.py"""
import ctypes as C, os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dm3d_amd._lib import lib
buf = torch.zeros(4 * 128, dtype=torch.int64, device="cuda:0")
lib().dm3d_debug_set_stamps_attn.argtypes = [C.c_void_p]
lib().dm3d_debug_set_stamps_attn(buf.data_ptr())
sys.argv = ["attn_bench.py", "32"]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "attn_bench.py"), run_name="__main__")
torch.cuda.synchronize()
s = buf.cpu().numpy().reshape(4, 128)
names = ["wait+barrier", "issue DMA", "S mfma", "softmax", "PV mfma (to loop end)"]
for wg in range(2):
    print(f"workgroup {wg}: total loop {s[wg, 120] - s[wg, 0]} cycles, epilogue {s[wg, 121] - s[wg, 120]}")
    for t in range(8):
        r = s[wg, t * 8: t * 8 + 6]
        nxt = s[wg, (t + 1) * 8] if t < 7 else s[wg, 120]
        print(f"  tile {t}: " + "  ".join(f"{n} {int(b - a)}" for n, a, b in zip(names, r[:-1], r[1:])) + f"   | tile total {int(nxt - r[0]) if t < 7 else 0}")
