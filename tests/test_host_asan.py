"""CPU-only: the host halves of the library under AddressSanitizer.  This file is listed in .gpurunignore: the GPU
pool refuses sanitizer builds, and nothing here needs a device."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_side_under_address_sanitizer(tmp_path):
    """SURVEY 5 (sanitizers): the host halves of every .hip file rebuilt with -fsanitize=address (GPU ASAN does not exist on
    this pool; the flag is ignored for the gfx950 half) and driven by tests/abi_asan.c: hostile descriptors — nulls, misaligned
    pointers, bad sizes — through every entry that validates before it launches.  Any host-side overrun aborts the program."""
    from concurrent.futures import ThreadPoolExecutor
    csrc = os.path.join(ROOT, "3d-condtional-stable-diffusion_amd", "csrc")
    srcs = sorted(f for f in os.listdir(csrc) if f.endswith(".hip"))
    flags = ["-O1", "-g", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
             "-Wno-unused-function", "-Wno-option-ignored", "-ffp-contract=off", "-fsanitize=address", "-fno-omit-frame-pointer"]

    def cc(f):
        subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-c", os.path.join(csrc, f), "-o", str(tmp_path / (f + ".o"))], check=True,
                       capture_output=True)
    with ThreadPoolExecutor(4) as ex:
        list(ex.map(cc, srcs))
    so = tmp_path / "libdm3d_hip.so"
    subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-fsanitize=address", "-Wno-option-ignored",
                    "-o", str(so)] + [str(tmp_path / (f + ".o")) for f in srcs], check=True, capture_output=True)
    exe = tmp_path / "abi_asan"
    subprocess.run(["/opt/rocm/lib/llvm/bin/clang", "-fsanitize=address", "-g", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "abi_asan.c"), "-o", str(exe), "-L", str(tmp_path), "-ldm3d_hip",
                    f"-Wl,-rpath,{tmp_path}"], check=True, capture_output=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "refused 13 of 13" in r.stdout and "AddressSanitizer" not in r.stderr
