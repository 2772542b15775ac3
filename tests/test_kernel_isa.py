"""Properties of the compiled decode kernels that performance depends on and that a source change can silently lose
(with two waves per SIMD nothing hides a stall).  Cross-compiles the kernels to gfx950 assembly; no GPU needed."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mod-interleaveavx_multithreads-faid_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def kernel4_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / "kernel4.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S",
                    "--cuda-device-only", "-o", str(out), os.path.join(CSRC, "lnsfaid_kernel4.hip")], check=True, capture_output=True)
    return out.read_text()


def kernel_bodies(asm):
    """name -> assembly text of every lnsfaid_decode4_kernel<METHOD> instance"""
    parts = re.split(r"^(_Z\w+):", asm, flags=re.M)
    return {parts[i]: parts[i + 1].split(".end_amdhsa_kernel")[0] for i in range(1, len(parts) - 1, 2) if "lnsfaid_decode4_kernel" in parts[i]}


def test_no_scratch_and_two_waves_per_simd(kernel4_asm):
    sizes = [int(x) for x in re.findall(r"\.private_segment_fixed_size:\s*(\d+)", kernel4_asm)]
    assert sizes and all(s == 0 for s in sizes), sizes  # a struct passed by reference to a non-inlined function ends up in scratch
    vgprs = [int(x) for x in re.findall(r"\.vgpr_count:\s*(\d+)", kernel4_asm)]
    assert vgprs and max(vgprs) <= 256, vgprs  # 512 registers per SIMD lane / 2 waves
    assert all(int(x) == 0 for x in re.findall(r"\.vgpr_spill_count:\s*(\d+)", kernel4_asm))


def test_hot_path_has_no_function_calls(kernel4_asm):
    # the only device function left out of line is the EF_ELIMINATION 2 erasure plane (rare path of DecodeMethod 2)
    funcs = [m for m in re.findall(r"^(_Z\w+):", kernel4_asm, flags=re.M) if "lnsfaid_decode4_kernel" not in m]
    assert all("build_erasure_plane4" in f for f in funcs), funcs
    for name, body in kernel_bodies(kernel4_asm).items():
        calls = len(re.findall(r"s_swappc_b64", body))
        assert calls <= (2 if "ILi2E" in name else 0), (name, calls)


def test_layer_loop_waits_for_no_memory_but_the_prefetch(kernel4_asm):
    # inside the layer loops (depth 2) vector memory is: prefetch of the next layer's messages at the top, one wait for it at the
    # bottom, the store of this layer's messages; a vmcnt wait anywhere else is a memory round trip per layer
    for name, body in kernel_bodies(kernel4_asm).items():
        info, n = "", 0
        for line in body.split("\n"):
            m = re.match(r"^(\.LBB\d+_\d+):\s*;?(.*)", line)
            if m:
                info, n = m.group(2), 0
                continue
            if re.match(r"^\s+[a-z]", line):
                n += 1
                if "Depth=2" in info and "Header" in info and "s_waitcnt" in line and "vmcnt" in line:
                    assert n < 120, (name, n, line.strip())  # only in the short block that takes over the prefetched data
