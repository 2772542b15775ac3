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
        assert calls <= (2 if "ILi2ELb0ELb1E" in name else 0), (name, calls)  # <2, RM = false, EF2 = true>


def layer_loop_blocks(body):
    """[(label, info, [instruction lines])] of the loop that holds the layer step (the largest depth-2 block and its loop mates)"""
    blocks, cur = [], None
    for line in body.split("\n"):
        m = re.match(r"^(\.LBB\d+_\d+):\s*;?(.*)", line)
        if m:
            cur = [m.group(1), m.group(2), []]
            blocks.append(cur)
        elif cur is not None and re.match(r"^\s+;.*(Loop|Depth)", line):
            cur[1] += " " + line.strip()
        elif cur is not None and re.match(r"^\s+[a-z]", line):
            cur[2].append(line.strip())
    big = max((b for b in blocks if "Depth=2" in b[1]), key=lambda b: len(b[2]))
    header = re.search(r"Header=(BB\d+_\d+)", big[1]).group(1)
    return [b for b in blocks if "Depth=2" in b[1] and ("Header=" + header + " " in b[1] + " " or b[0] == ".L" + header)]


def test_messages_in_registers_means_no_memory_traffic_in_the_layer_loop(kernel4_asm):
    # RM instances <METHOD, true, false>: the compressed messages live in registers (indexed moves under s_set_gpr_idx_on), so the
    # layer loop holds no global / scratch store and no load but the one-dword prefetch of the next layer's edge table
    bodies = {n: b for n, b in kernel_bodies(kernel4_asm).items() if "Lb1ELb0E" in n}
    assert len(bodies) == 5, sorted(bodies)
    for name, body in bodies.items():
        loop = layer_loop_blocks(body)
        ops = [i for b in loop for i in b[2]]
        assert sum(len(b[2]) for b in loop) > 1500, name  # the two per-degree instances of the layer step are in it
        stores = [i for i in ops if re.match(r"(global|flat|buffer|scratch)_store", i)]
        loads = [i for i in ops if re.match(r"(global|flat|buffer|scratch)_load", i)]
        assert not stores, (name, stores)
        assert len(loads) <= 1 and all(i.startswith("global_load_dword ") for i in loads), (name, loads)
        assert any("s_set_gpr_idx_on" in i for i in ops), name


def test_layer_loop_waits_for_no_memory_but_the_prefetch(kernel4_asm):
    # inside the layer loops (depth 2) vector memory is: prefetch of the next layer's messages at the top, one wait for it at the
    # bottom, the store of this layer's messages; a vmcnt wait anywhere else is a memory round trip per layer
    for name, body in kernel_bodies(kernel4_asm).items():
        if "Lb1ELb0E" in name:
            continue  # messages in registers: covered by the test above
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


@pytest.fixture(scope="module")
def kernel5_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa5") / "kernel5.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S",
                    "--cuda-device-only", "-o", str(out), os.path.join(CSRC, "lnsfaid_kernel5.hip")], check=True, capture_output=True)
    return out.read_text()


def test_two_waves_per_codeword_kernel_fits_four_waves_per_simd(kernel5_asm):
    # experimental lnsfaid_kernel5.hip: 128 registers per wave (8 codewords per CU x 2 waves = 4 waves per SIMD), no static LDS, and
    # what the register budget spills stays outside the layer loops of both roles
    parts = re.split(r"^(_Z\w+):", kernel5_asm, flags=re.M)
    bodies = {parts[i]: parts[i + 1].split(".end_amdhsa_kernel")[0] for i in range(1, len(parts) - 1, 2) if "lnsfaid_decode5_kernel" in parts[i]}
    assert len(bodies) == 5, sorted(bodies)
    vgprs = [int(x) for x in re.findall(r"\.vgpr_count:\s*(\d+)", kernel5_asm)]
    assert vgprs and max(vgprs) <= 128, vgprs
    assert all(int(x) == 0 for x in re.findall(r"\.group_segment_fixed_size:\s*(\d+)", kernel5_asm))
    for name, body in bodies.items():
        info, big = "", []
        for line in body.split("\n"):
            m = re.match(r"^(\.LBB\d+_\d+):\s*;?(.*)", line)
            if m:
                big.append([m.group(2), []])
            elif big and re.match(r"^\s+;.*(Loop|Depth)", line):
                big[-1][0] += " " + line.strip()
            elif big and re.match(r"^\s+[a-z]", line):
                big[-1][1].append(line.strip())
        layer_blocks = [b for b in big if "Depth=2" in b[0] and len(b[1]) > 300]
        assert len(layer_blocks) >= 4, (name, len(layer_blocks))  # two per-degree instances of the layer step for either role
        for info, ops in layer_blocks:
            assert not [i for i in ops if i.startswith("scratch_")], (name, info)
            assert sum(1 for i in ops if i.startswith("s_barrier")) >= 3, (name, info)
