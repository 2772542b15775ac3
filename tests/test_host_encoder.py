"""CPU-side check of the host driver's encoder (host/CEncoder.cpp): no GPU involved."""
import os
import subprocess

import oracle_abi as oa


def test_encoder_selftest_builds_and_every_frame_is_a_codeword():
    host = os.path.join(oa.PKG_DIR, "host")
    subprocess.check_call(["make", "-s", "-C", host, "encoder_selftest"])
    res = subprocess.run([os.path.join(host, "encoder_selftest")], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "unsatisfied checks over 32 frames: 0" in res.stdout
    assert "systematic part differs in 0 positions" in res.stdout
