"""The --comm-file rendezvous of the host driver (`lnsfaid_sim --ranks N`, host/CommFile.h) without a GPU: a record left by an earlier
run must never be taken for this run's RCCL id (ncclCommInitRank with a dead id blocks for ever with the GPU held), and a rank that
finds nothing acceptable gives up with a non-zero exit code instead of waiting for ever."""
import os
import subprocess
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "mod-interleaveavx_multithreads-faid_amd", "host")
TOOL = os.path.join(HOST, "commfile_selftest")


@pytest.fixture(scope="module")
def tool():
    subprocess.check_call(["make", "-C", HOST, "commfile_selftest"], stdout=subprocess.DEVNULL)
    return TOOL


def run(*args, **kw):
    return subprocess.run([TOOL] + [str(a) for a in args], capture_output=True, text=True, **kw)


def test_fresh_id_of_the_same_run_is_taken(tool, tmp_path):
    f = tmp_path / "comm"
    assert run("publish", f, "run-1").returncode == 0
    r = run("fetch", f, "run-1", 500)
    assert r.returncode == 0 and r.stdout.strip() == "0"
    assert not (tmp_path / "comm.tmp").exists()  # published by rename


def test_reader_that_starts_first_waits_for_the_writer(tool, tmp_path):
    f = tmp_path / "comm"
    reader = subprocess.Popen([TOOL, "fetch", str(f), "run-2", "5000"], stdout=subprocess.PIPE, text=True)
    time.sleep(0.3)
    assert reader.poll() is None  # still polling: nothing there yet
    assert run("publish", f, "run-2").returncode == 0
    out, _ = reader.communicate(timeout=10)
    assert reader.returncode == 0 and out.strip() == "0"


def test_stale_record_of_another_run_id_is_ignored_and_the_wait_is_bounded(tool, tmp_path):
    f = tmp_path / "comm"
    assert run("publish", f, "yesterday").returncode == 0
    t0 = time.time()
    r = run("fetch", f, "today", 400)
    assert r.returncode == 3 and "no RCCL id" in r.stderr
    assert time.time() - t0 < 5


def test_old_record_of_the_same_run_id_is_ignored(tool, tmp_path):
    """a crashed run left its id behind an hour ago and the re-run uses the same --run-id: the reader that starts before the new
    rank 0 must not pick the dead id up"""
    f = tmp_path / "comm"
    assert run("publish", f, "nightly", 3600).returncode == 0
    assert run("fetch", f, "nightly", 300).returncode == 3
    # rank 0 of the new run clears and publishes: now the reader gets the NEW id (first byte 0, the stale one had 3600 & 0xff ^ 0)
    assert run("clear", f).returncode == 0 and not f.exists()
    assert run("publish", f, "nightly").returncode == 0
    r = run("fetch", f, "nightly", 300)
    assert r.returncode == 0 and r.stdout.strip() == "0"


def test_truncated_or_foreign_file_is_ignored(tool, tmp_path):
    f = tmp_path / "comm"
    f.write_bytes(b"\x00" * 128)  # what round 2's driver wrote: a bare id, no header
    assert run("fetch", f, "", 200).returncode == 3
    f.write_bytes(b"LNS")
    assert run("fetch", f, "", 200).returncode == 3
