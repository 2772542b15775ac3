#!/usr/bin/env python3
"""bench.py — decoded information Gb/s of the MI355X LDPC decode path (BASELINE.json metric).

One "step" = one pass of the hot path (lnsfaid_decode_device + the error-counter pass) over one batch of
synthetic AWGN frames that is already resident in HBM:
  workload (BASELINE.json configs[1]): 50G-PON code, QPSK, DecodeMethod 2 (3-bit LNS-FAID, FAID3 tables,
  DTBF), MaxIteration 10, batch 65 536 codewords = 2048 groups of 32 per GPU, all-zero codeword,
  LLR = clamp(trunc(13 * (-0.707107 + n)), -7, 7), n ~ N(0, sigma^2 / 2)  (reference CSimulate.cpp:73,126).
Headline point: Eb/N0 = 3.0 dB, where no frame converges, so every codeword executes exactly 10 layered
iterations + 10 bit-flipping iterations: the data-independent "@ 10 iters" worst case.  Eb/N0 3.6 and 4.2 dB
(early stop active) are reported next to it in "points".

N > 1 (`python bench.py --gpus N`, run plainly): this process touches neither torch nor the GPU; it starts
`python -m torch.distributed.run --nproc-per-node N` on itself — N fresh children, one per GPU, backend nccl = RCCL —
and relays rank 0's JSON line.  Run under torch.distributed.run already (RANK in the environment) it is a worker.
Groups of 32 codewords are independent, so there is no data-path collective: the only exchange is the all-reduce of the
four error counters per step (reference main.cpp:174-182).  Two legs, SURVEY.md 8(d) config 4:
  weak   (the JSON line's `value`): every rank decodes its own --groups groups (2048 = 65 536 codewords per GPU);
  strong (`strong` object):        the --groups groups are split into contiguous ranges of whole groups, one per rank
                                    (dist.shard_groups: 256 groups per GPU at N = 8).

The CPU port under oracle/ is used here only for the cpu_baseline leg and its parity flag; it is never part of the
measured path (`--launcher-selftest` is a CPU-only test of the spawn / shard / reduce plumbing and says so in its line).
"""
import argparse
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "mod-interleaveavx_multithreads-faid_amd")


def load_pkg_module(name):
    """The package directory name is not a Python identifier: its modules (pyabi.py, dist.py) are imported by path."""
    import importlib.util
    if "lnsfaid_" + name in sys.modules:
        return sys.modules["lnsfaid_" + name]
    spec = importlib.util.spec_from_file_location("lnsfaid_" + name, os.path.join(PKG_DIR, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["lnsfaid_" + name] = mod  # one instance per process: ctypes structure classes are compared by identity
    spec.loader.exec_module(mod)
    return mod


def load_oracle_abi():
    """tests/oracle_abi.py: the ctypes view of oracle/ (test infrastructure).  Imported ONLY by the cpu_baseline leg and by the
    CPU-only --launcher-selftest; the measured path never sees it."""
    tests_dir = os.path.join(ROOT, "tests")
    if tests_dir not in sys.path:
        sys.path.insert(0, tests_dir)
    import oracle_abi
    return oracle_abi


N_VAR, N_CHECK, N_EDGES = 17664, 3072, 70400
K_INFO = N_VAR - N_CHECK
RATE = 0.8444444  # m_Rate, reference CLDPC.cpp:4780
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# VALU issue roof (MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 32 lanes wide, 2.4 GHz): one wave64 VALU instruction
# occupies its SIMD for 2 cycles at the full rate
N_SIMD, CLOCK_GHZ, CYCLES_PER_WAVE64_VALU = 1024, 2.4, 2
VALU_PEAK_GINSTR = N_SIMD * CLOCK_GHZ / CYCLES_PER_WAVE64_VALU
KERNEL_SOURCES = ["lnsfaid_kernel4.hip", "lnsfaid_rows4.h", "lnsfaid_swar.h", "lnsfaid_phases.h", "lnsfaid_kernels.hip", "lnsfaid_device.h", "Makefile"]


def kernel_source_hash():
    """Stamp of the decode kernel's source: counter files under profiles/ are only replayed for the kernel they measured."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "mod-interleaveavx_multithreads-faid_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def algorithmic_bytes(i_layered, j_bf):
    """bytes the reference's data layout moves per codeword (int8 En and Lmn)."""
    return 2 * N_VAR + i_layered * (4 * N_EDGES + N_VAR) + j_bf * 2 * N_VAR


def synth_llr(torch, device, n_groups, eb_n0, seed, mod_type=2, scale=13.0):
    """All-zero codeword through the reference's mapper / AWGN / max-log demapper / 4-bit quantiser, on the GPU.
    mod_type 2: QPSK (LLR = -0.707107 + n); mod_type 4: 16-QAM (per symbol r, i, |r| - c, |i| - c with r, i = -0.316228 + n,
    reference CModulate.cpp:5, :283-293).  The all-zero word makes every position statistically alike, so the LLRs are drawn
    directly in the decoder's [32][K] | [32][M] layout."""
    sigma = 1.0 / math.sqrt(RATE * mod_type * 10.0 ** (0.1 * eb_n0))  # CSimulate.cpp:73
    sigma_ch = sigma / math.sqrt(2.0)  # CSimulate.cpp:126
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    out = torch.empty((n_groups, 32 * N_VAR), dtype=torch.int8, device=device)
    chunk = 128
    for g0 in range(0, n_groups, chunk):
        g1 = min(n_groups, g0 + chunk)
        if mod_type == 2:
            x = torch.randn((g1 - g0, 32 * N_VAR), generator=gen, device=device, dtype=torch.float32)
            x = x * sigma_ch - 0.707107
        else:
            ri = torch.randn((g1 - g0, 32 * N_VAR // 4, 2), generator=gen, device=device, dtype=torch.float32) * sigma_ch - 0.316228
            x = torch.cat([ri, ri.abs() - 0.6324555], dim=2).reshape(g1 - g0, 32 * N_VAR)
        out[g0:g1] = (x * scale).trunc().clamp_(-7, 7).to(torch.int8)  # float2LimitChar_4bit, CLDPC.cpp:4553-4573
    return out


def kernel_instance_name(method, rows_per_lane, message_store, pyabi, waves=1):
    """The template instance a context launches, as rocprofv3 prints it (what the counter files under profiles/ are keyed on)."""
    if waves == 2:
        return "void lnsfaid_decode5_kernel<%d>(LfKernelArgs)" % method
    if rows_per_lane == 4:
        return "void lnsfaid_decode4_kernel<%d, %s, false>(LfKernelArgs)" % (method, "true" if message_store == pyabi.MSG_REGISTERS else "false")
    return "void lnsfaid_decode_kernel<%d, true>(LfKernelArgs)" % method


def host_cores():
    """Cores this process may run on: the affinity mask, cut down to the cgroup's CPU quota where one is set."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(math.ceil(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--groups", type=int, default=2048, help="groups of 32 codewords per GPU (weak leg) / in total (strong leg)")
    ap.add_argument("--eb-n0", type=float, default=3.0)
    ap.add_argument("--method", type=int, default=2)
    ap.add_argument("--max-iter", type=int, default=10)
    ap.add_argument("--max-bf", type=int, default=None, help="override _maxBFiter (experiments only)")
    ap.add_argument("--factor-1", type=int, default=None, help="Profile.txt Factor_1 (default: the shipped 1; NMS runs use e.g. 24)")
    ap.add_argument("--factor-2", type=int, default=None, help="Profile.txt Factor_2 (default: the shipped 6)")
    ap.add_argument("--mod-type", type=int, default=2, choices=[2, 4], help="Profile.txt modType: 2 QPSK, 4 16-QAM")
    ap.add_argument("--scale", type=float, default=13.0, help="Profile.txt scale (12.5 for the hybrid 2B1C decoder)")
    ap.add_argument("--no-points", action="store_true", help="skip the 3.6 / 4.2 dB side measurements")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-dropin", action="store_true", help="skip the drop-in call-shape leg (host/dropin_bench)")
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the strong-scaling leg")
    ap.add_argument("--cpu-groups", type=int, default=2048, help="groups of the cpu_baseline sample per Eb/N0 point")
    ap.add_argument("--cpu-seconds", type=float, default=1.0, help="minimum wall time of the cpu_baseline leg per Eb/N0 point")
    ap.add_argument("--cpu-threads", type=int, default=0, help="cap on the cpu_baseline threads (0: every core this process may use)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="counter all-reduce transport (nccl = RCCL)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="let several ranks use one GPU (rank -> device rank %% device_count; rehearsal on a one-GPU box, "
                         "needs --backend gloo: RCCL refuses two ranks on one device)")
    ap.add_argument("--spawn", action="store_true", help="go through the torch.distributed.run launcher for N = 1 as well")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="CPU-only test of the launcher / sharding / counter reduction: the CPU port stands in for the GPU "
                         "library on a tiny batch, gloo backend; the line it prints is not a measurement and says so")
    return ap.parse_args(argv)


# ---- parent: spawn one fresh worker per GPU before anything touches torch or the GPU ---------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_children(args, argv):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out_line in proc.stdout.splitlines():
        if out_line.startswith("{") and '"metric"' in out_line:
            line = out_line
    if proc.returncode != 0 or line is None:
        sys.stderr.write(proc.stdout)
        raise SystemExit("bench.py: the %d-rank job failed (exit code %d)" % (args.gpus, proc.returncode))
    print(line)
    return 0


# ---- worker ----------------------------------------------------------------------------------------------------------
class SelftestDecoder:
    """--launcher-selftest only: the CPU port behind the Decoder interface the worker uses (numpy arrays instead of
    device pointers).  Exists so that the spawn / shard / all-reduce plumbing can be tested without a GPU."""

    def __init__(self, oa, code, cfg):
        self.oa, self.code, self.cfg = oa, code, cfg
        self.ms, self.launches = 0.0, 0

    def decode(self, fix, n_groups):
        t0 = time.perf_counter()
        dec, stats = self.oa.decode_mt(self.code, self.cfg, fix, n_groups, threads=2, kind="avx2")
        self.ms += (time.perf_counter() - t0) * 1e3
        self.launches += 1
        return dec, stats

    def count(self, dec, n_groups):
        return self.oa.Oracle(self.code, self.cfg).count_errors(dec, None, n_groups)

    def kernel_time(self, reset=False):
        r = (self.ms, self.launches)
        if reset:
            self.ms, self.launches = 0.0, 0
        return r

    def close(self):
        pass


def worker(args):
    import numpy as np
    import torch
    pyabi = load_pkg_module("pyabi")
    lnsfaid_dist = load_pkg_module("dist")
    oa = load_oracle_abi() if args.launcher_selftest else None  # the GPU legs do not import it (cpu_baseline does, below)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    selftest = args.launcher_selftest
    backend = "gloo" if selftest else args.backend
    if selftest:
        device = torch.device("cpu")
        dev_index = -1
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the decode path has no CPU fallback")
        ndev = torch.cuda.device_count()
        if local_rank >= ndev and not args.share_gpu:
            raise SystemExit("rank %d has no GPU of its own (%d visible); --share-gpu --backend gloo rehearses on fewer" % (local_rank, ndev))
        if args.share_gpu and backend == "nccl" and world > ndev:
            raise SystemExit("--share-gpu needs --backend gloo (RCCL refuses two ranks on one device)")
        dev_index = local_rank % ndev
        torch.cuda.set_device(dev_index)
        device = torch.device("cuda", dev_index)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend="gloo")
    red_device = device if backend == "nccl" else torch.device("cpu")

    lib = pyabi.load()  # code / configuration helpers are host-only; create() below needs the GPU
    code = pyabi.Code50GPON(lib)
    cfg = pyabi.default_cfg(args.method, args.max_iter, lib)
    if args.max_bf is not None:
        cfg.max_bf_iter = args.max_bf
    if args.factor_1 is not None:
        cfg.factor_1 = args.factor_1
    if args.factor_2 is not None:
        cfg.factor_2 = args.factor_2
    if selftest:
        dec = SelftestDecoder(oa, code, cfg)
    else:
        dec = pyabi.Decoder(code, cfg, device=dev_index, max_groups=args.groups, lib=lib)

    # ---- how the four error counters are summed over the ranks (the path's only exchange, reference main.cpp:174-182) --------
    # N > 1 with one GPU per rank: the PRODUCT's reduction - rank 0 makes an RCCL id (lnsfaid_comm_unique_id), the 128 bytes travel
    # through the torch.distributed store, every rank joins with lnsfaid_comm_init, and every step calls
    # lnsfaid_allreduce_counters (one ncclAllReduce of 4 x uint64 on the decoder's stream).  torch.distributed stays for the
    # barrier, the MAX over the ranks' times and the gather of per-rank lines.  Ranks that share a GPU (--share-gpu rehearsal)
    # and the CPU self-test cannot form an RCCL communicator (one rank per device) and sum over gloo; N = 1 has nothing to sum.
    reduce_via = "none (world size 1: no collective)"
    if dist is not None and backend == "nccl" and not selftest:
        # (also for a world of one under the launcher: still the product's RCCL path, on a communicator of one rank)
        import ctypes
        ok, why = 1, ""
        try:
            cid = (ctypes.c_uint8 * 128)()
            if rank == 0:
                rc = lib.lnsfaid_comm_unique_id(cid)
                if rc != 0:
                    raise RuntimeError("lnsfaid_comm_unique_id: %d" % rc)
            if world > 1:
                box = [bytes(cid) if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                cid = (ctypes.c_uint8 * 128).from_buffer_copy(box[0])
            dec.comm_init(world, rank, cid)
        except Exception as e:  # noqa: BLE001
            ok, why = 0, repr(e)
        # every rank must take the same path: agree over the torch process group
        flag = torch.tensor([ok], dtype=torch.int32, device=red_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            reduce_via = "lnsfaid_allreduce_counters"
        else:
            lib.lnsfaid_comm_destroy(dec.ctx)
            reduce_via = "torch.distributed nccl (lnsfaid_comm_init failed on some rank%s)" % (": " + why if why else "")
    elif dist is not None and world > 1:
        reduce_via = "torch.distributed gloo"

    def reduce_counters(local):
        if reduce_via == "lnsfaid_allreduce_counters":
            return [int(c) for c in dec.allreduce_counters(local)]
        if reduce_via.startswith("torch.distributed"):
            return lnsfaid_dist.allreduce_counters(local, dist, red_device)
        return [int(c) for c in local]

    def sync():
        if not selftest:
            torch.cuda.synchronize()

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    def run_point(eb_n0, steps, warmup, n_groups, seed):
        """`steps` timed passes over `n_groups` groups resident on this rank's GPU."""
        if selftest:
            fix = oa.synth_llr(n_groups, N_VAR, eb_n0, seed, args.scale)
            d_fix = d_out = None
            d_stats = None
        else:
            d_fix = synth_llr(torch, device, n_groups, eb_n0, seed, args.mod_type, args.scale)
            d_out = torch.empty((n_groups, 32 * N_VAR), dtype=torch.int8, device=device)
            d_stats = torch.zeros((n_groups, 2), dtype=torch.int32, device=device)
            torch.cuda.synchronize()
        totals, local = None, None
        host_stats = None

        def step():
            nonlocal totals, local, host_stats
            if selftest:
                out, host_stats = dec.decode(fix, n_groups)
                local = dec.count(out, n_groups)
            else:
                dec.decode_device(d_fix.data_ptr(), n_groups, d_out.data_ptr(), d_stats.data_ptr())
                local = dec.count_errors_device(d_out.data_ptr(), None, n_groups)
            totals = reduce_counters(local)  # the path's only exchange (reference main.cpp:174-182)

        for _ in range(warmup):
            step()
        barrier()
        dec.kernel_time(reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        k_ms, k_launches = dec.kernel_time(reset=True)
        per_rank = [local]
        per_rank_ms = [round(dt / steps * 1e3, 4)]
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=red_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)  # the slowest rank's time
            gathered = [None] * world
            dist.all_gather_object(gathered, ([int(c) for c in local], round(dt / steps * 1e3, 4)))
            per_rank = [g[0] for g in gathered]
            per_rank_ms = [g[1] for g in gathered]
            dt = float(t.item())
        stats = host_stats if selftest else d_stats.cpu().numpy()
        alg_bytes = float(sum(32 * algorithmic_bytes(int(i), int(j)) for i, j in stats)) * steps
        return dict(eb_n0=eb_n0, dt=dt, steps=steps, kernel_ms=k_ms, launches=k_launches, alg_bytes=alg_bytes,
                    mean_I=float(stats[:, 0].mean()), mean_J=float(stats[:, 1].mean()), counters=totals,
                    per_rank_counters=per_rank, per_rank_ms=per_rank_ms, n_groups=n_groups,
                    rows_per_lane=(0 if selftest else dec.rows_per_lane()),
                    message_store=(0 if selftest else dec.message_store()),
                    waves=(1 if selftest else dec.kernel_waves()))

    # ---- weak leg = headline ---------------------------------------------------------------------------------------
    n_groups = args.groups
    n_cw = n_groups * 32
    head = run_point(args.eb_n0, args.steps, args.warmup, n_groups, 1234 + 7919 * rank)
    info_bits = float(world) * n_cw * K_INFO * args.steps
    value = info_bits / head["dt"] / 1e9
    avg_launch_ms = head["kernel_ms"] / max(1, head["launches"])
    alg_gbs = head["alg_bytes"] / (head["kernel_ms"] * 1e-3) / 1e9

    # ---- roofline: VALU issue (the binding resource) ------------------------------------------------
    # The instruction count comes from a separate rocprofv3 --pmc SQ_INSTS_VALU pass (tools/gpu_pmc_sq.sh ->
    # profiles/valu_issue_per_launch.json): it is REPLAYED here, stamped with the kernel source hash it was taken
    # at, and dropped (null) when the kernel has changed since or the workload is not the one it was counted on.
    # A counter file is replayed only for what it was measured on: the same kernel SOURCE (hash), the same kernel INSTANCE (the
    # name the library reports for this context: rows per lane, DecodeMethod, where the messages live), the default BUILD (no
    # EXTRA experiment flags in lnsfaid_version) and the headline WORKLOAD (incl. the quantiser scale).
    here_hash = kernel_source_hash()
    library = "selftest" if selftest else lib.lnsfaid_version().decode()
    kernel_name = None if selftest else kernel_instance_name(args.method, head["rows_per_lane"], head["message_store"], pyabi, head["waves"])
    valu_inst, valu_src, valu_half_frac = None, None, None
    vpath = os.path.join(ROOT, "profiles", "valu_issue_per_launch.json")
    headline_workload = (args.method == 2 and abs(args.eb_n0 - 3.0) < 1e-6 and args.groups == 2048 and args.max_iter == 10
                         and args.max_bf is None and args.factor_1 is None and args.factor_2 is None
                         and args.mod_type == 2 and abs(args.scale - 13.0) < 1e-6 and not selftest
                         and "[" not in library)

    def replayable(rec):
        return (rec.get("kernel_source_hash") == here_hash and rec.get("kernel_instance") == kernel_name
                and rec.get("library") == library)

    if os.path.exists(vpath) and headline_workload:
        try:
            v = json.load(open(vpath))
            if replayable(v):
                valu_inst = float(v["valu_instructions_per_launch"])
                valu_src = v.get("source")
                valu_half_frac = v.get("valu_half_rate_fraction")
        except Exception:
            valu_inst = None
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic_per_launch.json")
    if os.path.exists(tpath) and headline_workload:
        try:
            t = json.load(open(tpath))
            if replayable(t):
                traffic = t.get("bytes_per_launch")
        except Exception:
            traffic = None
    ach_ginstr = valu_inst / (avg_launch_ms * 1e-3) / 1e9 if valu_inst else None
    # work-normalised figures: they only improve when the decoder gets faster, not when a change adds instructions
    simd_cycles_per_launch = avg_launch_ms * 1e-3 * CLOCK_GHZ * 1e9 * N_SIMD
    edge_updates_per_launch = float(n_cw) * head["mean_I"] * N_EDGES * args.steps / max(1, head["launches"])
    valu_half = valu_inst * valu_half_frac if (valu_inst and valu_half_frac is not None) else None
    valu_full = valu_inst - valu_half if valu_half is not None else None

    result = {
        "metric": "decoded Gb/s @ 10 iters, 50G-PON LDPC; FER match vs AVX512 ref",
        "value": round(value, 4),
        "unit": "Gb/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(head["dt"] / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "i8",
        "data": "synthetic",
        "config": {
            "workload": "50G-PON N=17664 K=14592 Z=256, %s all-zero codeword + AWGN, DecodeMethod=%d (%s), "
                        "MaxIteration=%d, scale %g, %d codewords (%d groups of 32) per GPU, Eb/N0 %.1f dB"
                        % ({2: "QPSK", 4: "16-QAM"}[args.mod_type], args.method,
                           {0: "NMS", 1: "OMS", 2: "3-bit LNS-FAID FAID3 + DTBF", 3: "OMS + BF", 4: "OMS + DTBF", 5: "FAID + 2B1C"}[args.method],
                           args.max_iter, args.scale, n_cw, n_groups, args.eb_n0),
            "eb_n0_db": args.eb_n0,
            "mean_layered_iterations": head["mean_I"],
            "mean_bf_iterations": head["mean_J"],
            "parallelism": ("one process per GPU, whole groups per rank, no data-path collective; the 4 error counters are summed "
                            "once per step by %s (world size %d)" % (reduce_via, world)) if world > 1 or reduce_via != "none (world size 1: no collective)"
                           else "one GPU, one process: no collective ran (the counters of the one rank are the totals)",
            "world_size": world,
            "reduce": reduce_via,
            "backend": backend if dist is not None else "none (no process group at N = 1 without the launcher)",
            "counters_TestFrame_ErrorFrame_ErrorBits_LT3": head["counters"],
            "per_rank_counters": head["per_rank_counters"],
            "per_rank_ms_per_step": head["per_rank_ms"],
        },
        "roofline": {
            "bound": "valu-issue",
            "achieved": round(ach_ginstr, 2) if ach_ginstr else None,
            "peak": VALU_PEAK_GINSTR,
            "unit": "G wave64 VALU instructions/s",
            "frac": round(ach_ginstr / VALU_PEAK_GINSTR, 4) if ach_ginstr else None,
            "traffic": traffic,
            "kernel": ("lnsfaid_decode5_kernel<%d> (EXPERIMENTAL: two waves per codeword, four check rows per lane, compressed messages streamed through HBM)"
                       % args.method) if head["waves"] == 2
                      else ("lnsfaid_decode4_kernel<%d, %s> (one wave per codeword, four check rows per lane, compressed messages %s)"
                       % (args.method, "true, false" if head["message_store"] == 1 else "false, false",
                          "in registers" if head["message_store"] == 1 else "streamed through HBM")) if head["rows_per_lane"] == 4
                      else "lnsfaid_decode_kernel<%d> (128 threads per codeword, two check rows per lane)" % args.method,
            "launches": head["launches"],
            "avg_launch_ms": round(avg_launch_ms, 4),
            "valu_instructions_per_launch": valu_inst,
            "simd_cycles_per_edge_update": round(simd_cycles_per_launch / edge_updates_per_launch, 4) if edge_updates_per_launch else None,
            "edge_updates_per_launch": edge_updates_per_launch,
            "valu_full_rate": valu_full,
            "valu_half_rate": valu_half,
            "valu_busy_frac": round((2.0 * valu_full + 4.0 * valu_half) / simd_cycles_per_launch, 4) if valu_half is not None else None,
            "kernel_instance": kernel_name,
            "library": library,
            "kernel_source_hash": here_hash,
            "replayed_from_profiles": {
                "valu_instructions_per_launch": valu_src if valu_inst else None,
                "traffic": "profiles/hbm_traffic_per_launch.json" if traffic else None,
            },
            "hbm_counter_frac": round(traffic / (avg_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
            "hbm_counter_GBs": round(traffic / (avg_launch_ms * 1e-3) / 1e9, 1) if traffic else None,
            "compulsory_io_GBs": round(2.0 * N_VAR * n_cw * args.steps / (head["kernel_ms"] * 1e-3) / 1e9, 1),
            "algorithmic_GBs": round(alg_gbs, 2),
            "algorithmic_bytes_per_launch": head["alg_bytes"] / max(1, head["launches"]),
            "algorithmic_over_hbm_peak": round(alg_gbs / HBM_PEAK_GBS, 4),
            "note": "measured live: avg_launch_ms (HIP events on the decoder's stream), launches, algorithmic_*; replayed from "
                    "profiles/ (separate rocprofv3 --pmc passes on this exact kernel source, null when the source hash "
                    "differs): valu_instructions_per_launch, traffic.  peak = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU "
                    "instruction.  `frac` is issued instructions over that peak, so it rises when a change ADDS instructions; the "
                    "work-normalised figures do not: simd_cycles_per_edge_update = launch time x 2.4 GHz x 1024 SIMDs / (codewords x "
                    "layered iterations x 70 400 edges), measured live; valu_busy_frac = (2 x valu_full_rate + 4 x valu_half_rate) / "
                    "SIMD cycles of the launch, the split of the counted instructions into the two issue classes taken from the "
                    "compiled layer step (tools/parse_pmc_sq.py).  The kernel keeps En in LDS and the compressed messages (6 bytes per check row) on chip, so it moves far fewer HBM "
                    "bytes than the reference layout's algorithmic figure (SURVEY.md 8(d): 2N + I(4E+N) + J*2N per codeword); "
                    "algorithmic_over_hbm_peak above 1 is therefore not an HBM saturation claim and is not `frac`",
        },
    }
    if selftest:
        result["data"] = "launcher self-test on CPU (oracle port as stand-in) - NOT a measurement"
        result["invalid_for_measurement"] = True

    # ---- strong leg: the same number of groups in total, split into contiguous ranges of whole groups ---------------
    if world > 1 and not args.no_strong:
        first, last = lnsfaid_dist.shard_groups(args.groups, rank, world)
        if last > first:
            sp = run_point(args.eb_n0, args.steps, 1, last - first, 4321 + 7919 * rank)
        else:  # more ranks than groups: this rank only joins the collectives
            sp = None
            for _ in range(1 + args.steps):
                reduce_counters([0, 0, 0, 0])
            barrier(); barrier()
            t = torch.tensor([0.0], dtype=torch.float64, device=red_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            gathered = [None] * world
            dist.all_gather_object(gathered, ([0, 0, 0, 0], 0.0))
        if rank == 0 and sp is not None:
            result["strong"] = {
                "value": round(args.groups * 32 * K_INFO * args.steps / sp["dt"] / 1e9, 4), "unit": "Gb/s",
                "total_groups": args.groups, "groups_on_rank0": sp["n_groups"],
                "ms_per_step": round(sp["dt"] / args.steps * 1e3, 4),
                "counters_TestFrame_ErrorFrame_ErrorBits_LT3": sp["counters"],
                "per_rank_counters": sp["per_rank_counters"],
                "per_rank_ms_per_step": sp["per_rank_ms"],
            }

    if rank == 0 and world == 1 and not args.no_points and not selftest:
        pts = []
        for eb in ((3.6, 4.2) if args.mod_type == 2 else (8.1, 8.6)):
            torch.cuda.empty_cache()
            p = run_point(eb, max(2, args.steps // 3), 1, n_groups, 1234)
            pts.append({"eb_n0_db": eb, "value": round(n_cw * K_INFO * p["steps"] / p["dt"] / 1e9, 4), "unit": "Gb/s",
                        "mean_layered_iterations": p["mean_I"], "mean_bf_iterations": p["mean_J"],
                        "launches_per_step": p["launches"] / p["steps"],
                        "algorithmic_GBs": round(p["alg_bytes"] / (p["kernel_ms"] * 1e-3) / 1e9, 2)})
        result["points"] = pts

    if rank == 0 and world == 1 and not args.no_cpu and not selftest:
        # cpu_baseline: the AVX2 port (oracle/, test infrastructure - imported here and nowhere else in the GPU legs) on ALL host
        # cores this process may use, one group stream per thread, at the three Eb/N0 points.  Per point the sample is the whole
        # batch the GPU has just decoded (every frame compared with the GPU's), and the pass is repeated until the point has
        # run for at least --cpu-seconds of wall time.
        oa = load_oracle_abi()
        threads = max(1, min(host_cores(), args.cpu_threads or 1 << 30))
        ng = min(args.cpu_groups, n_groups)
        d_out = torch.empty((n_groups, 32 * N_VAR), dtype=torch.int8, device=device)
        d_stats = torch.zeros((n_groups, 2), dtype=torch.int32, device=device)
        samples = []
        for eb in ((args.eb_n0, 3.6, 4.2) if args.mod_type == 2 else (args.eb_n0,)):
            torch.cuda.empty_cache()
            d_fix = synth_llr(torch, device, n_groups, eb, 1234, args.mod_type, args.scale)
            torch.cuda.synchronize()  # the decoder runs on its own stream
            dec.decode_device(d_fix.data_ptr(), n_groups, d_out.data_ptr(), d_stats.data_ptr())
            torch.cuda.synchronize()
            fix_host = d_fix[:ng].cpu().numpy().reshape(-1)
            gpu_dec = d_out[:ng].cpu().numpy().reshape(-1)
            gpu_stats = d_stats[:ng].cpu().numpy()
            del d_fix
            passes, dt, parity = 0, 0.0, True
            while passes == 0 or (dt < args.cpu_seconds and passes < 64):
                t0 = time.perf_counter()
                cpu_dec, cpu_stats = oa.decode_mt(code, cfg, fix_host, ng, threads=threads, kind="avx2")
                dt += time.perf_counter() - t0
                if passes == 0:
                    parity = bool(np.array_equal(cpu_dec, gpu_dec) and np.array_equal(cpu_stats, gpu_stats))
                passes += 1
            samples.append({"eb_n0_db": eb, "value": round(passes * ng * 32 * K_INFO / dt / 1e9, 5), "unit": "Gb/s", "wall_s": round(dt, 2),
                            "passes": passes, "cpu_work_s": round(dt * threads, 1), "parity_with_gpu": parity})
        result["cpu_baseline"] = {
            "value": samples[0]["value"],
            "unit": "Gb/s",
            "cores": threads,
            "host_cpu_count": os.cpu_count(),
            "cpu_model": cpu_model(),
            "kind": "port",
            "sample": "per Eb/N0 point %d groups (%d codewords: %s batch the GPU decoded), decoded by oracle/lnsfaid_cpu_avx2.c on %d "
                      "host threads (every core this process may run on; one group stream per thread), the pass repeated until the "
                      "point has run >= %.1f s of wall time; `value` is the %.1f dB point"
                      % (ng, ng * 32, "the whole" if ng == n_groups else "the first groups of the", threads, args.cpu_seconds, args.eb_n0),
            "points": samples,
            "parity_with_gpu": all(s["parity_with_gpu"] for s in samples),
            "note": "the port is an upper bound for the reference on this host: AVX2 with 32 codewords per 256-bit register like "
                    "the reference, but the FAID table is one pshufb where the reference emulates it with nine masked adds per "
                    "edge (CDecoder_FAID.cpp:710-851) and the dead flip_vote work is not done; the reference's own AVX-512 "
                    "build is not possible here (it needs Intel MKL's mkl.h)",
        }

    if rank == 0 and world == 1 and not args.no_dropin and not selftest and args.method in (1, 2, 5):
        # The decoder in the reference's OWN call shape (not `value`): T host threads, one context each, ONE group of 32 frames
        # per lnsfaid_decode call on pageable host buffers - what an unmodified CSimulate::Run does through the binding of
        # INTEGRATION.md section 2 (reference CSimulate.cpp:136-164, main.cpp:164-172).  host/dropin_bench is run as a child
        # process on the same GPU after the batched measurement; PCIe copies and every host-side cost are inside its clock.
        exe = os.path.join(PKG_DIR, "host", "dropin_bench")
        rows = []
        if os.path.exists(exe):
            for t in (1, 8, 64):
                for reg in ([], ["--register"]):
                    try:
                        p = subprocess.run([exe, "--threads", str(t), "--calls", "40", "--eb-n0", str(args.eb_n0), "--method", str(args.method),
                                            "--max-iter", str(args.max_iter), "--device", str(dev_index)] + reg,
                                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=120)
                        if p.returncode == 0 and p.stdout.strip().startswith("{"):
                            rows.append(json.loads(p.stdout.strip().splitlines()[-1]))
                    except Exception:
                        pass
        result["dropin"] = {
            "what": "T host threads x one lnsfaid context each x ONE group of 32 frames per lnsfaid_decode call, host buffers in and out "
                    "(pageable = plain malloc as the reference's CLDPC::Initial allocates them; registered = lnsfaid_host_register); "
                    "aggregate_Gbps includes PCIe and every host-side cost; the headline `value` is the batched device-resident rate",
            "unit": "Gb/s",
            "runs": rows,
            "best_aggregate_Gbps": max([r["aggregate_Gbps"] for r in rows], default=None),
        }

    dec.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not under_launcher and (args.gpus > 1 or args.spawn):
        return launch_children(args, argv)
    return worker(args)


if __name__ == "__main__":
    sys.exit(main())
