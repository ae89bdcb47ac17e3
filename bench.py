#!/usr/bin/env python3
"""bench.py -- throughput of the partitioned-FIR hot path on MI355X.

Workload (BASELINE.json configs[2], the headline): 8 channels, 131072-tap IR,
4096-sample partitions (N = 8192, B = 32), fp32, synthetic uniform noise.
One STEP = one bfir_engine_run_device call over `--blocks` consecutive blocks
of input that is already resident in HBM (generated there).  Metric: output
channel-samples/s.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1: one process per GPU.  Under torch.distributed.run (WORLD_SIZE set) this
process is one rank; started plainly with --gpus N it spawns the N ranks itself
(fresh child processes, before anything touches a GPU) and relays rank 0's line.
  --shard replicas  (default) every rank runs its own independent 8-channel
                    stream: weak scaling, the units are streams
  --shard channels  ONE 8-channel stream, rank r owns channels r*C/N .. : strong
                    scaling (channels never mix, brutefir/brutefir.cpp:252-334)
  --streams S       S independent engines dealt out to the ranks (configs[3])
No rank ever exchanges audio with another; the only communication is the MAX /
SUM reduction of the timing.

The JSON line also carries
  roofline      per kernel: algorithmic HBM bytes / mean launch time (HIP events
                on the launch stream, inside the timed region) against 8 TB/s, the
                same from an untimed serial-schedule pass (*_exclusive), the
                PMC-measured traffic when profiles/traffic_*.json matches this
                very csrc/ (sha stamped), and the pipeline-level traffic / time
  cpu_baseline  the CPU oracle (a port of the reference algorithm) on a bounded
                sample of the same workload on this host: 1 thread and all cores
  parity_rel_err_vs_oracle   sampled blocks of the TIMED output buffer
  end_to_end    the host-pointer entry (bfir_engine_run, PCIe inclusive)
  latency_one_block_us       one run() per block, the plug-in's call pattern
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)

WORKLOADS = {
    # name: (channels, taps, L, realsize)
    "cfg3_8ch_131072tap_L4096_fp32": (8, 131072, 4096, 4),
    "cfg2_2ch_65536tap_L8192_fp32": (2, 65536, 8192, 4),
    "cfg5_2ch_262144tap_L4096_fp64": (2, 262144, 4096, 8),
    "cfg4_stereo_65536tap_L4096_fp32": (2, 65536, 4096, 4),   # per stream; use with --streams
    # the headline's shape with 24 partitions (a count between two of k_mac_stream's register batches)
    "hl_8ch_98304tap_L4096_fp32": (8, 98304, 4096, 4),
    # not BASELINE configs: the plug-in's own partition size (FILTER_LEN 1024)
    "plugin_8ch_131072tap_L1024_fp32": (8, 131072, 1024, 4),
    "plugin_8ch_65536tap_L1024_fp32": (8, 65536, 1024, 4),
    "plugin_8ch_49152tap_L1024_fp32": (8, 49152, 1024, 4),
    "plugin_8ch_98304tap_L1024_fp32": (8, 98304, 1024, 4),
    "plugin_2ch_65536tap_L1024_fp64": (2, 65536, 1024, 8),    # the shipped REALSIZE 8 (common.h:17-19)
    "plugin_2ch_65536tap_L1024_fp32": (2, 65536, 1024, 4),
    # ... with the frames the plug-in really hands over: FLOAT_LE (32-bit) in and out around fp64 arithmetic
    # (foo_dsp_bfir.cpp:279-289); fifth field = frame sample bytes
    "plugin_2ch_65536tap_L1024_fp64_f32frames": (2, 65536, 1024, 8, 4),
    # ... with longer impulses (the plug-in cuts whatever file it is given into 1024-sample partitions,
    # foo_dsp_bfir.cpp:275-276): 48 / 96 / 128 / 192 / 256 partitions = 1.1 ... 5.9 s at 44.1 kHz
    "plugin_2ch_49152tap_L1024_fp64_f32frames": (2, 49152, 1024, 8, 4),
    "plugin_2ch_98304tap_L1024_fp64_f32frames": (2, 98304, 1024, 8, 4),
    "plugin_2ch_131072tap_L1024_fp64_f32frames": (2, 131072, 1024, 8, 4),
    "plugin_2ch_196608tap_L1024_fp64_f32frames": (2, 196608, 1024, 8, 4),
    "plugin_2ch_262144tap_L1024_fp64_f32frames": (2, 262144, 1024, 8, 4),
}


def csrc_sha():
    """Identity of the kernels a PMC traffic file belongs to: sha256 over csrc/ sources."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "foo-dsp-bfir_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode()); h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def algorithmic_bytes_per_block(C, B, N, L, s, fused_io=False):
    """bytes one run() block must move, split by the kernel that moves them.
    fused_io: the pair path has no staging kernels -- k_fwd reads the input block itself, k_inv
    writes the output block; the sum stays C*s*(2BN + N + 2L)."""
    return {
        "k_stage_in": 0 if fused_io else C * s * L,   # input block
        "k_fwd": C * s * (N + L) if fused_io else C * s * N,   # new delay-line slot (+ the input block)
        "k_mac": C * s * 2 * B * N,                   # all partition spectra + all delay-line spectra
        "k_inv": C * s * L if fused_io else 0,
        "k_stage_out": 0 if fused_io else C * s * L,  # output block
    }


def measured_traffic_table(workload, chunk):
    """Per-kernel HBM bytes per launch from a committed PMC pass (profiles/traffic_*.json, written by
    scripts/summarize_pmc.py from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this command).
    A file is used only when it was taken on THIS csrc/ (meta.csrc_sha), this workload and this
    blocks-per-launch; anything else would be a stale number, so it yields None."""
    import glob
    sha = csrc_sha()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_*.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        m = d.get("meta", {})
        if m.get("workload") == workload and int(m.get("chunk", -1)) == int(chunk) and m.get("csrc_sha") == sha:
            return {k: v["total_bytes"] for k, v in d["per_launch"].items()}, os.path.basename(path)
    return None, None


def measured_stream_peaks():
    """Best read / write / copy GB/s of scripts/ubench/hbm_stream2.hip on this GPU model
    (profiles/r03_hbm_stream.txt; SURVEY 8(d) asks for the measured stream peak next to the 8 TB/s spec).
    copy_GBs counts read + written bytes of a 1 : 1 mix -- the mix the convolution pipeline moves."""
    import re
    path = os.path.join(ROOT, "profiles", "r03_hbm_stream.txt")
    best = {}
    try:
        for line in open(path):
            m = re.match(r"^(read|write|copy)[, ].*?(\d+) GB/s\s*$", line)
            if m:
                key = m.group(1) + "_GBs"
                best[key] = max(best.get(key, 0), int(m.group(2)))
    except OSError:
        return None
    if best:
        best["source"] = "profiles/r03_hbm_stream.txt (contiguous slab per workgroup; grid-stride loops reach 4.6-4.8 copy)"
    return best or None


def flops_per_block(C, B, N):
    """Arithmetic of one run() block: 8 flops per complex multiply-add over C * B * N/2 bins, and a forward and an
    inverse real transform per channel at 2.5 N log2 N (two channels share one N-point complex transform on the
    pair path: 5 N log2 N per pair and direction -- the same count)."""
    import math
    return {"mac": 8.0 * C * B * (N // 2), "fft": 2 * C * 2.5 * N * math.log2(N)}


FP32_VECTOR_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md; fp64 vector: half of it
# bytes a three-kernel pipeline (fwd | mac | inv, spectra handed over through memory) cannot avoid, per block:
# input in, delay-line spectrum out + in, product spectrum out + in, output out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return ""


def cpu_baseline(O, C, taps, L, B, s, h, x, budget_s=12.0):
    """Time the oracle (1 thread) on a bounded sample: B warm-up blocks, then blocks until ~budget."""
    eng = O.Engine(L, B, s, C)
    assert eng.set_coeff(h) == 0
    nb_avail = x.shape[0] // L
    warm = min(B, nb_avail)
    eng.run(x[:warm * L])
    n, dt = 0, 0.0
    while dt < budget_s:       # whole passes over the sample (the stream simply continues)
        t0 = time.perf_counter()
        rc, _ = eng.run(x[:nb_avail * L])
        dt += time.perf_counter() - t0
        n += nb_avail
        assert rc == 0
    fft = O.fft_backend() if hasattr(O, "fft_backend") else "own CPU FFT (FFTW unavailable)"
    out = {"value": n * L * C / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port", "fft": fft,
           "cpu_model": cpu_model(),
           "sample": "%d blocks (%d channel-samples) after %d warm-up blocks, oracle/bfir_oracle.c, "
                     "1 thread, %.1f s" % (n, n * L * C, warm, dt)}
    # also all host cores.  The reference is single-threaded per instance, so the
    # all-cores arrangement is replicated engines: one single-channel engine per thread, thread i
    # on channel i mod C of the same sample.  ctypes releases the GIL inside the oracle.
    try:
        from concurrent.futures import ThreadPoolExecutor
        avail = len(os.sched_getaffinity(0))
        threads = max(1, min(avail, int(os.environ.get("BFIR_CPU_THREADS", "64"))))
        nb_par = min(nb_avail, 2048)
        engs, cols = [], []
        for i in range(threads):
            e1 = O.Engine(L, B, s, 1)
            assert e1.set_coeff([h[i % C]]) == 0
            engs.append(e1)
            cols.append(np.ascontiguousarray(x[:nb_par * L, (i % C):(i % C) + 1]))

        def one(i):
            engs[i].run(cols[i][:warm * L])
            return engs[i].run(cols[i][:nb_par * L])[0]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=threads) as pool:
            rcs = list(pool.map(one, range(threads)))
        dtp = time.perf_counter() - t0
        assert all(r == 0 for r in rcs)
        out["all_cores"] = {"value": threads * (nb_par + warm) * L / dtp / 1e6, "unit": "Msamples/s",
                            "cores": threads, "host_cores_available": avail,
                            "sample": "%d single-channel engines (channel i mod %d) on %d threads, %d blocks "
                                      "each, %.1f s" % (threads, C, threads, nb_par + warm, dtp)}
    except Exception as exc:   # the extra row must never cost the bench line
        out["all_cores"] = {"error": repr(exc)}
    return out


# ---------------------------------------------------------------------------
# N > 1 without a launcher: spawn the ranks ourselves
# ---------------------------------------------------------------------------
def spawn_ranks(n, argv):
    """Start n fresh rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), relay
    rank 0's stdout, return the worst exit code.  The parent never initialises a GPU and never re-execs."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    for line in out.decode().splitlines():          # stdout carries the JSON line only; library chatter (gloo) goes to stderr
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def spawn_check(args):
    """CPU rehearsal of the launcher plumbing (tests/test_bench_spawn.py): rendezvous over gloo, the two
    timing reductions, one JSON line from rank 0.  No GPU, no engine, not a measurement."""
    import torch.distributed as dist
    from foo_dsp_bfir_amd import sharding
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    worst = sharding.max_over_ranks(1.0 + rank)
    total = sharding.sum_over_ranks(10.0)
    lo, hi = sharding.shard_range(8, rank, world)
    chans = sharding.sum_over_ranks(hi - lo)
    # configs[3]: 256 stereo streams dealt out to the ranks; the widest share bounds a rank's memory
    slo, shi = sharding.shard_range(256, rank, world)
    streams = sharding.sum_over_ranks(shi - slo)
    widest = sharding.max_over_ranks(shi - slo)
    most_ch = sharding.max_over_ranks(hi - lo)
    if rank == 0:
        print(json.dumps({"spawn_check": True, "n_gpus": world, "max_elapsed": worst, "sum_units": total,
                          "channels_covered": chans, "channels_rank0": hi - lo, "channels_max": most_ch,
                          "streams_covered": streams, "streams_rank0": shi - slo, "streams_max": widest}), flush=True)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=0,
                    help="blocks per step (one run_device call; the chunk pipeline drains at call boundaries); "
                         "0 = sized so that 20 steps time at least a second (headline: 163840 = 40 launches)")
    ap.add_argument("--chunk", type=int, default=int(os.environ.get("BFIR_CHUNK", "0")),
                    help="blocks per kernel launch (0 = the engine's automatic choice: 4096, fewer for many channels)")
    ap.add_argument("--workload", default="cfg3_8ch_131072tap_L4096_fp32", choices=sorted(WORKLOADS))
    ap.add_argument("--shard", default="replicas", choices=["replicas", "channels"],
                    help="N > 1: replicas = one independent stream per rank (weak scaling); channels = one "
                         "stream, its channels dealt out to the ranks (strong scaling)")
    ap.add_argument("--streams", type=int, default=0,
                    help="total independent engines dealt out to the ranks (0 = one per rank); "
                         "configs[3]: --workload cfg4_stereo_65536tap_L4096_fp32 --streams 256 --blocks 1024")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="no oracle at all: neither the CPU timing nor the parity check")
    ap.add_argument("--no-cpu-timing", action="store_true", help="keep the parity check of the timed output, skip the CPU timing")
    ap.add_argument("--no-extras", action="store_true", help="skip end_to_end and latency_one_block_us")
    ap.add_argument("--no-exclusive-pass", action="store_true",
                    help="skip the untimed serial-schedule pass (rocprofv3 runs: keeps the kernel trace to the overlapped schedule)")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not bracket kernels with HIP events in the timed region")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for the timing reduction (nccl = RCCL; gloo lets several "
                         "ranks rehearse the N > 1 path on one GPU together with --device)")
    ap.add_argument("--device", type=int, default=-1, help="HIP device ordinal (default: LOCAL_RANK)")
    ap.add_argument("--channels", type=int, default=0, help="override the workload's channel count (1..8); the line's config says so")
    ap.add_argument("--spawn-check", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def auto_chunk(n_ch, N, B, s):
    """The engine's automatic blocks-per-launch (csrc/engine.hip ensure_chunk): the work of 4096 headline-shaped
    blocks (4096 ... 32768 blocks), fewer when the delay line of that many blocks would pass 4 GiB."""
    slots = (4 << 30) // (n_ch * N * s)
    want = max(4096, min(32768, (8 * 4096 * 4096) // (n_ch * (N // 2))))
    return int(max(16, min(want, (slots - B) // 2)))


def default_blocks(n_eng, C, L, B, s, chunk):
    """Blocks per step such that 20 steps time at least a second on one MI355X (round-1 speeds):
    the resident job grows, the launch geometry (blocks per launch) does not.  At most 32 GiB per
    I/O buffer and rank."""
    gsps = (95.0 if s == 4 else 22.0) * (32.0 / B if B > 32 else 1.0)      # rough Gsamples/s, profiles/r01_*
    want = 0.1 * gsps * 1e9 / (n_eng * C * L)                              # ~100 ms per step
    cap = (32 << 30) // (n_eng * C * L * s)
    nb = int(min(want, cap)) // chunk * chunk
    return max(chunk, nb)


def mac_range(tc, N, n_ch, B):
    """Blocks per range of the streaming MAC for a launch of tc blocks (csrc/kernels.hip
    launch_mac_stream); 0 when the LDS-shared kernel runs instead (B > 32)."""
    if B > 32:
        return 0
    PB = 4 if B <= 4 else 8 if B <= 8 else 16 if B <= 16 else 32
    want = max(1, 512 // max(1, (N // 2 // 256) * n_ch))
    return max(1, -(-tc // (want * PB))) * PB


def main():
    args = parse_args()
    argv = sys.argv[1:]
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, argv))
    if args.spawn_check:
        return spawn_check(args)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) if args.device < 0 else args.device
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", local)
    red_dev = dev if args.dist_backend == "nccl" else None      # gloo reduces CPU tensors

    import foo_dsp_bfir_amd as bfir   # raises if the HIP library is missing
    from foo_dsp_bfir_amd import sharding

    C_all, taps, L, s = WORKLOADS[args.workload][:4]
    if args.channels > 0:
        C_all = args.channels          # the workload's shape with another channel count (tuning aid / odd-channel-count runs)
    fb = WORKLOADS[args.workload][4] if len(WORKLOADS[args.workload]) > 4 else s     # bytes per frame sample
    fmt = 8 if fb == 4 else 10                                                          # BF_SAMPLE_FORMAT_FLOAT_LE / FLOAT64_LE
    N, B = 2 * L, (taps + L - 1) // L
    rdt = np.float32 if s == 4 else np.float64
    tdt = torch.float32 if fb == 4 else torch.float64
    # Units.  replicas: independent streams, one per rank (weak).  channels: the channels of ONE
    # stream (strong).  --streams S: S engines dealt out to the ranks (strong, configs[3]).
    ch_lo, ch_hi = 0, C_all
    if args.streams > 0:
        lo, hi = sharding.shard_range(args.streams, rank, world)
        n_eng, first_stream, scaling, mode = hi - lo, lo, "strong", "streams dealt out to ranks"
    elif args.shard == "channels" and world > 1:
        ch_lo, ch_hi = sharding.shard_range(C_all, rank, world)
        n_eng, first_stream, scaling, mode = (1 if ch_hi > ch_lo else 0), 0, "strong", "channels of one stream dealt out to ranks"
    else:
        n_eng, first_stream, scaling, mode = 1, rank, "weak", "one independent stream per rank"
    C = ch_hi - ch_lo
    chunk = args.chunk if args.chunk > 0 else auto_chunk(max(1, n_eng * C), N, B, s)
    nb = args.blocks if args.blocks > 0 else default_blocks(max(1, n_eng), C_all, L, B, s, chunk)
    chunk = min(chunk, nb)

    # synthetic audio / IR of SURVEY.md 8(d): IR = uniform[-1,1) * exp(-6 n / taps), sum|h| = 1 (host,
    # numpy default_rng(3 + 1000 k)); audio = uniform [-1,1) generated in HBM, one torch generator per
    # (stream, channel) so a stream's channel is the same samples whichever rank owns it.
    n_idx = np.arange(taps, dtype=np.float64)
    hs = []
    for k in range(first_stream, first_stream + n_eng):
        rng = np.random.default_rng(3 + 1000 * k)
        h_all = []
        for _ in range(C_all):
            v = rng.uniform(-1.0, 1.0, taps) * np.exp(-6.0 * n_idx / taps)
            h_all.append((v / np.abs(v).sum()).astype(rdt))
        hs.append(h_all[ch_lo:ch_hi])

    eng = d_in = d_out = None
    if n_eng and C:
        d_in = torch.empty((n_eng, nb * L, C), dtype=tdt, device=dev)
        for k in range(n_eng):
            for c in range(C):
                g = torch.Generator(device=dev)
                g.manual_seed(7 + 1000 * (first_stream + k) + (ch_lo + c))
                d_in[k, :, c].uniform_(-1.0, 1.0, generator=g)
        d_out = torch.empty_like(d_in)
        eng = bfir.Brutefir(L, B, s, C, fmt, fmt, device=local, n_engines=n_eng)
        eng.set_chunk(chunk)
        for k in range(n_eng):
            assert eng.set_coeff(hs[k], engine_index=k) == 0
    eng_stride = nb * L * C * fb
    stream = torch.cuda.current_stream()

    def step():
        if eng is not None:
            eng.run_device(d_in.data_ptr(), d_out.data_ptr(), nb, in_stride_bytes=eng_stride,
                           out_stride_bytes=eng_stride, stream=stream.cuda_stream)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    if eng is not None:
        assert eng.sync() == 0
        eng.set_profiling(not args.no_kernel_events)   # also zeroes the counters
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    prof = {}
    if eng is not None:
        assert eng.sync() == 0
        prof = eng.profile()
        eng.set_profiling(False)

    # parity of the TIMED output: sampled blocks of d_out as the last timed step left it.  Every step
    # reads the same resident input and continues the previous step's history, so global block g of
    # the stream is block g mod nb of the buffer; a block needs the B+1 input blocks ending at it.
    parity = None
    parity_note = None
    if eng is not None and rank == 0 and not args.no_cpu_baseline and args.steps + args.warmup > 0:
        from oracle import oracle as O   # checker + CPU baseline only
        O.build()
        total_steps = args.warmup + args.steps
        g_base = (total_steps - 1) * nb
        R = mac_range(chunk, N, n_eng * C, B) if s == 4 else 0
        cand = {0, 1, nb // 2, nb - 2, nb - 1, chunk - 1, chunk % nb}
        if R and R < nb:
            cand |= {R - 1, R, nb - R - 1, nb - R}
        cand = sorted(t for t in cand if 0 <= t < nb)

        def in_block(k, g):
            if g < 0:
                return np.zeros((L, C), np.float32 if fb == 4 else np.float64)
            t = g % nb
            return d_in[k, t * L:(t + 1) * L].cpu().numpy()
        worst = 0.0
        for k in sorted({0, n_eng - 1}):
            ref = O.sampled_reference(hs[k], lambda g: in_block(k, g), [g_base + t for t in cand], L, B, s, C, fmt, fmt)
            for t in cand:
                y = d_out[k, t * L:(t + 1) * L].cpu().numpy().astype(np.float64)
                r = ref[g_base + t].astype(np.float64)
                worst = max(worst, float(np.abs(y - r).max() / np.abs(r).max()))
        parity = worst
        parity_note = ("%d sampled blocks of the timed output buffer (last step; first/last block, launch and "
                       "MAC-range boundaries, middle) of %d engine(s), each vs the oracle fed the B+1 input "
                       "blocks ending at it" % (len(cand), len({0, n_eng - 1})))
        assert parity <= (1e-5 if min(s, fb) == 4 else 1e-12), parity

    # untimed extra pass for the roofline object: the same engine configuration on a serial schedule
    exclusive = None
    if eng is not None and rank == 0 and not args.no_kernel_events and not args.no_exclusive_pass:
        old_pipe = os.environ.get("BFIR_PIPE")
        os.environ["BFIR_PIPE"] = "1"
        try:
            ser = bfir.Brutefir(L, B, s, C, fmt, fmt, device=local, n_engines=n_eng)
        finally:
            if old_pipe is None:
                del os.environ["BFIR_PIPE"]
            else:
                os.environ["BFIR_PIPE"] = old_pipe
        ser.set_chunk(chunk)
        for k in range(n_eng):
            assert ser.set_coeff(hs[k], engine_index=k) == 0
        nbx = min(nb, 4 * chunk)
        ser.run_device(d_in.data_ptr(), d_out.data_ptr(), nbx, in_stride_bytes=eng_stride,
                       out_stride_bytes=eng_stride, stream=stream.cuda_stream)
        assert ser.sync() == 0
        ser.set_profiling(True)
        for _ in range(3):
            ser.run_device(d_in.data_ptr(), d_out.data_ptr(), nbx, in_stride_bytes=eng_stride,
                           out_stride_bytes=eng_stride, stream=stream.cuda_stream)
        assert ser.sync() == 0
        exclusive = ser.profile()
        ser.close()

    # the job's time is the slowest rank's; its work is the sum of every rank's units
    elapsed = sharding.max_over_ranks(elapsed, red_dev)
    total_samples = sharding.sum_over_ranks(n_eng * nb * L * C * args.steps, red_dev)
    value = total_samples / elapsed / 1e6

    result = None
    if rank == 0:
        fused_io = bool(prof) and prof.get("k_stage_in", (0, 0))[1] == 0
        alg = algorithmic_bytes_per_block(C, B, N, L, s, fused_io)
        roofline = None
        if not args.no_kernel_events and any(v[1] for v in prof.values()):
            traffic, traffic_file = measured_traffic_table(args.workload, chunk)
            dom = max(prof, key=lambda k: prof[k][0])
            ms, launches = prof[dom]
            blocks_per_launch = n_eng * args.steps * nb / launches   # engine-blocks in one launch
            roofline = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s"}
            dom_achieved = alg[dom] * blocks_per_launch / (ms / launches * 1e-3) / 1e9
            dominant = {"kernel": dom, "avg_launch_ms": round(ms / launches, 5),
                        "algorithmic_bytes_per_launch": int(alg[dom] * blocks_per_launch),
                        "achieved": round(dom_achieved, 1), "frac": round(dom_achieved / HBM_PEAK_GBS, 4),
                        "traffic": traffic.get(dom) if traffic else None,
                        "what": "the kernel with the most launch time: ITS share of SURVEY 8(d)'s algorithmic bytes over "
                                "ITS mean launch duration (HIP events on its stream, inside the timed region, while it "
                                "shares the GPU with the other two kernels)"}
            per = {}
            for kname, (kms, kl) in prof.items():
                if not kl:
                    continue
                bpl = n_eng * args.steps * nb / kl
                ach = alg[kname] * bpl / (kms / kl * 1e-3) / 1e9
                per[kname] = {"avg_launch_ms": round(kms / kl, 5), "algorithmic_bytes_per_launch": int(alg[kname] * bpl),
                              "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4),
                              "traffic": traffic.get(kname) if traffic else None}
            # whole path: one launch set (fwd + mac + inv of one chunk) per elapsed/launches of wall time
            ms_set = elapsed * 1e3 / max(launches, 1)
            alg_set = sum(alg.values()) * blocks_per_launch
            compulsory = (2 * C * s * L) * blocks_per_launch          # input once in, output once out
            three_kernel_floor = (2 * C * s * L + 4 * C * s * N) * blocks_per_launch   # + X and Y each written and read once
            tset = sum(traffic.get(k, 0) for k in per) if traffic else None
            fl = flops_per_block(C, B, N)
            flops_set = (fl["mac"] + fl["fft"]) * blocks_per_launch
            valu_peak = FP32_VECTOR_PEAK_TFLOPS * (1.0 if s == 4 else 0.5)
            moved = tset if tset else three_kernel_floor
            rate = moved / (ms_set * 1e-3) / 1e9
            peaks = measured_stream_peaks()
            roofline.update({
                "kernel": "fwd | mac | inv pipeline of one launch set (three kernels sharing the GPU; most launch time: %s)" % dom,
                "achieved": round(rate, 1),
                "frac": round(rate / HBM_PEAK_GBS, 4),
                "traffic": int(tset) if tset else None,
                "frac_is": ("bytes at the L2's memory side per launch set (rocprofv3 PMC, %s) / wall time per launch set / 8 TB/s"
                            % traffic_file) if tset else
                           ("NO counter file matches this csrc/: bytes a three-kernel pipeline cannot avoid (input, output, "
                            "X and Y spectra each written and read once) / wall time per launch set / 8 TB/s -- a lower "
                            "bound on the traffic, see traffic_source"),
                "traffic_source": traffic_file or "none matching this csrc/ (stale files are refused)",
                "useful_frac": round(compulsory / (ms_set * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "useful_frac_is": "compulsory bytes (input once in, output once out) / wall / 8 TB/s",
                "literal_8d_x_peak": round(alg_set / (ms_set * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "literal_8d_is": "SURVEY 8(d)'s streaming bytes (every block re-reads all partition and delay-line spectra) "
                                 "/ wall / 8 TB/s: above 1 because those bytes are not moved -- the MAC keeps a bin's "
                                 "partitions in registers across the blocks in flight (8(d) caveat 2)",
                "valu_frac": round(flops_set / (ms_set * 1e-3) / 1e12 / valu_peak, 4),
                "valu_frac_is": "(MAC + FFT flops per launch set) / wall / %.1f TFLOP/s vector peak" % valu_peak,
                "frac_of_measured_mixed_stream": round(rate / peaks["copy_GBs"], 4) if peaks and peaks.get("copy_GBs") else None,
                "ms_per_launch_set": round(ms_set, 5),
                "dominant_kernel": dominant,
                "kernels": per,
                "peak_measured": peaks,
            })
            pipe = {"ms_per_launch_set": round(ms_set, 5),
                    "algorithmic_bytes_per_launch_set": int(alg_set),
                    "compulsory_bytes_per_launch_set": int(compulsory),
                    "three_kernel_floor_bytes_per_launch_set": int(three_kernel_floor),
                    "flops_per_launch_set": int(flops_set),
                    "TFLOPs": round(flops_set / (ms_set * 1e-3) / 1e12, 2)}
            if tset:
                pipe.update({"traffic_bytes_per_launch_set": int(tset),
                             "traffic_GBs": round(tset / (ms_set * 1e-3) / 1e9, 1),
                             "traffic_over_compulsory": round(tset / compulsory, 3)})
            roofline["pipeline"] = pipe
            # The timed region runs fwd(k+1), mac(k) and inv(k-1) concurrently on three streams, so
            # the launch durations above are those of kernels SHARING the GPU.  One extra untimed pass
            # on a serial schedule (BFIR_PIPE=1) gives each kernel's duration with the GPU to itself.
            if exclusive is not None:
                ex = {k: v[0] / max(v[1], 1) for k, v in exclusive.items() if v[1]}
                roofline["exclusive_launch_ms"] = {k: round(v, 5) for k, v in ex.items()}
                for kname, v in per.items():
                    if kname in ex:
                        v["exclusive_launch_ms"] = round(ex[kname], 5)
                        v["frac_exclusive"] = round(alg[kname] * (n_eng * args.steps * nb / prof[kname][1])
                                                    / (ex[kname] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                        if v["traffic"]:
                            v["traffic_rate_exclusive_GBs"] = round(v["traffic"] / (ex[kname] * 1e-3) / 1e9, 1)
        cpu = None
        extras = {}
        if not args.no_cpu_baseline and not args.no_cpu_timing and world == 1:
            from oracle import oracle as O
            nb_cpu = min(nb, 4096)
            x_cpu = d_in[0, :nb_cpu * L].cpu().numpy()
            cpu = cpu_baseline(O, C, taps, L, B, s, hs[0], x_cpu)
        if not args.no_extras and world == 1 and eng is not None:
            extras = host_path_figures(bfir, torch, L, B, s, C, hs[0], d_in, local)
        result = {
            "metric": "Msamples/s (output channel-samples), 8ch 131072-tap FIR @4096-sample partitions"
                      if args.workload.startswith("cfg3") else "Msamples/s (output channel-samples)",
            "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32" if s == 4 else "f64", "data": "synthetic",
            "config": {"workload": args.workload, "channels": C_all, "taps": taps, "partition": L,
                       "fft_size": N, "partitions": B, "blocks_per_step": nb,
                       "blocks_per_launch": chunk, "engines_rank0": n_eng, "channels_rank0": C,
                       "streams_total": args.streams if args.streams > 0 else (1 if scaling == "strong" else world),
                       "io": "interleaved frames resident in HBM (generated there)",
                       "parallelism": mode + ", no collective"},
            "timed_region_s": round(elapsed, 4),
            "per_gpu_value": round(value / world, 1),
            "pct_of_hbm_roofline_algorithmic": round(
                100.0 * (value / world * 1e6) * (sum(algorithmic_bytes_per_block(C_all, B, N, L, s).values()) / (L * C_all))
                / (HBM_PEAK_GBS * 1e9), 2),
            "roofline": roofline, "cpu_baseline": cpu, "parity_rel_err_vs_oracle": parity,
            "parity_checked": parity_note,
        }
        result.update(extras)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


def host_path_figures(bfir, torch, L, B, s, C, h, d_in, local):
    """end_to_end: the host-pointer entry (bfir_engine_run: pageable host buffers -> pinned staging ->
    H2D -> kernels -> D2H -> host), PCIe inclusive, on a bounded sample.  latency_one_block_us: one
    run() per L-frame block, the plug-in's call pattern (foo_dsp_bfir/foo_dsp_bfir.cpp:311-349)."""
    out = {}
    try:
        nbh = min(d_in.shape[1] // L, 2048)
        x = d_in[0, :nbh * L].cpu().numpy()
        e = bfir.Brutefir(L, B, s, C, device=local)
        assert e.set_coeff(h) == 0
        y = np.empty_like(x)
        e.run(x, y)                                        # allocates the staging buffers
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            rc, _ = e.run(x, y)
            assert rc == 0
        dt = (time.perf_counter() - t0) / reps
        out["end_to_end"] = {"value": round(nbh * L * C / dt / 1e6, 1), "unit": "Msamples/s",
                             "what": "bfir_engine_run on host buffers, PCIe and host staging copies inclusive",
                             "sample": "%d blocks per call, mean of %d calls" % (nbh, reps)}
        # the same call on page-locked caller buffers (bfir_pinned_malloc): no staging memcpy, the copy engines take them directly
        xp = bfir.pinned_frames(e.in_format, x.shape); xp[...] = x
        yp = bfir.pinned_frames(e.out_format, x.shape)
        e.run(xp, yp)
        t0 = time.perf_counter()
        for _ in range(reps):
            rc, _ = e.run(xp, yp)
            assert rc == 0
        dtp = (time.perf_counter() - t0) / reps
        out["end_to_end"]["pinned_caller_buffers"] = {"value": round(nbh * L * C / dtp / 1e6, 1), "unit": "Msamples/s",
                                                      "same_bits_as_pageable": bool(np.array_equal(yp, y))}
        del xp, yp
        lat = []
        for t in range(min(nbh, 300)):
            blk = x[t * L:(t + 1) * L]
            t0 = time.perf_counter()
            rc, _ = e.run(blk, y[:L])
            lat.append(time.perf_counter() - t0)
        lat = np.array(lat[20:]) * 1e6
        out["latency_one_block_us"] = {"median": round(float(np.median(lat)), 1), "p99": round(float(np.percentile(lat, 99)), 1),
                                       "what": "one bfir_engine_run of one %d-frame block of %d channels, host to host" % (L, C),
                                       "calls": int(lat.size)}
        e.close()
    except Exception as exc:   # never cost the bench line
        out["host_path_error"] = repr(exc)
    return out


if __name__ == "__main__":
    main()
