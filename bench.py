#!/usr/bin/env python3
"""bench.py -- throughput of the partitioned-FIR hot path on MI355X.

Workload (BASELINE.json configs[2], the headline): 8 channels, 131072-tap IR,
4096-sample partitions (N = 8192, B = 32), fp32, synthetic uniform noise.
One STEP = one bfir_engine_run_device call over `--blocks` consecutive blocks
of input that is already resident in HBM.  Metric: output channel-samples/s.

    python bench.py [--gpus N] [--steps K] [--warmup W]
N > 1: launched by torch.distributed.run, one rank per GPU; every rank runs its
own independent 8-channel engine (weak scaling, no data-path collective).

The JSON line also carries
  roofline     algorithmic HBM bytes of the dominant kernel / its mean launch
               time (HIP events on the launch stream) against 8 TB/s; the timed
               region overlaps three kernels, so the same figure from an untimed
               serial-schedule pass is given beside it (*_exclusive)
  cpu_baseline the CPU oracle (a port of the reference algorithm) timed on a
               bounded sample of the same workload on this host, 1 thread.
"""
import argparse
import json
import os
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
    # not a BASELINE config: the plug-in's own partition size (FILTER_LEN 1024), B = 128 partitions
    "plugin_8ch_131072tap_L1024_fp32": (8, 131072, 1024, 4),
    "plugin_8ch_65536tap_L1024_fp32": (8, 65536, 1024, 4),
    "plugin_8ch_98304tap_L1024_fp32": (8, 98304, 1024, 4),
}


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


def measured_traffic(kernel, workload, chunk):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/traffic_*.json,
    written by scripts/summarize_pmc.py from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this
    same command), or None when no pass matches this workload and blocks-per-launch."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_*.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        m = d.get("meta", {})
        if m.get("workload") == workload and int(m.get("chunk", -1)) == int(chunk) and kernel in d["per_launch"]:
            return d["per_launch"][kernel]["total_bytes"]
    return None


def measured_stream_peaks():
    """Best read / write / copy GB/s of scripts/ubench/hbm_stream.hip on this GPU model
    (profiles/r01_hbm_stream.txt; SURVEY 8(d) asks for the measured stream peak next to the 8 TB/s spec)."""
    import re
    path = os.path.join(ROOT, "profiles", "r01_hbm_stream.txt")
    best = {}
    try:
        for line in open(path):
            for key in ("read", "write", "copy"):
                m = re.search(key + r" (\d+) GB/s", line)
                if m:
                    best[key + "_GBs"] = max(best.get(key + "_GBs", 0), int(m.group(1)))
    except OSError:
        return None
    return best or None


def cpu_baseline(O, C, taps, L, B, s, h, x, budget_s=12.0):
    """Time the oracle (1 thread) on a bounded sample: B warm-up blocks, then blocks until ~budget."""
    eng = O.Engine(L, B, s, C)
    assert eng.set_coeff(h) == 0
    nb_avail = min(x.shape[0] // L, 8192)      # one timed pass stays near the budget
    warm = min(B, nb_avail)
    eng.run(x[:warm * L])
    # timed: whole passes over the resident input (the stream simply continues), ~budget_s
    n, dt = 0, 0.0
    while dt < budget_s:
        t0 = time.perf_counter()
        rc, _ = eng.run(x[:nb_avail * L])
        dt += time.perf_counter() - t0
        n += nb_avail
        assert rc == 0
    out = {"value": n * L * C / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
           "sample": "%d blocks (%d channel-samples) after %d warm-up blocks, oracle/bfir_oracle.c, "
                     "1 thread, %.1f s" % (n, n * L * C, warm, dt)}
    # also all host cores, channel-parallel (one single-channel engine per thread; the
    # reference itself is single-threaded per instance).  ctypes releases the GIL inside the oracle.
    try:
        from concurrent.futures import ThreadPoolExecutor
        threads = max(1, min(C, len(os.sched_getaffinity(0))))
        engs, cols = [], []
        for c in range(C):
            e1 = O.Engine(L, B, s, 1)
            assert e1.set_coeff([h[c]]) == 0
            engs.append(e1)
            cols.append(np.ascontiguousarray(x[:nb_avail * L, c:c + 1]))
        nb_par = nb_avail
        def one(i):
            engs[i].run(cols[i][:warm * L])
            return engs[i].run(cols[i][:nb_par * L])[0]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=threads) as pool:
            rcs = list(pool.map(one, range(C)))
        dtp = time.perf_counter() - t0
        assert all(r == 0 for r in rcs)
        model = ""
        try:
            for line in open("/proc/cpuinfo"):
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip(); break
        except OSError:
            pass
        out["all_cores"] = {"value": (nb_par + warm) * L * C / dtp / 1e6, "unit": "Msamples/s", "cores": threads,
                            "cpu_model": model, "host_cores_available": len(os.sched_getaffinity(0)),
                            "sample": "%d blocks per channel, %d single-channel engines on %d threads, %.1f s"
                                      % (nb_par + warm, C, threads, dtp)}
    except Exception as exc:   # the extra row must never cost the bench line
        out["all_cores"] = {"error": repr(exc)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--blocks", type=int, default=16384,
                    help="blocks per step (one run_device call; the chunk pipeline drains at call boundaries)")
    ap.add_argument("--chunk", type=int, default=int(os.environ.get("BFIR_CHUNK", "4096")),
                    help="blocks per kernel launch")
    ap.add_argument("--workload", default="cfg3_8ch_131072tap_L4096_fp32", choices=sorted(WORKLOADS))
    ap.add_argument("--streams", type=int, default=0,
                    help="total independent engines dealt out to the ranks (0 = one per rank); "
                         "configs[3]: --workload cfg4_stereo_65536tap_L4096_fp32 --streams 256 --blocks 64")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exclusive-pass", action="store_true",
                    help="skip the untimed serial-schedule pass (rocprofv3 runs: keeps the kernel trace to the overlapped schedule)")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not bracket kernels with HIP events in the timed region")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for the timing reduction (nccl = RCCL; gloo lets several "
                         "ranks rehearse the N > 1 path on one GPU together with --device)")
    ap.add_argument("--device", type=int, default=-1, help="HIP device ordinal (default: LOCAL_RANK)")
    args = ap.parse_args()

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

    C, taps, L, s = WORKLOADS[args.workload]
    N, B = 2 * L, (taps + L - 1) // L
    rdt = np.float32 if s == 4 else np.float64
    nb = args.blocks
    # Units = independent engines (streams).  Headline: one 8-channel engine per GPU (weak
    # scaling).  --streams S: S engines in total, dealt out to the ranks (strong scaling,
    # BASELINE.json configs[3] with S = 256); no rank ever exchanges audio with another.
    if args.streams > 0:
        lo, hi = sharding.shard_range(args.streams, rank, world)
        n_eng, first_stream, scaling = hi - lo, lo, "strong"
    else:
        n_eng, first_stream, scaling = 1, rank, "weak"

    # synthetic audio / IR of SURVEY.md 8(d); stream k is the same whichever rank owns it
    n = np.arange(taps, dtype=np.float64)
    hs, xs = [], []
    for k in range(first_stream, first_stream + n_eng):
        rng = np.random.default_rng(3 + 1000 * k)
        h = []
        for _ in range(C):
            v = rng.uniform(-1.0, 1.0, taps) * np.exp(-6.0 * n / taps)
            h.append((v / np.abs(v).sum()).astype(rdt))
        hs.append(h)
        if s == 4:   # generated in float32, in place: no float64 temporary of the whole job
            xk = rng.random((nb * L, C), dtype=np.float32)
            xk *= 2.0; xk -= 1.0
        else:
            xk = rng.uniform(-1.0, 1.0, (nb * L, C)).astype(rdt)
        xs.append(xk)
    h, x_host = (hs[0], xs[0]) if n_eng else (None, None)

    eng = d_in = d_out = None
    if n_eng:
        eng = bfir.Brutefir(L, B, s, C, device=local, n_engines=n_eng)
        eng.set_chunk(args.chunk)
        for k in range(n_eng):
            assert eng.set_coeff(hs[k], engine_index=k) == 0
        d_in = torch.from_numpy(xs[0][None] if n_eng == 1 else np.stack(xs)).to(dev)       # [n_eng, nb*L, C]
        d_out = torch.empty_like(d_in)
    eng_stride = nb * L * C * (4 if s == 4 else 8)
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

    # untimed extra pass for the roofline object: the same engine configuration on a serial schedule
    exclusive = None
    if eng is not None and rank == 0 and not args.no_kernel_events and not args.no_exclusive_pass:
        os.environ["BFIR_PIPE"] = "1"
        try:
            ser = bfir.Brutefir(L, B, s, C, device=local, n_engines=n_eng)
        finally:
            del os.environ["BFIR_PIPE"]
        ser.set_chunk(args.chunk)
        for k in range(n_eng):
            assert ser.set_coeff(hs[k], engine_index=k) == 0
        ser.run_device(d_in.data_ptr(), d_out.data_ptr(), nb, in_stride_bytes=eng_stride,
                       out_stride_bytes=eng_stride, stream=stream.cuda_stream)
        assert ser.sync() == 0
        ser.set_profiling(True)
        for _ in range(2):
            ser.run_device(d_in.data_ptr(), d_out.data_ptr(), nb, in_stride_bytes=eng_stride,
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
            dom = max(prof, key=lambda k: prof[k][0])
            ms, launches = prof[dom]
            blocks_per_launch = n_eng * args.steps * nb / launches   # engine-blocks in one launch
            achieved = alg[dom] * blocks_per_launch / (ms / launches * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "traffic": measured_traffic(dom, args.workload, args.chunk),
                        "algorithmic_bytes_per_launch": int(alg[dom] * blocks_per_launch),
                        "avg_launch_ms": round(ms / launches, 5),
                        "kernel_ms_share": {k: round(v[0] / max(sum(p[0] for p in prof.values()), 1e-12), 4)
                                            for k, v in prof.items()}}
            # the same figures for every kernel of the path (the dominant one is repeated above)
            per = {}
            for kname, (kms, kl) in prof.items():
                if not kl:
                    continue
                bpl = n_eng * args.steps * nb / kl
                ach = alg[kname] * bpl / (kms / kl * 1e-3) / 1e9
                per[kname] = {"avg_launch_ms": round(kms / kl, 5), "algorithmic_bytes_per_launch": int(alg[kname] * bpl),
                              "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4),
                              "traffic": measured_traffic(kname, args.workload, args.chunk)}
            roofline["kernels"] = per
            roofline["peak_measured"] = measured_stream_peaks()
            # The timed region runs fwd(k+1), mac(k) and inv(k-1) concurrently on three streams, so
            # the launch durations above are those of kernels SHARING the GPU.  One extra untimed pass
            # on a serial schedule (BFIR_PIPE=1) gives each kernel's duration with the GPU to itself.
            if exclusive is not None:
                ex = {k: v[0] / max(v[1], 1) for k, v in exclusive.items() if v[1]}
                roofline["exclusive_launch_ms"] = {k: round(v, 5) for k, v in ex.items()}
                if dom in ex:
                    a_ex = alg[dom] * blocks_per_launch / (ex[dom] * 1e-3) / 1e9
                    roofline["achieved_exclusive"] = round(a_ex, 1)
                    roofline["frac_exclusive"] = round(a_ex / HBM_PEAK_GBS, 4)
                    t = roofline["traffic"]
                    if t:
                        roofline["traffic_rate_exclusive_GBs"] = round(t / (ex[dom] * 1e-3) / 1e9, 1)
                for kname, v in per.items():
                    if kname in ex:
                        v["exclusive_launch_ms"] = round(ex[kname], 5)
                        v["frac_exclusive"] = round(alg[kname] * (n_eng * args.steps * nb / prof[kname][1])
                                                    / (ex[kname] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                        if v["traffic"]:
                            v["traffic_rate_exclusive_GBs"] = round(v["traffic"] / (ex[kname] * 1e-3) / 1e9, 1)
        cpu = None
        parity = None
        if not args.no_cpu_baseline:
            from oracle import oracle as O   # checker + CPU baseline only
            O.build()
            cpu = cpu_baseline(O, C, taps, L, B, s, h, x_host)
            # parity of this very workload: first blocks of a fresh GPU engine vs the oracle
            k = min(nb, B + 3)
            ref = O.Engine(L, B, s, C); ref.set_coeff(h)
            _, y_ref = ref.run(x_host[:k * L])
            chk = bfir.Brutefir(L, B, s, C, device=local); chk.set_coeff(h)
            rc, y = chk.run(x_host[:k * L])
            parity = float(np.abs(y.astype(np.float64) - y_ref).max() / np.abs(y_ref).max())
            assert rc == 0 and parity <= (1e-5 if s == 4 else 1e-12), parity
        result = {
            "metric": "Msamples/s (output channel-samples), 8ch 131072-tap FIR @4096-sample partitions"
                      if args.workload.startswith("cfg3") else "Msamples/s (output channel-samples)",
            "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32" if s == 4 else "f64", "data": "synthetic",
            "config": {"workload": args.workload, "channels": C, "taps": taps, "partition": L,
                       "fft_size": N, "partitions": B, "blocks_per_step": nb,
                       "blocks_per_launch": args.chunk, "engines_rank0": n_eng,
                       "streams_total": args.streams if args.streams > 0 else world,
                       "io": "interleaved frames resident in HBM",
                       "parallelism": "independent engines per GPU, no collective"},
            "per_gpu_value": round(value / world, 1),
            "pct_of_hbm_roofline_algorithmic": round(
                100.0 * (value / world * 1e6) * (sum(alg.values()) / (L * C)) / (HBM_PEAK_GBS * 1e9), 2),
            "roofline": roofline, "cpu_baseline": cpu, "parity_rel_err_vs_oracle": parity,
        }
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
