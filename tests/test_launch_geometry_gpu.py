"""Parity at the launch geometry the benchmark times ("What's weak" 1).

bench.py runs the headline shape with 4096 blocks per kernel launch, 16384 per call: the
time-streaming MAC then has ngrp = 32 groups per wave, nR = 4 ranges per bin column (its
MODE 1 full-group loop does > 90 % of the work), the delay-line ring (2*4096 + B slots) wraps
inside a launch, and calls continue each other's history.  The tests here drive exactly those
launches through bfir_engine_run_device and check

  * sampled output blocks (first block of every call, both sides of every range boundary,
    around the ring wrap, the middle, the last) against the oracle.  A FIR has B blocks of
    memory, so the oracle only needs the B+1 input blocks that end at the sampled one
    (oracle.sampled_reference; brutefir/brutefir.cpp:288-299 fixes the partition order);
  * EVERY output block, bit for bit, against the same engine cut into 64-block launches --
    the geometry the rest of the suite compares with the oracle in full.

The other launch-shape-dependent MAC kernels get the same treatment: the LDS-shared kernel on the
pair layout (B > 32), the batched streaming kernel that continues sums left in Y (ACC, B > 32,
ngrp > 1 -- a tuning path behind BFIR_MAC_BATCHED), the general (planar staging) path, the fp64
LDS kernel and a batch of engines sharing launches."""
import os

import numpy as np
import pytest

from conftest import TOL

pytestmark = pytest.mark.gpu


class _Env:
    """Set environment switches the library reads per engine / per launch, restore afterwards."""
    def __init__(self, **kv):
        self.kv = {k: v for k, v in kv.items() if v is not None}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _synth(orc, s, C, taps, n_eng, seed):
    rng = np.random.default_rng(seed)
    return [orc.synth_ir(rng, C, taps, orc.real_dtype(s)) for _ in range(n_eng)]


def _run_calls(bfir, torch, L, B, s, C, hs, d_in, calls, chunk, env):
    """One engine (batch), `calls` consecutive run_device calls over the head of the same resident
    input (the stream of call k+1 continues call k's history).  Returns the outputs as device tensors."""
    n_eng = d_in.shape[0]
    nb_in = d_in.shape[1] // L
    fb = d_in.element_size()                               # bytes per frame sample: 4 = FLOAT_LE frames, 8 = FLOAT64_LE
    fmt = 8 if fb == 4 else 10
    env = {k: v for k, v in env.items() if not k.startswith("_")}
    with _Env(**env):
        eng = bfir.Brutefir(L, B, s, C, fmt, fmt, n_engines=n_eng)
        eng.set_chunk(chunk)
        for k in range(n_eng):
            assert eng.set_coeff(hs[k], engine_index=k) == 0
        stride = nb_in * L * C * fb
        outs = []
        for n in calls:
            d_out = torch.empty((n_eng, n * L, C), dtype=d_in.dtype, device=d_in.device)
            eng.run_device(d_in.data_ptr(), d_out.data_ptr(), n, in_stride_bytes=stride,
                           out_stride_bytes=n * L * C * fb, stream=torch.cuda.current_stream().cuda_stream)
            assert eng.sync() == 0
            outs.append(d_out)
        for c in range(n_eng * C):
            assert eng.overflow(c).n_overflows == 0
        eng.close()
    return outs


def _mac_range(tc, N, n_ch, B, env):
    """Blocks per k_mac_stream range for a launch of tc blocks, as launch_mac_stream picks it
    (csrc/kernels.hip); 0 when another MAC kernel runs (no ranges)."""
    if B > 32 and "BFIR_MAC_BATCHED" not in env and tc >= 32:
        return 0
    PB = 4 if B <= 4 else 8 if B <= 8 else 16 if B <= 16 else 32
    if "BFIR_MAC_RANGE" in env:
        return max(1, int(env["BFIR_MAC_RANGE"]) // PB) * PB
    cols = (N // 2 // 256) * n_ch
    want = max(1, 512 // cols)
    return max(1, -(-tc // (want * PB))) * PB


def _sample_points(calls, chunk, ring, rng_of):
    """(call index, local block) pairs worth checking: call starts (history hand-over), launch
    and MAC-range boundaries, the ring wrap, middles and ends."""
    pts, g0 = set(), 0
    for ci, n in enumerate(calls):
        want = {0, 1, n // 2, n - 2, n - 1}
        for c0 in range(0, n, chunk):                      # launches of this call
            tc = min(chunk, n - c0)
            want |= {c0, c0 + tc - 1}
            R = rng_of(tc)
            if R:
                for r in range(R, tc, R):                  # range boundaries inside the launch
                    want |= {c0 + r - 1, c0 + r, c0 + r + 1}
        wrap = ring - (g0 % ring)                          # local block that lands in ring slot 0
        if 0 < wrap < n:
            want |= {wrap - 1, wrap, wrap + 1}
        pts |= {(ci, t) for t in want if 0 <= t < n}
        g0 += n
    return sorted(pts)


def _check(orc, torch, L, B, s, C, hs, x_host, calls, outs, chunk, rng_of, max_points=28, fmt=None):
    """Sampled oracle parity.  x_host: [n_eng, nb_in*L, C]; call k reads blocks 0 .. calls[k]-1 of it."""
    ring = 2 * chunk + B
    pts = _sample_points(calls, chunk, ring, rng_of)
    if len(pts) > max_points:                              # keep the run in seconds: thin the interior evenly
        keep = {p for p in pts if p[1] in (0, calls[p[0]] - 1)}
        rest = [p for p in pts if p not in keep]
        step = max(1, len(rest) // (max_points - len(keep)))
        pts = sorted(keep | set(rest[::step]))
    starts = np.cumsum([0] + list(calls))

    def block_of(e, g):                                    # global block g of engine e's virtual input stream
        ci = int(np.searchsorted(starts, g, side="right") - 1)
        t = g - starts[ci]
        return x_host[e, t * L:(t + 1) * L]

    worst = 0.0
    for e in range(x_host.shape[0]):
        glob = [int(starts[ci] + t) for ci, t in pts]
        ref = orc.sampled_reference(hs[e], lambda g: block_of(e, g), glob, L, B, s, C, fmt, fmt)
        for (ci, t), g in zip(pts, glob):
            y = outs[ci][e, t * L:(t + 1) * L].cpu().numpy().astype(np.float64)
            r = ref[g].astype(np.float64)
            err = float(np.abs(y - r).max() / np.abs(r).max())
            assert err <= (TOL[4] if fmt == 8 else TOL[s]), (e, ci, t, err)     # float32 output frames round to 6e-8
            worst = max(worst, err)
    return worst, len(pts)


# name, s, L, B, C, n_eng, blocks resident, calls, chunk, env
GEOMETRIES = [
    # the timed shape: pair path, k_mac_stream<32>, ngrp = 32, nR = 4; 512-block launch: ngrp = 4, nR = 4;
    # ring 8224 wraps inside the third call; a one-block call continues the history
    ("cfg3_pair_4096", 4, 4096, 32, 8, 1, 4096, [4096, 512, 4096, 1], 4096, {}),
    # the same shape through the general path (stage_in / k_fwd / k_mac_stream / k_inv / stage_out)
    ("cfg3_planar_1024", 4, 4096, 32, 8, 1, 1024, [1024, 1024, 520], 1024, {"BFIR_PAIR": "0"}),
    # explicit short ranges: many ranges per column, last range with fewer groups than the others
    ("cfg3_pair_range96", 4, 4096, 32, 8, 1, 1000, [1000, 1000], 1000, {"BFIR_MAC_RANGE": "96"}),
    # the plug-in's partition size with B = 64: k_mac_lds on the pair layout, 32-block time tiles
    ("plugin_B64_lds", 4, 1024, 64, 8, 1, 2048, [2048, 2048, 300], 2048, {}),
    # B = 64 through the batched streaming kernel: second batch continues the sums in Y (ACC), ngrp = 4
    ("plugin_B64_acc", 4, 1024, 64, 8, 1, 2048, [2048, 2048, 300], 2048, {"BFIR_MAC_BATCHED": "1"}),
    # configs[4]: fp64, B = 64, LDS-shared fp64 MAC with 16-block time tiles
    ("cfg5_fp64_1024", 8, 4096, 64, 2, 1, 1024, [1024, 1024, 100], 1024, {}),
    # configs[3] in small: 8 stereo engines sharing launches, PB = 16
    ("cfg4_batch8", 4, 4096, 16, 2, 8, 1024, [1024, 1024, 64], 1024, {}),
    # the plug-in's shape on the float fast path, eight launches per call (persistent pair kernels at N = 2048;
    # a sporadic fault in their epilogue was found at exactly this shape in round 2)
    ("plugin_stereo_8_launches", 4, 1024, 64, 2, 1, 4096, [4096, 100], 512, {}),
    # configs[1]: largest pair-path partition
    ("cfg2_L8192", 4, 8192, 8, 2, 1, 1024, [1024, 1024, 3], 1024, {}),
    # the plug-in as shipped: REALSIZE 8 arithmetic, FILTER_LEN 1024, float32 stereo frames, 64 partitions -- both
    # channels of a block in one workgroup (direct mode), fp64 LDS MAC with two partitions per barrier, 4096-block
    # launches, three engines sharing them, then a one-block call (k_mac_small)
    ("plugin_fp64_f32frames_4096", 8, 1024, 64, 2, 3, 4096, [4096, 1000, 1], 4096, {"_FRAMES": "f32"}),
]


@pytest.mark.parametrize("name,s,L,B,C,n_eng,nb,calls,chunk,env", GEOMETRIES, ids=[g[0] for g in GEOMETRIES])
def test_large_launches_match_oracle_and_small_launches(orc, bfir, name, s, L, B, C, n_eng, nb, calls, chunk, env):
    import torch
    taps = B * L - 37                                      # ragged last partition
    hs = _synth(orc, s, C, taps, n_eng, seed=len(name) + L)
    rng = np.random.default_rng(B + C)
    rdt = orc.real_dtype(s)
    f32 = s == 4 or env.get("_FRAMES") == "f32"
    if f32:
        x_host = rng.random((n_eng, nb * L, C), dtype=np.float32)
        x_host *= 2.0; x_host -= 1.0
    else:
        x_host = rng.uniform(-1.0, 1.0, (n_eng, nb * L, C)).astype(rdt)
    d_in = torch.from_numpy(x_host).cuda()

    big = _run_calls(bfir, torch, L, B, s, C, hs, d_in, calls, chunk, env)
    ilv = s == 4                                           # fp32 engines with N >= 512 use the streaming MAC family
    rng_of = (lambda tc: _mac_range(tc, 2 * L, n_eng * C, B, env)) if ilv else (lambda tc: 0)
    worst, n_pts = _check(orc, torch, L, B, s, C, hs, x_host, calls, big, chunk, rng_of,
                          fmt=8 if (s == 8 and f32) else None)

    # every block, bit for bit, against 64-block launches (ngrp = 1: the geometry checked in full elsewhere)
    env_small = {k: v for k, v in env.items() if k != "BFIR_MAC_RANGE"}
    small = _run_calls(bfir, torch, L, B, s, C, hs, d_in, calls, 64, env_small)
    for a, b in zip(big, small):
        assert torch.equal(a, b)
    print("%s: %d sampled blocks, worst rel err %.3g; all %d blocks bit-identical to 64-block launches"
          % (name, n_pts, worst, sum(calls)))
