"""PCIe-inclusive numbers for DESIGN.md: bfir_engine_run on host buffers (pinned double-buffered staging)
and the plug-in's real-time pattern (one run() per block, foo_dsp_bfir.cpp:311-349)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import foo_dsp_bfir_amd as bfir

rng = np.random.default_rng(0)

def engine(L, B, C, s=4, in_fmt=None, out_fmt=None):
    e = bfir.Brutefir(L, B, s, C, in_fmt, out_fmt)
    dt = np.float32 if s == 4 else np.float64
    e.set_coeff([(rng.standard_normal(L * B) * 0.01).astype(dt) for _ in range(C)])
    return e

# batched host call, headline shape
L, B, C, nb = 4096, 32, 8, 4096
e = engine(L, B, C); e.set_chunk(512)
x = rng.uniform(-1, 1, (nb * L, C)).astype(np.float32)
e.run(x)                                   # warm-up (allocations, pinned buffers)
t0 = time.perf_counter(); rc, y = e.run(x); dt = time.perf_counter() - t0
print("host-buffer run(): %d blocks of the headline shape in %.1f ms = %.2f Gsamples/s (PCIe + host memcpy inclusive)"
      % (nb, dt * 1e3, nb * L * C / dt / 1e9))
xp = bfir.pinned_frames(bfir.SAMPLE_FORMAT_FLOAT_LE, x.shape); xp[...] = x
yp = bfir.pinned_frames(bfir.SAMPLE_FORMAT_FLOAT_LE, x.shape)
e.run(xp, yp)
t0 = time.perf_counter(); rc, _ = e.run(xp, yp); dtp = time.perf_counter() - t0
print("host-buffer run() on page-locked caller buffers (bfir_pinned_malloc, no staging memcpy): %.1f ms = %.2f Gsamples/s, same bits: %s"
      % (dtp * 1e3, nb * L * C / dtp / 1e9, bool(np.array_equal(yp, y))))
del xp, yp
e.close()

# one block per call: the plug-in's own shape first (REALSIZE 8, FILTER_LEN 1024, float32 frames, stereo), its fp32 sibling, the headline
for (L, B, C, s) in [(1024, 64, 2, 8), (1024, 64, 2, 4), (4096, 32, 8, 4)]:
    for small in (True, False):
        if small: os.environ.pop("BFIR_NO_SMALL_RUN", None)
        else: os.environ["BFIR_NO_SMALL_RUN"] = "1"
        e = engine(L, B, C, s, 8, 8)                       # FLOAT_LE frames in and out (foo_dsp_bfir.cpp:283-284)
        blk = rng.uniform(-1, 1, (L, C)).astype(np.float32)
        out = np.empty_like(blk)
        for _ in range(50): e.run(blk, out)
        ts = []
        for _ in range(400):
            t0 = time.perf_counter(); e.run(blk, out); ts.append(time.perf_counter() - t0)
        ts = np.array(ts) * 1e6
        print("one run() per block, realsize %d L=%d B=%d C=%d, %s: median %.0f us, p95 %.0f us per call (block = %.1f ms of 44.1 kHz audio)"
              % (s, L, B, C, "latency path" if small else "pipelined path (round 1)", np.median(ts), np.percentile(ts, 95), L / 44.1))
        e.close()
os.environ.pop("BFIR_NO_SMALL_RUN", None)
