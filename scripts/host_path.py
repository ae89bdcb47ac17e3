"""PCIe-inclusive numbers for DESIGN.md: bfir_engine_run on host buffers (pinned double-buffered staging)
and the plug-in's real-time pattern (one run() per block, foo_dsp_bfir.cpp:311-349)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import foo_dsp_bfir_amd as bfir

rng = np.random.default_rng(0)

def engine(L, B, C):
    e = bfir.Brutefir(L, B, 4, C)
    e.set_coeff([(rng.standard_normal(L * B) * 0.01).astype(np.float32) for _ in range(C)])
    return e

# batched host call, headline shape
L, B, C, nb = 4096, 32, 8, 4096
e = engine(L, B, C); e.set_chunk(512)
x = rng.uniform(-1, 1, (nb * L, C)).astype(np.float32)
e.run(x)                                   # warm-up (allocations, pinned buffers)
t0 = time.perf_counter(); rc, y = e.run(x); dt = time.perf_counter() - t0
print("host-buffer run(): %d blocks of the headline shape in %.1f ms = %.2f Gsamples/s (PCIe + host memcpy inclusive)"
      % (nb, dt * 1e3, nb * L * C / dt / 1e9))
e.close()

# one block per call
for (L, B, C) in [(1024, 64, 2), (4096, 32, 8)]:
    e = engine(L, B, C)
    blk = rng.uniform(-1, 1, (L, C)).astype(np.float32)
    for _ in range(20): e.run(blk)
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); e.run(blk); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    print("one run() per block, L=%d B=%d C=%d: median %.0f us, p95 %.0f us per call (block = %.1f ms of 44.1 kHz audio)"
          % (L, B, C, np.median(ts), np.percentile(ts, 95), L / 44.1))
    e.close()
