"""One run() per block for one engine shape (argv: L B C realsize [calls]) -- the subject of rocprofv3 kernel traces of the latency path."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import foo_dsp_bfir_amd as bfir
L, B, C, s = (int(v) for v in sys.argv[1:5])
calls = int(sys.argv[5]) if len(sys.argv) > 5 else 300
rng = np.random.default_rng(0)
e = bfir.Brutefir(L, B, s, C, 8, 8)
e.set_coeff([(rng.standard_normal(L * B) * 0.001).astype(np.float32 if s == 4 else np.float64) for _ in range(C)])
blk = rng.uniform(-1, 1, (L, C)).astype(np.float32)
out = np.empty_like(blk)
for _ in range(50): e.run(blk, out)
ts = []
for _ in range(calls):
    t0 = time.perf_counter(); e.run(blk, out); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e6
print("L=%d B=%d C=%d realsize %d: median %.1f us, p95 %.1f us" % (L, B, C, s, np.median(ts), np.percentile(ts, 95)))
