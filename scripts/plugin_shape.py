"""Throughput of the reference plug-in's exact engine shape (foo_dsp_bfir/common.h:17-18, foo_dsp_bfir.cpp:279-289):
REALSIZE 8 arithmetic, FILTER_LEN 1024, FLOAT_LE (32-bit) frames in and out, stereo, and its fp32 sibling."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import foo_dsp_bfir_amd as bfir

rng = np.random.default_rng(1)
L, C, nb = 1024, 2, 32768
x = torch.from_numpy((rng.random((nb * L, C), dtype=np.float32) * 2 - 1)).cuda()
y = torch.empty_like(x)
# argv[1] = coefficient gain (default 0.01: a fifth of the output samples exceed full scale, so the overflow
# bookkeeping of real2raw is busy; 0.0005: none do)
gain = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
for s in (8, 4):
    for taps in (65536, 131072):
        B = taps // L
        e = bfir.Brutefir(L, B, s, C, 8, 8)              # FLOAT_LE in and out
        dt = np.float64 if s == 8 else np.float32
        e.set_coeff([(rng.standard_normal(taps) * gain).astype(dt) for _ in range(C)])
        for _ in range(2):
            e.run_device(x.data_ptr(), y.data_ptr(), nb); e.sync()
        t0 = time.perf_counter()
        for _ in range(4):
            e.run_device(x.data_ptr(), y.data_ptr(), nb)
        e.sync(); dtm = (time.perf_counter() - t0) / 4
        print("realsize %d, L=1024, B=%3d, stereo float32 frames, gain %g: %.2f Gsamples/s (%d blocks in %.2f ms), %d samples over full scale"
              % (s, B, gain, nb * L * C / dtm / 1e9, nb, dtm * 1e3, sum(e.overflow(c).n_overflows for c in range(C))))
        e.close()
