import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import foo_dsp_bfir_amd as bfir
from oracle import oracle as orc
for (s, L, B, C, taps, nb, chunk) in [(4, 16384, 2, 1, 20000, 3, 2), (4, 16384, 2, 1, 20000, 3, 4), (4, 8192, 2, 1, 9000, 3, 2), (4, 16384, 1, 1, 9000, 3, 2), (4, 16384, 5, 2, 70000, 6, 3)]:
    rng = np.random.default_rng(L + B + C)
    dt = orc.real_dtype(s)
    h, x = orc.synth_ir(rng, C, taps, dt), orc.synth_audio(rng, nb * L, C, dt)
    ref = orc.Engine(L, B, s, C); ref.set_coeff(h); _, y_ref = ref.run(x)
    eng = bfir.Brutefir(L, B, s, C); eng.set_chunk(chunk); eng.set_coeff(h)
    rc, y = eng.run(x)
    pk = np.abs(y_ref).max()
    errs = [float(np.abs(y[b * L:(b + 1) * L] - y_ref[b * L:(b + 1) * L]).max() / pk) for b in range(nb)]
    # spectra check of H
    hb = eng.coeff_block(0, 0); rb = ref.coeff_block(0, 0) if hasattr(ref, "coeff_block") else None
    herr = None if rb is None else float(np.abs(hb - rb).max() / np.abs(rb).max())
    print((s, L, B, C, taps, nb, chunk), "per-block err", ["%.1e" % e for e in errs], "H err", herr)
    eng.close()
