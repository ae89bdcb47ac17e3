import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import foo_dsp_bfir_amd as bfir
L, B, C = 16384, 1, 1
N = 2 * L
for j in (0, 1, 5):
    h = np.zeros((C, L), dtype=np.float32); h[0, j] = 1.0
    eng = bfir.Brutefir(L, B, 4, C); eng.set_coeff(list(h))
    hb = eng.coeff_block(0, 0).astype(np.float64) * N
    k = np.arange(N // 2)
    want = np.exp(-2j * np.pi * k * (L + j) / N)
    got = hb.reshape(-1, 2, 4)[:, 0, :].reshape(-1) + 1j * hb.reshape(-1, 2, 4)[:, 1, :].reshape(-1)
    err = np.abs(got - want); err[0] = 0
    bad = np.nonzero(err > 1e-3)[0]
    print("delta at", j, "bad bins:", len(bad), "first", bad[:24], "last", bad[-8:] if len(bad) else None)
    if len(bad):
        b0 = bad[0]
        print("  got", got[b0:b0 + 4], "want", want[b0:b0 + 4])
        # is got[k] == want[k'] for some k'?
        for b in bad[:6]:
            m = np.argmin(np.abs(want - got[b])); print("   bin", b, "holds value of bin", m, "err", abs(want[m] - got[b]))
    eng.close()
