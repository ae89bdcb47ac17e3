#!/usr/bin/env python3
"""Generate tests/golden/*.npz.

PARITY UNPINNED: the reference (vsu/foo-dsp-bfir) holds no golden vectors and
cannot be built or run in this image (Win32/MSVC sources, FFTW only as Win32
DLLs), so these vectors are NOT outputs of the reference.  Each file holds
seeded inputs, the expected output of an independent long-double direct-form
convolution (oracle.direct_conv, truncated to the filter_blocks*L taps an
engine of that shape uses), and the CPU oracle's output for the same inputs.
They pin the oracle and the HIP engine to the mathematical definition of the
path and guard both against regressions.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402

# name: (realsize, L, B, C, taps, n_blocks)  -- reduced-size versions of BASELINE.json configs
CASES = {
    "cfg1_2ch_8192tap_L4096_f32": (4, 4096, 2, 2, 8192, 5),
    "cfg2r_2ch_2048tap_L256_f32": (4, 256, 8, 2, 2048, 19),
    "cfg3r_8ch_4096tap_L128_f32": (4, 128, 32, 8, 4096, 67),
    "cfg5r_2ch_4096tap_L64_f64": (8, 64, 64, 2, 4096, 131),
    "ragged_3ch_1000tap_L256_f64": (8, 256, 4, 3, 1000, 9),
    "single_partition_2ch_200tap_L256_f32": (4, 256, 1, 2, 200, 4),
}


def main():
    for seed, (name, (s, L, B, C, taps, nb)) in enumerate(sorted(CASES.items())):
        rng = np.random.default_rng(100 + seed)
        dt = O.real_dtype(s)
        h = O.synth_ir(rng, C, taps, dt)
        x = O.synth_audio(rng, nb * L, C, dt)
        eng = O.Engine(L, B, s, C)
        assert eng.set_coeff(h) == 0
        rc, y = eng.run(x)
        assert rc == 0
        yd = np.stack([O.direct_conv(x[:, c], h[c][:B * L]) for c in range(C)], axis=1)
        np.savez_compressed(os.path.join(HERE, name + ".npz"),
                            params=np.array([s, L, B, C, taps, nb]), h=np.stack(h), x=x,
                            y_direct=yd, y_oracle=y)
        print(name, "oracle vs direct: %.3g" % (np.abs(y - yd).max() / np.abs(yd).max()))


if __name__ == "__main__":
    main()
