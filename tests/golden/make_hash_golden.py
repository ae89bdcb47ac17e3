"""Generates tests/golden/djb_hash_ref.json from the REFERENCE's own hash.c, compiled in place
(oracle/Makefile target `ref` -> oracle/_ref/libref_hash.so; the source never enters the repo).

DJBHash (brutefir/hash.c:113-124) names the reference's on-disk caches: eq-<hash>-... for the equalizer's
impulse (equalizer.cpp:152-180, over the raw bytes of the freq | mag | phase doubles), file-<hash>-... for the
impulse pre-convolver (preprocessor.cpp:88-98) and ir-<hash>-... for resampled impulses (buffer.cpp:240-253), both
over a file name string.  The fixture holds inputs (hex) and the 32-bit values the reference's function returned
here; tests/test_wavio.py checks the Python and C++ mirrors against it on any machine.

    make -C oracle ref && python tests/golden/make_hash_golden.py
"""
import ctypes as C
import json
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
ISO = [20, 25, 31.5, 40, 50, 63, 80, 100, 125, 160, 200, 250, 315, 400, 500, 630, 800, 1000, 1250, 1600,
       2000, 2500, 3150, 4000, 5000, 6300, 8000, 10000, 12500, 16000, 20000]


def band_blob(freq, mag, phase):
    return b"".join(struct.pack("<%dd" % len(v), *v) for v in (freq, mag, phase))


def main():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_hash.so"))
    lib.DJBHash.restype = C.c_uint
    lib.DJBHash.argtypes = [C.c_char_p, C.c_uint]
    rng = np.random.default_rng(2024)
    cases = []

    def add(kind, data, **extra):
        cases.append(dict(kind=kind, hex=data.hex(), djb=int(lib.DJBHash(data, len(data))), **extra))

    add("bytes", b"")
    add("bytes", b"a")
    add("bytes", bytes([200]))                      # a byte >= 128: the reference hashes plain (signed) chars
    add("bytes", bytes(range(256)))
    for n in (1, 7, 64, 1000):
        add("bytes", rng.integers(0, 256, n, dtype=np.uint8).tobytes())
    # equalizer band tables as equalizer::make_filename sees them (n_bands doubles of freq, mag, phase)
    flat = [0.0] * len(ISO)
    add("bands", band_blob(ISO, flat, flat), n_bands=len(ISO))
    for _ in range(6):
        nb = int(rng.integers(1, len(ISO) + 1))
        idx = sorted(rng.choice(len(ISO), nb, replace=False))
        freq = [float(ISO[i]) for i in idx]
        mag = [float(x) for x in np.round(rng.uniform(-12, 12, nb), 1)]
        ph = [float(x) for x in np.round(rng.uniform(-180, 180, nb), 0)]
        add("bands", band_blob(freq, mag, ph), n_bands=nb, freq=freq, mag=mag, phase=ph)
    # file-name strings as preprocessor.cpp:88 / buffer.cpp:245 hash them
    for name in ("C:\\impulses\\room_L.wav", "C:\\impulses\\room_L.wavC:\\impulses\\room_R.wav",
                 "/tmp/ir/häll-48k.wav".encode("latin-1").decode("latin-1"), "x" * 300):
        add("name", name.encode("latin-1"))
    out = os.path.join(HERE, "djb_hash_ref.json")
    json.dump(dict(source="DJBHash of /root/reference/brutefir/hash.c:113-124 compiled in place with gcc -O2 "
                          "(oracle/Makefile `ref`); inputs as hex, outputs as returned",
                   cases=cases), open(out, "w"), indent=1)
    print("wrote", out, len(cases), "cases")


if __name__ == "__main__":
    main()
