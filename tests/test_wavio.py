"""The minimal WAV reader/writer used for the reference's cache files (no GPU)."""
import struct

import numpy as np
import pytest


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_float_wav_roundtrip(tmp_path, dt):
    from foo_dsp_bfir_amd import wavio
    x = np.random.default_rng(0).uniform(-1, 1, (1001, 3)).astype(dt)
    p = str(tmp_path / "a.wav")
    wavio.write_wav_float(p, x, 44100)
    raw = open(p, "rb").read()
    assert raw[:4] == b"RIFF" and raw[8:16] == b"WAVEfmt " and struct.unpack("<H", raw[20:22])[0] == 3
    y, rate = wavio.read_wav(p)
    assert rate == 44100 and y.dtype == dt and np.array_equal(x, y)


def test_reader_skips_unknown_chunks_and_reads_pcm16(tmp_path):
    from foo_dsp_bfir_amd import wavio
    pcm = np.array([[0, 16384], [-32768, 32767]], dtype="<i2")
    fmt = struct.pack("<HHIIHH", 1, 2, 8000, 8000 * 4, 4, 16)
    peak = b"PEAK" + struct.pack("<I", 8) + b"\x00" * 8            # libsndfile writes such chunks
    body = b"WAVE" + b"fmt " + struct.pack("<I", 16) + fmt + peak + b"data" + struct.pack("<I", pcm.nbytes) + pcm.tobytes()
    p = tmp_path / "b.wav"
    p.write_bytes(b"RIFF" + struct.pack("<I", len(body)) + body)
    y, rate = wavio.read_wav(str(p))
    assert rate == 8000 and np.array_equal(y, pcm.astype(np.float32) / np.float32(32768.0))


def test_djb_hash_uses_signed_chars():
    from foo_dsp_bfir_amd.equalizer import djb_hash
    assert djb_hash(b"") == 5381 and djb_hash(b"a") == (5381 * 33 + 97) & 0xFFFFFFFF
    assert djb_hash(bytes([200])) == (5381 * 33 - 56) & 0xFFFFFFFF
