"""Minimal WAV reader/writer for the reference's on-disk caches: libsndfile's
SF_FORMAT_WAV | SF_FORMAT_FLOAT / SF_FORMAT_DOUBLE, little endian
(brutefir/buffer.cpp:107-139).  Unknown chunks ('PEAK', 'fact', 'LIST' ...) are skipped
on read; PCM 16/24/32 files are read too (impulse responses often are PCM)."""
import struct

import numpy as np


def write_wav_float(path, frames, sampling_rate):
    """frames: [n, channels] float32 or float64 -> WAVE_FORMAT_IEEE_FLOAT file."""
    x = np.ascontiguousarray(frames)
    assert x.ndim == 2 and x.dtype in (np.float32, np.float64)
    n, ch = x.shape
    bits = 8 * x.itemsize
    data = x.astype(x.dtype.newbyteorder("<")).tobytes()
    fmt = struct.pack("<HHIIHH", 3, ch, sampling_rate, sampling_rate * ch * x.itemsize, ch * x.itemsize, bits)
    fact = struct.pack("<I", n)
    body = (b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"fact" + struct.pack("<I", 4) + fact +
            b"data" + struct.pack("<I", len(data)) + data + (b"\x00" if len(data) & 1 else b""))
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def read_wav(path):
    """Returns (frames [n, channels] as float32/float64, sampling_rate).  PCM is scaled to [-1, 1)."""
    raw = open(path, "rb").read()
    if raw[:4] != b"RIFF" or raw[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE file")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(raw):
        cid, size = raw[pos:pos + 4], struct.unpack("<I", raw[pos + 4:pos + 8])[0]
        body = raw[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
            if fmt[0] == 0xFFFE and size >= 26:      # WAVE_FORMAT_EXTENSIBLE: sub-format GUID's first word
                fmt = (struct.unpack("<H", body[24:26])[0],) + fmt[1:]
        elif cid == b"data":
            data = body
        pos += 8 + size + (size & 1)
    if fmt is None or data is None:
        raise ValueError("missing fmt or data chunk")
    tag, ch, rate, _, _, bits = fmt
    if tag == 3 and bits in (32, 64):
        x = np.frombuffer(data, dtype="<f4" if bits == 32 else "<f8")
        x = x.astype(np.float32 if bits == 32 else np.float64)
    elif tag == 1 and bits == 16:
        x = np.frombuffer(data, dtype="<i2").astype(np.float32) / np.float32(32768.0)
    elif tag == 1 and bits == 32:
        x = (np.frombuffer(data, dtype="<i4").astype(np.float64) / 2147483648.0)
    elif tag == 1 and bits == 24:
        b = np.frombuffer(data, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = (np.where(v >= 1 << 23, v - (1 << 24), v).astype(np.float64) / 8388608.0)
    else:
        raise ValueError("unsupported WAV encoding tag=%d bits=%d" % (tag, bits))
    n = x.size // ch
    return x[:n * ch].reshape(n, ch), rate
