// wav_io.hpp -- minimal writer/reader for the WAV files the reference caches derived
// impulse responses in (libsndfile SF_FORMAT_WAV | SF_FORMAT_FLOAT / _DOUBLE, little
// endian: brutefir/buffer.cpp:107-139).  Unknown chunks are skipped on read.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace wav_io {

inline void put32(std::vector<uint8_t> &v, uint32_t x) { for (int i = 0; i < 4; i++) v.push_back((uint8_t)(x >> (8 * i))); }
inline void put16(std::vector<uint8_t> &v, uint16_t x) { v.push_back((uint8_t)x); v.push_back((uint8_t)(x >> 8)); }
inline void puts4(std::vector<uint8_t> &v, const char *s) { v.insert(v.end(), s, s + 4); }

// frames: interleaved float (realsize 4) or double (realsize 8)
inline bool save_float(const std::string &path, const void *frames, int n_channels, int n_frames, int realsize,
                       int sampling_rate)
{
    const uint32_t data_bytes = (uint32_t)n_frames * n_channels * realsize;
    std::vector<uint8_t> h;
    puts4(h, "RIFF"); put32(h, 4 + 24 + 12 + 8 + data_bytes + (data_bytes & 1)); puts4(h, "WAVE");
    puts4(h, "fmt "); put32(h, 16); put16(h, 3); put16(h, (uint16_t)n_channels); put32(h, (uint32_t)sampling_rate);
    put32(h, (uint32_t)sampling_rate * n_channels * realsize); put16(h, (uint16_t)(n_channels * realsize));
    put16(h, (uint16_t)(8 * realsize));
    puts4(h, "fact"); put32(h, 4); put32(h, (uint32_t)n_frames);
    puts4(h, "data"); put32(h, data_bytes);
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = fwrite(h.data(), 1, h.size(), f) == h.size() && fwrite(frames, 1, data_bytes, f) == data_bytes;
    if (data_bytes & 1) ok = ok && fputc(0, f) != EOF;
    return fclose(f) == 0 && ok;
}

// IEEE float files only (what the caches hold); returns interleaved samples in their stored width
inline bool load_float(const std::string &path, std::vector<uint8_t> *frames, int *n_channels, int *n_frames,
                       int *realsize, int *sampling_rate)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    std::vector<uint8_t> raw;
    uint8_t buf[65536];
    size_t got;
    while ((got = fread(buf, 1, sizeof(buf), f)) > 0) raw.insert(raw.end(), buf, buf + got);
    fclose(f);
    auto u32 = [&](size_t p) { return (uint32_t)raw[p] | (uint32_t)raw[p + 1] << 8 | (uint32_t)raw[p + 2] << 16 | (uint32_t)raw[p + 3] << 24; };
    auto u16 = [&](size_t p) { return (uint16_t)(raw[p] | raw[p + 1] << 8); };
    if (raw.size() < 12 || memcmp(raw.data(), "RIFF", 4) || memcmp(raw.data() + 8, "WAVE", 4)) return false;
    int tag = 0, ch = 0, bits = 0, rate = 0;
    size_t pos = 12, dpos = 0, dsize = 0;
    while (pos + 8 <= raw.size()) {
        const uint32_t size = u32(pos + 4);
        if (!memcmp(raw.data() + pos, "fmt ", 4) && size >= 16) { tag = u16(pos + 8); ch = u16(pos + 10); rate = (int)u32(pos + 12); bits = u16(pos + 22); }
        else if (!memcmp(raw.data() + pos, "data", 4)) { dpos = pos + 8; dsize = size; }
        pos += 8 + size + (size & 1);
    }
    if (tag != 3 || (bits != 32 && bits != 64) || ch < 1 || dpos == 0 || dpos + dsize > raw.size()) return false;
    frames->assign(raw.begin() + dpos, raw.begin() + dpos + dsize);
    *n_channels = ch; *realsize = bits / 8; *sampling_rate = rate; *n_frames = (int)(dsize / (ch * (bits / 8)));
    return true;
}

}  // namespace wav_io
