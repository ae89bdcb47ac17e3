// equalizer_hip.hpp -- the reference's graphic equalizer renderer (brutefir/equalizer.hpp:67-115,
// equalizer.cpp) over the GPU: the band placement of generate() is 33 scalars of host
// arithmetic exactly as in the reference; the taps-bin spectrum and its HC2R transform run in
// bfir_equalizer_render.  Writes / re-uses the same cache WAV name scheme (make_filename).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/bfir_hip.h"
#include "wav_io.hpp"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

class equalizer {
public:
    static constexpr int BAND_COUNT = 31;   // ISO_BANDS_SIZE, equalizer.hpp:12-14

    // equalizer.cpp:29-69
    equalizer(int block_length, int n_blocks, int realsize, int n_channels, int sampling_rate, int device = 0)
        : m_realsize(realsize), m_channels(n_channels), m_rate(sampling_rate), m_taps(block_length * n_blocks)
    {
        if (m_taps < 32 || (m_taps & (m_taps - 1))) throw std::runtime_error("Equalizer length is not a power of two.");
        int order = 0, err = 0;
        while ((1 << order) < m_taps) order++;
        m_plan = bfir_fft_plan_create(order, 1, 1, realsize, device, &err);   // create_fft_plan(order, true, true)
        if (!m_plan) throw std::runtime_error(bfir_strerror(err));
        static const double iso[BAND_COUNT] = {20, 25, 31.5, 40, 50, 63, 80, 100, 125, 160, 200, 250, 315, 400, 500, 630,
                                               800, 1000, 1250, 1600, 2000, 2500, 3150, 4000, 5000, 6300, 8000, 10000,
                                               12500, 16000, 20000};
        m_count = BAND_COUNT + 2;
        memset(m_mag, 0, sizeof(m_mag)); memset(m_phase, 0, sizeof(m_phase));
        m_freq[0] = 0.0; m_freq[m_count - 1] = (double)sampling_rate / 2.0;
        for (int n = 0; n < BAND_COUNT; n++) m_freq[n + 1] = iso[n];
    }
    ~equalizer() { bfir_fft_plan_destroy(m_plan); }
    equalizer(const equalizer &) = delete;
    equalizer &operator=(const equalizer &) = delete;

    // equalizer::generate (equalizer.cpp:86-140).  Returns the cache file's path (in `dir`);
    // the file holds taps/2 frames of n_channels identical channels.  Empty on failure.
    std::string generate(int n_bands, double *freq, double *mag, double *phase, const std::string &dir)
    {
        if (n_bands > BAND_COUNT) throw std::runtime_error("Number of bands exceeds limit.");
        for (int n = 0, i = 0; n < n_bands; n++) {
            while (freq[n] > m_freq[i]) i++;
            m_mag[i] = mag[n]; m_phase[i] = phase[n];
            i++;
        }
        m_mag[0] = m_mag[1];
        m_mag[m_count - 1] = m_mag[m_count - 2];
        for (int n = 0; n < m_count; n++) {
            m_freq[n] /= (double)m_rate;
            m_mag[n] = pow(10, m_mag[n] / 20);
            m_phase[n] /= (180 * M_PI);
        }
        const std::string path = dir + "/" + make_filename(n_bands, freq, mag, phase);
        FILE *probe = fopen(path.c_str(), "rb");
        if (probe) { fclose(probe); return path; }          // render only if the file does not exist (:127)
        const int half = m_taps >> 1;
        std::vector<uint8_t> ir((size_t)half * m_realsize);
        if (bfir_equalizer_render(m_plan, m_count, m_freq, m_mag, m_phase, ir.data()) != 0) return std::string();
        std::vector<uint8_t> inter((size_t)half * m_channels * m_realsize);   // buffer::interlace (:284)
        for (int f = 0; f < half; f++)
            for (int c = 0; c < m_channels; c++)
                memcpy(&inter[((size_t)f * m_channels + c) * m_realsize], &ir[(size_t)f * m_realsize], m_realsize);
        if (!wav_io::save_float(path, inter.data(), m_channels, half, m_realsize, m_rate)) return std::string();
        return path;
    }

    // DJBHash (brutefir/hash.c:113-124) over plain chars, which are signed on the reference's compiler and here;
    // pinned against the reference's own function by tests/golden/djb_hash_ref.json
    static unsigned int djb_hash(const char *str, size_t len)
    {
        unsigned int hash = 5381;
        for (size_t i = 0; i < len; i++) hash = ((hash << 5) + hash) + (unsigned int)(int)(signed char)str[i];
        return hash;
    }
    // equalizer::make_filename (:152-180): the hash of the raw freq | mag | phase doubles, in hex, then
    // taps/2, realsize, channels, sampling rate
    static std::string make_filename_for(int taps, int realsize, int n_channels, int sampling_rate, int n_bands,
                                         const double *freq, const double *mag, const double *phase)
    {
        std::vector<char> blob(3 * (size_t)n_bands * sizeof(double));
        memcpy(blob.data(), freq, n_bands * sizeof(double));
        memcpy(blob.data() + n_bands * sizeof(double), mag, n_bands * sizeof(double));
        memcpy(blob.data() + 2 * n_bands * sizeof(double), phase, n_bands * sizeof(double));
        char name[128];
        snprintf(name, sizeof(name), "eq-%x-%d-%d-%d-%d.wav", djb_hash(blob.data(), blob.size()), taps >> 1, realsize,
                 n_channels, sampling_rate);
        return name;
    }
    std::string make_filename(int n_bands, const double *freq, const double *mag, const double *phase) const
    {
        return make_filename_for(m_taps, m_realsize, m_channels, m_rate, n_bands, freq, mag, phase);
    }

private:
    int m_realsize, m_channels, m_rate, m_taps, m_count = 0;
    bfir_fft_plan *m_plan = nullptr;
    double m_freq[BAND_COUNT + 2], m_mag[BAND_COUNT + 2], m_phase[BAND_COUNT + 2];
};
