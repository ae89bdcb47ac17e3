// preprocessor_hip.hpp -- the reference's offline drivers (brutefir/preprocessor.cpp)
// as batch jobs on the GPU engine.  Same call sequences as the reference, each loop
// of run() calls replaced by one brutefir::run_blocks; impulses and noise arrive as
// arrays because sound-file I/O (libsndfile) is outside the convolution path.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "brutefir_hip.hpp"

namespace preprocessor {

// util::get_next_multiple (brutefir/util.cpp:46-56)
inline uint32_t get_next_multiple(uint32_t value, uint32_t factor)
{
    uint32_t multiple = factor;
    while (value > multiple) multiple += factor;
    return multiple;
}

// One impulse response as buffer::load_from_snd_file hands it over: interleaved
// frames in working precision (float for realsize 4, double for 8) plus the
// user scale of preprocessor.cpp:176.
struct impulse_array {
    const void *frames;
    int n_channels;
    int n_frames;
    double scale;
};

namespace detail {
// buffer::deinterlace (brutefir/buffer.cpp:343-390): one contiguous array per channel
inline std::vector<std::vector<uint8_t>> deinterlace(const void *buf, int C, int frames, int realsize)
{
    std::vector<std::vector<uint8_t>> out(C, std::vector<uint8_t>((size_t)frames * realsize));
    const uint8_t *src = (const uint8_t *)buf;
    for (int f = 0; f < frames; f++)
        for (int c = 0; c < C; c++)
            memcpy(&out[c][(size_t)f * realsize], src + ((size_t)f * C + c) * realsize, realsize);
    return out;
}
inline int fmt_for(int realsize) { return realsize == 4 ? BF_SAMPLE_FORMAT_FLOAT_LE : BF_SAMPLE_FORMAT_FLOAT64_LE; }
}  // namespace detail

// preprocessor::convolve_impulses (preprocessor.cpp:33-233).  `out` receives g_frames
// interleaved frames (what the reference saves to its cache WAV).  false on error.
inline bool convolve_impulses(const std::vector<impulse_array> &impulses, int filter_length, int realsize,
                              std::vector<uint8_t> *out, int *out_channels, int *out_frames, int device = 0)
{
    int g_channels = 0, g_frames = 0;
    for (const auto &im : impulses) {
        if (im.n_frames > g_frames) g_frames = im.n_frames;
        if (g_channels != 0 && g_channels != im.n_channels) return false;   // the reference throws (:76)
        g_channels = im.n_channels;
    }
    if (impulses.empty() || g_channels < 1) return false;
    const int length = (int)get_next_multiple((uint32_t)g_frames, (uint32_t)filter_length);
    const int filter_blocks = length / filter_length;
    brutefir filter(filter_length, filter_blocks, realsize, g_channels, detail::fmt_for(realsize),
                    detail::fmt_for(realsize), 44100, false, device);
    const size_t bytes = (size_t)length * g_channels * realsize;
    std::vector<uint8_t> inbuf(bytes), outbuf(bytes, 0);
    // a dirac for the initial coefficients (coeff::load_dirac_coeff, coeff.cpp:33-59)
    std::vector<std::vector<uint8_t>> coeffs(g_channels, std::vector<uint8_t>((size_t)filter_length * realsize, 0));
    for (auto &c : coeffs) { if (realsize == 4) ((float *)c.data())[0] = 1.0f; else ((double *)c.data())[0] = 1.0; }
    std::vector<void *> ptrs(g_channels);
    for (int c = 0; c < g_channels; c++) ptrs[c] = coeffs[c].data();
    if (filter.set_coeff(ptrs.data(), g_channels, filter_length, filter_blocks, 1.0) != 0) return false;
    for (const auto &im : impulses) {
        memset(inbuf.data(), 0, bytes);                                    // zero padded to the filter span
        memcpy(inbuf.data(), im.frames, (size_t)im.n_frames * g_channels * realsize);
        if (filter.run_blocks(inbuf.data(), outbuf.data(), filter_blocks) != 0) return false;
        // the output becomes the coefficients of the next pass (:169-178; note the length argument)
        coeffs = detail::deinterlace(outbuf.data(), g_channels, length, realsize);
        for (int c = 0; c < g_channels; c++) ptrs[c] = coeffs[c].data();
        if (filter.set_coeff(ptrs.data(), g_channels, filter_length, filter_blocks, im.scale) != 0) return false;
    }
    out->assign(outbuf.begin(), outbuf.begin() + (size_t)g_frames * g_channels * realsize);
    *out_channels = g_channels;
    *out_frames = g_frames;
    return true;
}

// preprocessor::calculate_attenuation (preprocessor.cpp:249-412).  coeffs: interleaved
// [n_frames][n_channels] impulse; noise: filter_length*filter_blocks interleaved frames of
// full-scale white noise (the reference draws them with buffer::load_white_noise).
inline bool calculate_attenuation(const void *coeffs, int n_channels, int n_frames, int filter_length,
                                  int realsize, const void *noise, double *attenuation, int device = 0)
{
    *attenuation = 0;
    const int length = (int)get_next_multiple((uint32_t)n_frames, (uint32_t)filter_length);
    const int filter_blocks = length / filter_length;
    brutefir filter(filter_length, filter_blocks, realsize, n_channels, detail::fmt_for(realsize),
                    detail::fmt_for(realsize), 44100, false, device);
    std::vector<uint8_t> padded((size_t)length * n_channels * realsize, 0);
    memcpy(padded.data(), coeffs, (size_t)n_frames * n_channels * realsize);
    auto taps = detail::deinterlace(padded.data(), n_channels, length, realsize);
    std::vector<void *> ptrs(n_channels);
    for (int c = 0; c < n_channels; c++) ptrs[c] = taps[c].data();
    if (filter.set_coeff(ptrs.data(), n_channels, filter_length, filter_blocks, 1.0) != 0) return false;   // :313
    std::vector<uint8_t> outbuf((size_t)length * n_channels * realsize, 0);
    double max_value = 0;
    if (filter.run_blocks(const_cast<void *>(noise), outbuf.data(), filter_blocks) == 0) {
        // every run() succeeded: the largest |y| over all channels is what the engine's overflow bookkeeping tracks
        for (int c = 0; c < n_channels; c++) {
            bfir_overflow of;
            bfir_engine_get_overflow(filter.handle(), c, &of);
            if (of.largest > max_value) max_value = of.largest;
        }
    } else {
        // Some run() failed.  The reference's loop (preprocessor.cpp:329-356) skips the scan of THAT block and goes
        // on; and because brutefir::run returns before it flips curbuf and advances blockcounter (brutefir.cpp:316-321,
        // 337-340), the failed block never enters the history: the next call overwrites its delay-line slot and its
        // half of the time buffer.  Replayed block by block; after a failure the engine is rebuilt from the blocks
        // that were kept so far (an error path: no need to be fast).
        const size_t blk = (size_t)filter_length * n_channels * realsize;
        std::vector<int> kept;
        std::vector<uint8_t> hist, scratch;
        brutefir *f = nullptr;
        auto rebuild = [&]() -> bool {
            delete f;
            f = new brutefir(filter_length, filter_blocks, realsize, n_channels, detail::fmt_for(realsize),
                             detail::fmt_for(realsize), 44100, false, device);
            if (f->set_coeff(ptrs.data(), n_channels, filter_length, filter_blocks, 1.0) != 0) return false;
            if (kept.empty()) return true;
            hist.resize(kept.size() * blk); scratch.resize(hist.size());
            for (size_t i = 0; i < kept.size(); i++) memcpy(&hist[i * blk], (const uint8_t *)noise + (size_t)kept[i] * blk, blk);
            return f->run_blocks(hist.data(), scratch.data(), (int)kept.size()) == 0;
        };
        bool ok = rebuild();
        for (int n = 0; ok && n < filter_blocks; n++) {
            if (f->run((uint8_t *)const_cast<void *>(noise) + (size_t)n * blk, outbuf.data()) == 0) {
                kept.push_back(n);
                for (size_t i = 0; i < (size_t)filter_length * n_channels; i++) {   // :336-354, NaN never compares greater
                    const double v = realsize == 4 ? (double)fabsf(((const float *)outbuf.data())[i]) : fabs(((const double *)outbuf.data())[i]);
                    if (v > max_value) max_value = v;
                }
            } else {
                ok = rebuild();
            }
        }
        delete f;
        if (!ok) return false;
    }
    if (max_value > 1) *attenuation = -20.0 * log10(max_value);              // -TO_DB, util.hpp:15
    return true;
}

}  // namespace preprocessor
