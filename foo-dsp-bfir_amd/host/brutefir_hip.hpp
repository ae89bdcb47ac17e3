// brutefir_hip.hpp -- the reference's engine class (brutefir/brutefir.hpp:15-128)
// over the fused GPU engine: same public interface, same return codes, so
// foo_dsp_bfir.cpp:279-345 and preprocessor.cpp:287-333 call it unchanged.
// All block processing happens in bfir_engine_run (include/bfir_hip.h).
#pragma once
#include <cmath>
#include <cstdio>
#include <cstring>

#include "bfir_types.hpp"

class brutefir {
public:
    // brutefir.cpp:21-44.  On failure the object stays uninitialised (is_initialized()
    // false, run() returns an error), as the reference's does.
    brutefir(int filter_length, int filter_blocks, int realsize, int channels, int in_format,
             int out_format, int sampling_rate, bool apply_dither, int device = 0)
        : m_channels(channels)
    {
        int err = 0;
        m_e = bfir_engine_create(filter_length, filter_blocks, realsize, channels, in_format, out_format,
                                 sampling_rate, apply_dither ? 1 : 0, device, &err);
        m_err = err;
        memset(m_last, 0, sizeof(m_last));
    }
    ~brutefir() { bfir_engine_destroy(m_e); }
    brutefir(const brutefir &) = delete;
    brutefir &operator=(const brutefir &) = delete;

    bool is_initialized() { return m_e && bfir_engine_is_initialized(m_e); }

    // brutefir.cpp:88-165 loads a sound file through libsndfile; file I/O is outside the
    // convolution path, so coefficients arrive as arrays (the overload below).
    int set_coeff(const wchar_t *, int, double) { return -1; }

    // brutefir.cpp:179-228: 0, or -2 when a tap is NaN/Inf.
    int set_coeff(void **coeffs, int n_coeffs, int length, int coeff_blocks, double scale)
    {
        if (!m_e) return -1;
        return bfir_engine_set_coeff(m_e, (const void *const *)coeffs, n_coeffs, length, coeff_blocks, scale);
    }

    // brutefir.cpp:244-343: one block of filter_length interleaved frames; 0 or -1.
    int run(void *inbuf, void *outbuf) { return run_blocks(inbuf, outbuf, 1); }

    // n consecutive blocks in one call: what the offline drivers' loops
    // (preprocessor.cpp:145, 329-333) amount to.
    int run_blocks(void *inbuf, void *outbuf, int n_blocks)
    {
        if (!m_e) return -1;
        int rc = bfir_engine_run(m_e, inbuf, outbuf, n_blocks);
        return rc == 0 ? 0 : -1;
    }

    // brutefir.cpp:346-367
    void reset()
    {
        if (m_e) bfir_engine_reset(m_e);
        memset(m_last, 0, sizeof(m_last));
    }

    // brutefir.cpp:370-388 + print_overflows :585-629: report when any channel changed.
    void check_overflows()
    {
        if (!m_e) return;
        bfir_overflow of[BFIR_MAXCHANNELS];
        bool changed = false, any = false;
        for (int n = 0; n < m_channels; n++) {
            bfir_engine_get_overflow(m_e, n, &of[n]);
            changed |= memcmp(&of[n], &m_last[n], sizeof(bfir_overflow)) != 0;
            any |= of[n].n_overflows > 0;
        }
        if (!changed) return;
        memcpy(m_last, of, sizeof(bfir_overflow) * m_channels);
        if (!any) return;
        for (int n = 0; n < m_channels; n++) {
            double peak = of[n].largest;
            if (peak < (double)of[n].intlargest) peak = (double)of[n].intlargest;
            char line[96];
            if (peak != 0.0) {
                double db = 20.0 * log10(peak / of[n].max);
                if (db == 0.0) db = -0.0;
                snprintf(line, sizeof(line), "peak: %d/%u/%+.2f ", n, of[n].n_overflows, db);
            } else {
                snprintf(line, sizeof(line), "peak: %d/%u/-Inf ", n, of[n].n_overflows);
            }
            if (m_log) m_log(line);
        }
    }

    // pinfo-style sink for check_overflows (brutefir/pinfo.c:17-39)
    void set_log(bfir_log_fn fn) { m_log = fn; bfir_set_log_callback(fn); }
    int create_error() const { return m_err; }
    bfir_engine *handle() { return m_e; }

private:
    bfir_engine *m_e = nullptr;
    int m_channels = 0, m_err = 0;
    bfir_overflow m_last[BFIR_MAXCHANNELS];
    bfir_log_fn m_log = nullptr;
};
