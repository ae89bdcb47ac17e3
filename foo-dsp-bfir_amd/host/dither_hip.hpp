// dither_hip.hpp -- the reference's `dither` class (brutefir/dither.hpp:13-77) over the C ABI:
// same constructor arguments (it fills the caller's dither_state_t array, dither.cpp:105-109) and
// the same preloop method.  The per-sample quantisers dither{f,d}_real2int_* are not host functions
// here: real2raw runs on the GPU (csrc/dither.hip), reached through
// fftw_convolver::convolver_cbuf2raw(..., apply_dither, dither_state, overflow).
#pragma once
#include <stdexcept>

#include "bfir_types.hpp"

class dither {
public:
    dither(int n_channels, int sample_rate, int realsize, int max_size, int max_samples_per_loop,
           struct dither_state_t *dither_state, int device = 0)
    {
        int err = 0;
        m_d = bfir_dither_create(n_channels, sample_rate, realsize, max_size, max_samples_per_loop,
                                 (bfir_dither_state *)dither_state, device, &err);
        if (!m_d) throw std::runtime_error(bfir_strerror(err));   // the reference's bare `throw;`, dither.cpp:58
    }
    ~dither() { bfir_dither_destroy(m_d); }
    dither(const dither &) = delete;
    dither &operator=(const dither &) = delete;

    // dither.cpp:127-139
    void dither_preloop_real2int_hp_tpdf(struct dither_state_t *state, int samples_per_loop)
    {
        bfir_dither_preloop_real2int_hp_tpdf(m_d, (bfir_dither_state *)state, samples_per_loop);
    }

    bfir_dither *handle() { return m_d; }

private:
    bfir_dither *m_d = nullptr;
};
