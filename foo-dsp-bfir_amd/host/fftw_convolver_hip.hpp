// fftw_convolver_hip.hpp -- header-compatible stand-in for the reference's
// convolver class.  Same class name, same public methods with the same
// arguments and meaning as brutefir/fftw_convolver.hpp:28-166, so
// brutefir.cpp / coeff.cpp shaped callers compile against it unchanged; every
// method forwards to the stage-level C ABI (include/bfir_hip.h) and is
// computed by the gfx950 kernels.  No arithmetic happens on the host.
//
// Differences a maintainer must know:
//  * ctor failure throws std::runtime_error instead of the reference's bare
//    `throw;` (fftw_convolver.cpp:64-74) -- both are caught by the
//    `catch (...)` at brutefir.cpp:723.
//  * coefficient blocks returned by convolver_coeffs2cbuf are freed with
//    bfir_aligned_free (the reference caller uses _aligned_free,
//    brutefir.cpp:844-854).
//  * create_fft_plan returns an opaque token, not an FFTW plan; equalizer.cpp,
//    which executes the plan with FFTW directly, is outside this path.
//  * convolver_debug_dump_cbuf (text-file dump) and the convolver_td_* family (used only
//    by the dead `delay` class) are not provided.
#pragma once
#include <stdexcept>

#include "bfir_types.hpp"

class dither;   // accepted for signature compatibility; float outputs never use it

struct _td_conv_t_;
typedef struct _td_conv_t_ td_conv_t;

class fftw_convolver {
public:
    fftw_convolver(int length, int realsize, dither *dither_unused, int device = 0)
    {
        (void)dither_unused;
        int err = 0;
        m_c = bfir_convolver_create(length, realsize, device, &err);
        if (!m_c) throw std::runtime_error(bfir_strerror(err));
    }
    ~fftw_convolver() { bfir_convolver_destroy(m_c); }
    fftw_convolver(const fftw_convolver &) = delete;
    fftw_convolver &operator=(const fftw_convolver &) = delete;

    // fftw_convolver.cpp:156-185.  postprocess is always NULL in the tree.
    void convolver_raw2cbuf(void *rawbuf, void *cbuf, void *next_cbuf, struct buffer_format_t *bf,
                            void (*postprocess)(void *realbuf, int n_samples, void *arg), void *pp_arg)
    {
        (void)pp_arg;
        m_last = postprocess ? BFIR_ERR_UNSUPPORTED
                             : bfir_convolver_raw2cbuf(m_c, rawbuf, cbuf, next_cbuf, (const bfir_buffer_format *)bf);
    }
    // :187-212
    void convolver_time2freq(void *input_cbuf, void *output_cbuf)
    {
        m_last = bfir_convolver_time2freq(m_c, input_cbuf, output_cbuf);
    }
    // :214-229
    void convolver_mixnscale(void *input_cbufs[], void *output_cbuf, double scales[], int n_bufs, int mixmode)
    {
        m_last = bfir_convolver_mixnscale(m_c, input_cbufs, output_cbuf, scales, n_bufs, mixmode);
    }
    // :231-273
    void convolver_convolve_inplace(void *cbuf, void *coeffs)
    {
        m_last = bfir_convolver_convolve_inplace(m_c, cbuf, coeffs);
    }
    void convolver_convolve(void *input_cbuf, void *coeffs, void *output_cbuf)
    {
        m_last = bfir_convolver_convolve(m_c, input_cbuf, coeffs, output_cbuf);
    }
    void convolver_convolve_add(void *input_cbuf, void *coeffs, void *output_cbuf)
    {
        m_last = bfir_convolver_convolve_add(m_c, input_cbuf, coeffs, output_cbuf);
    }
    // :350-375
    void convolver_freq2time(void *input_cbuf, void *output_cbuf)
    {
        m_last = bfir_convolver_freq2time(m_c, input_cbuf, output_cbuf);
    }
    // :405-466.  Dither applies to integer formats only (:421); float formats ignore the flag.
    void convolver_cbuf2raw(void *cbuf, void *outbuf, struct buffer_format_t *bf, bool apply_dither,
                            struct dither_state_t *dither_state, struct bfoverflow_t *overflow)
    {
        (void)apply_dither; (void)dither_state;
        m_last = bfir_convolver_cbuf2raw(m_c, cbuf, outbuf, (const bfir_buffer_format *)bf, (bfir_overflow *)overflow);
    }
    // :468-472
    int convolver_cbufsize(void) { return bfir_convolver_cbufsize(m_c); }
    // :474-537
    void *convolver_coeffs2cbuf(void *coeffs, int n_coeffs, double scale, void *optional_dest)
    {
        return bfir_convolver_coeffs2cbuf(m_c, coeffs, n_coeffs, scale, optional_dest);
    }
    // ---- declared by the reference, called by nothing in its tree ----
    // :275-321.  buffer_cbuf: 1.5 cbufs.
    void convolver_crossfade_inplace(void *input_cbuf, void *crossfade_cbuf, void *buffer_cbuf)
    {
        m_last = bfir_convolver_crossfade_inplace(m_c, input_cbuf, crossfade_cbuf, buffer_cbuf);
    }
    // :323-348
    void convolver_dirac_convolve(void *input_cbuf, void *output_cbuf)
    {
        m_last = bfir_convolver_dirac_convolve(m_c, input_cbuf, output_cbuf);
    }
    void convolver_dirac_convolve_inplace(void *cbuf) { m_last = bfir_convolver_dirac_convolve_inplace(m_c, cbuf); }
    // :377-403.  buffer_cbuf: 1.5 cbufs, cleared before the first call, kept afterwards.
    void convolver_convolve_eval(void *input_cbuf, void *buffer_cbuf, void *output_cbuf)
    {
        m_last = bfir_convolver_convolve_eval(m_c, input_cbuf, buffer_cbuf, output_cbuf);
    }
    // :539-567
    void convolver_runtime_coeffs2cbuf(void *src, void *dest)
    {
        m_last = bfir_convolver_runtime_coeffs2cbuf(m_c, src, dest);
    }
    // :569-602
    bool convolver_verify_cbuf(void *cbufs[], int n_cbufs) { return bfir_convolver_verify_cbuf(m_c, cbufs, n_cbufs) == 1; }

    // :653-695: plans are device tables owned by the convolver; the token is only good
    // for passing back to destroy_fft_plan.
    void *create_fft_plan(int, int, int) { return m_c; }
    void destroy_fft_plan(int, int, int) {}

    // result of the last call (the reference's methods return void)
    int last_status() const { return m_last; }

private:
    bfir_convolver *m_c = nullptr;
    int m_last = BFIR_OK;
};
