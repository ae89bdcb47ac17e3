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
//  * create_fft_plan returns a bfir_fft_plan* (an R2HC / HC2R plan of 2^order reals on the GPU), not an
//    FFTW plan: a caller that executed it with fftw[f]_execute_r2r (equalizer.cpp:262, 357) calls
//    bfir_fft_plan_execute(plan, in, out) instead.
//  * td_conv_t (convolver_td_new) carries the reference's four fields plus the device-side handle
//    `td`; td_conv_t.coeffs is a read-only host copy of the filter's spectrum.  The reference never
//    frees a td_conv_t; convolver_td_free is the addition that does.
#pragma once
#include <stdexcept>

#include <map>

#include "bfir_types.hpp"
#include "dither_hip.hpp"

// fftw_convolver.hpp:18-26, plus the handle of the device-side object
struct _td_conv_t_ {
    void *fftplan;
    void *ifftplan;
    void *coeffs;
    int blocklen;
    bfir_td_conv *td;
};
typedef struct _td_conv_t_ td_conv_t;

class fftw_convolver {
public:
    // fftw_convolver.cpp:51-138.  `dither` is borrowed (owned by brutefir, brutefir.cpp:53-54) and only
    // used by convolver_cbuf2raw with apply_dither on an integer format.
    fftw_convolver(int length, int realsize, dither *dither, int device = 0)
        : m_dither(dither), m_realsize(realsize), m_device(device)
    {
        int err = 0;
        m_c = bfir_convolver_create(length, realsize, device, &err);
        if (!m_c) throw std::runtime_error(bfir_strerror(err));
    }
    ~fftw_convolver()
    {
        for (auto &kv : m_plans) bfir_fft_plan_destroy(kv.second);   // fftw_convolver.cpp:140-151
        bfir_convolver_destroy(m_c);
    }
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
    // :405-466.  Dither applies to integer formats only (:421, :444); a missing dither instance is the
    // reference's "Dither instance not set." failure (:412-416).
    void convolver_cbuf2raw(void *cbuf, void *outbuf, struct buffer_format_t *bf, bool apply_dither,
                            struct dither_state_t *dither_state, struct bfoverflow_t *overflow)
    {
        if (apply_dither && !bf->sf.isfloat)
            m_last = bfir_convolver_cbuf2raw_dither(m_c, m_dither ? m_dither->handle() : nullptr, cbuf, outbuf,
                                                    (const bfir_buffer_format *)bf, (bfir_dither_state *)dither_state,
                                                    (bfir_overflow *)overflow);
        else
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

    // :604-651: the coefficient lists behind cbufs[] as a text file, one "%.16e" line per tap
    void convolver_debug_dump_cbuf(const char filename[], void *cbufs[], int n_cbufs)
    {
        m_last = bfir_convolver_debug_dump_cbuf(m_c, filename, cbufs, n_cbufs);
    }

    // :697-777: the one-block convolver of the (dead) delay class (delay.cpp:174, 196, 240-257)
    int convolver_td_block_length(int n_coeffs) { return bfir_td_block_length(n_coeffs); }
    td_conv_t *convolver_td_new(void *coeffs, int n_coeffs)
    {
        const int blocklen = convolver_td_block_length(n_coeffs);
        if (blocklen == -1) return nullptr;
        int err = 0, order = 1;
        bfir_td_conv *td = bfir_td_new(coeffs, n_coeffs, m_realsize, m_device, &err);
        m_last = err;
        if (!td) return nullptr;
        while ((1 << order) < blocklen) order++;
        td_conv_t *tdc = new td_conv_t();
        tdc->fftplan = create_fft_plan(order + 1, 0, 1);     // :726-727
        tdc->ifftplan = create_fft_plan(order + 1, 1, 1);
        tdc->coeffs = const_cast<void *>(bfir_td_coeffs(td));
        tdc->blocklen = blocklen;
        tdc->td = td;
        return tdc;
    }
    void convolver_td_convolve(td_conv_t *tdc, void *overlap_block)
    {
        m_last = tdc ? bfir_td_convolve(tdc->td, overlap_block) : BFIR_ERR_ARG;
    }
    void convolver_td_free(td_conv_t *tdc)
    {
        if (!tdc) return;
        bfir_td_destroy(tdc->td);
        delete tdc;
    }

    // :653-695: a plan per (order, invert, inplace), created on first use, owned by the convolver.
    // Returns a bfir_fft_plan* of 2^order reals: execute with bfir_fft_plan_execute(plan, in, out).
    void *create_fft_plan(int order, int invert, int inplace)
    {
        const int key = (order << 2) | ((invert ? 1 : 0) << 1) | (inplace ? 1 : 0);
        auto it = m_plans.find(key);
        if (it != m_plans.end()) return it->second;
        int err = 0;
        bfir_fft_plan *p = bfir_fft_plan_create(order, invert ? 1 : 0, inplace ? 1 : 0, m_realsize, m_device, &err);
        m_last = err;
        if (p) m_plans[key] = p;
        return p;
    }
    void destroy_fft_plan(int order, int invert, int inplace)
    {
        const int key = (order << 2) | ((invert ? 1 : 0) << 1) | (inplace ? 1 : 0);
        auto it = m_plans.find(key);
        if (it == m_plans.end()) return;
        bfir_fft_plan_destroy(it->second);
        m_plans.erase(it);
    }

    // result of the last call (the reference's methods return void)
    int last_status() const { return m_last; }

private:
    bfir_convolver *m_c = nullptr;
    dither *m_dither = nullptr;
    int m_realsize = 4, m_device = 0;
    std::map<int, bfir_fft_plan *> m_plans;
    int m_last = BFIR_OK;
};
