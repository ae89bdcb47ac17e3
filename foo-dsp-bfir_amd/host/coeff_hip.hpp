// coeff_hip.hpp -- coeff::preprocess_coeff (brutefir/coeff.cpp:292-354) over the
// convolver stand-in: cut an impulse response into coeff_blocks partitions of
// filter_length taps and pre-transform each on the GPU.
#pragma once
#include <cstdint>
#include <cstring>

#include "fftw_convolver_hip.hpp"

namespace coeff {

// Returns coeff_blocks host buffers (each convolver_cbufsize() bytes); the
// caller frees every block and the table with bfir_aligned_free, as the
// reference frees them with _aligned_free (brutefir.cpp:844-854).  An entry is
// NULL when its block held a NaN/Inf tap.
inline void **preprocess_coeff(fftw_convolver *convolver, void *coeffs, int filter_length, int coeff_blocks,
                               int coeff_length, int realsize, double scale)
{
    if (!coeffs) return nullptr;
    void **cbuf = (void **)bfir_aligned_malloc(sizeof(void *) * (size_t)coeff_blocks, 16);
    // a block that starts past the end of the impulse is all zero, one that straddles
    // the end is zero filled (coeff.cpp:315-339)
    void *zeros = bfir_aligned_malloc((size_t)filter_length * realsize, 16);
    memset(zeros, 0, (size_t)filter_length * realsize);
    for (int n = 0; n < coeff_blocks; n++) {
        const long start = (long)n * filter_length;
        uint8_t *src = (uint8_t *)coeffs + (size_t)start * realsize;
        if (start > coeff_length) cbuf[n] = convolver->convolver_coeffs2cbuf(zeros, filter_length, scale, nullptr);
        else if (start + filter_length > coeff_length)
            cbuf[n] = convolver->convolver_coeffs2cbuf(src, coeff_length - (int)start, scale, nullptr);
        else cbuf[n] = convolver->convolver_coeffs2cbuf(src, filter_length, scale, nullptr);
    }
    bfir_aligned_free(zeros);
    return cbuf;
}

}  // namespace coeff
