// bfir_types.hpp -- the handful of structs that cross the convolver boundary by
// pointer (brutefir/global.h:39-102).  Inside the reference tree its own
// global.h provides them (guard _GLOBAL_H_); outside it, these layout-identical
// definitions do.  The layouts are asserted against include/bfir_hip.h.
#pragma once
#include <cstddef>
#include <cstdint>

#include "../../include/bfir_hip.h"

#ifndef _GLOBAL_H_
#define BF_MAXCHANNELS BFIR_MAXCHANNELS
#define BF_SAMPLE_FORMAT_FLOAT_LE BFIR_SAMPLE_FORMAT_FLOAT_LE
#define BF_SAMPLE_FORMAT_FLOAT64_LE BFIR_SAMPLE_FORMAT_FLOAT64_LE

struct sample_format_t {
    bool isfloat, swap;
    int bytes, sbytes;
    double scale;
    int format;
};
struct buffer_format_t {
    sample_format_t sf;
    int sample_spacing;   // in samples
    int byte_offset;      // in bytes
};
struct dither_state_t {   // brutefir/global.h:63-69
    int randtab_ptr;
    int8_t *randtab;
    float sf[2];
    double sd[2];
};
struct bfoverflow_t {
    unsigned int n_overflows;
    int32_t intlargest;
    double largest;
    double max;
};
#endif

static_assert(sizeof(buffer_format_t) == sizeof(bfir_buffer_format), "buffer_format_t layout");
static_assert(offsetof(buffer_format_t, byte_offset) == offsetof(bfir_buffer_format, byte_offset), "buffer_format_t layout");
static_assert(sizeof(dither_state_t) == sizeof(bfir_dither_state), "dither_state_t layout");
static_assert(offsetof(dither_state_t, sd) == offsetof(bfir_dither_state, sd), "dither_state_t layout");
static_assert(sizeof(bfoverflow_t) == sizeof(bfir_overflow), "bfoverflow_t layout");
static_assert(offsetof(bfoverflow_t, max) == offsetof(bfir_overflow, max), "bfoverflow_t layout");
// 64-bit quantities of the C ABI are int64_t whatever the platform's `long` is (MSVC: 32 bits)
static_assert(sizeof(int64_t) == 8, "int64_t");
static_assert(__is_same(decltype(&bfir_engine_run_device),
                        int (*)(bfir_engine *, const void *, int64_t, void *, int64_t, int, void *)), "engine strides are int64_t");
static_assert(__is_same(decltype(&bfir_fft_plan_length), int64_t (*)(const bfir_fft_plan *)), "plan lengths are int64_t");

#ifndef CONVOLVER_MIXMODE_INPUT
#define CONVOLVER_MIXMODE_INPUT BFIR_MIXMODE_INPUT
#define CONVOLVER_MIXMODE_INPUT_ADD BFIR_MIXMODE_INPUT_ADD
#define CONVOLVER_MIXMODE_OUTPUT BFIR_MIXMODE_OUTPUT
#endif
