"""MI355X-native partitioned FIR convolution engine (hot path of vsu/foo-dsp-bfir).

Product code: HIP kernels + C ABI in csrc/ (libbfir_hip.so) and the host-side
mirror of the reference's interface.  Nothing here imports oracle/."""
from . import _build
from ._lib import (ERR_ARG, ERR_COEFF, ERR_HIP, ERR_IO, ERR_NO_DEVICE, ERR_NONFINITE, ERR_STATE,
                   ERR_UNSUPPORTED, BfirError, BufferFormat, Overflow, SampleFormat, MIXMODE_INPUT, MIXMODE_OUTPUT,
                   SAMPLE_FORMAT_FLOAT64_LE, SAMPLE_FORMAT_FLOAT_LE, load, library_path)
from .convolver import Dither, FftwConvolver, TdConv
from .engine import Brutefir, pinned_frames
from .equalizer import Equalizer, FftPlan

build = _build.build

__all__ = ["Brutefir", "pinned_frames", "FftwConvolver", "TdConv", "Dither", "Equalizer", "FftPlan", "BfirError", "BufferFormat", "Overflow", "SampleFormat",
           "MIXMODE_INPUT", "MIXMODE_OUTPUT", "SAMPLE_FORMAT_FLOAT_LE", "SAMPLE_FORMAT_FLOAT64_LE",
           "build", "load", "library_path", "ERR_ARG", "ERR_COEFF", "ERR_HIP", "ERR_NO_DEVICE",
           "ERR_NONFINITE", "ERR_STATE", "ERR_UNSUPPORTED", "ERR_IO"]
