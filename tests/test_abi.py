"""The C-ABI library on a machine without a GPU: it loads, exports every function
include/bfir_hip.h declares, its structs have the reference's layout, and it fails
loudly (no CPU fallback) when no device is present."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "bfir_hip.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bfir_[a-z0-9_]+)\s*\(", src)))


def test_header_is_plain_c():
    """extern "C", plain pointers and sizes: it must compile as C99."""
    subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-x", "c", HEADER], check=True)


def test_no_long_in_the_abi():
    """The reference's platform is MSVC (brutefir/brutefir.vcxproj:66-70), where `long` has 32 bits: every
    stride, count and length that can pass 2^31 is int64_t / size_t."""
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    assert not re.search(r"\blong\b", src), [l for l in src.splitlines() if re.search(r"\blong\b", l)]
    assert re.search(r"bfir_engine_run_device\([^)]*int64_t in_stride_bytes[^)]*int64_t out_stride_bytes", src)
    from importlib import import_module
    sigs = import_module("foo_dsp_bfir_amd._lib").SIGNATURES
    assert sigs["bfir_engine_run_device"][1][2] is C.c_int64 and sigs["bfir_engine_run_device"][1][4] is C.c_int64
    assert sigs["bfir_fft_plan_length"][0] is C.c_int64


def test_library_exports_every_declared_symbol(bfir):
    lib = bfir.load()
    names = _declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libbfir_hip.so does not export %s" % n
    from importlib import import_module
    sigs = import_module("foo_dsp_bfir_amd._lib").SIGNATURES
    assert sorted(sigs) == names    # the binding covers exactly the header


def test_struct_layouts_match_reference(bfir):
    # bfoverflow_t: uint, int32, double, double (brutefir/global.h:96-102)
    assert C.sizeof(bfir.Overflow) == 24
    assert bfir.Overflow.largest.offset == 8 and bfir.Overflow.max.offset == 16
    # sample_format_t: bool, bool, int, int, double, int (brutefir/global.h:39-47)
    assert bfir.SampleFormat.bytes.offset == 4 and bfir.SampleFormat.scale.offset == 16
    assert C.sizeof(bfir.SampleFormat) == 32
    assert bfir.BufferFormat.sample_spacing.offset == 32 and C.sizeof(bfir.BufferFormat) == 40


def test_no_device_means_error_not_fallback(bfir):
    lib = bfir.load()
    if lib.bfir_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(bfir.BfirError) as ei:
        bfir.Brutefir(1024, 2, 4, 2)
    assert ei.value.code == bfir.ERR_NO_DEVICE
    with pytest.raises(bfir.BfirError):
        bfir.FftwConvolver(1024, 4)
    assert b"no HIP device" in lib.bfir_strerror(bfir.ERR_NO_DEVICE)
    # the newer entry points fail the same way: no page-locked memory, no td convolver, no plan -- never a CPU stand-in
    assert not lib.bfir_pinned_malloc(4096)
    with pytest.raises(bfir.BfirError):
        bfir.pinned_frames(bfir.SAMPLE_FORMAT_FLOAT_LE, (16, 2))
    err = C.c_int(0)
    taps = (C.c_float * 4)(1, 2, 3, 4)
    assert not lib.bfir_td_new(taps, 4, 4, 0, C.byref(err)) and err.value == bfir.ERR_NO_DEVICE
    assert lib.bfir_td_block_length(5) == 8 and lib.bfir_td_block_length(1) == -1      # pure arithmetic: no device needed
    with pytest.raises(bfir.BfirError):
        bfir.FftPlan(3, False, 4)


def test_product_does_not_reference_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "foo-dsp-bfir_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.replace("Nothing here imports oracle/", ""), f
