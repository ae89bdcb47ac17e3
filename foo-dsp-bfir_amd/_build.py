"""Build libbfir_hip.so (the C-ABI library) for gfx950 with hipcc, in-tree."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libbfir_hip.so")
SOURCES = ["kernels.hip", "engine.hip", "stage.hip", "bigfft.hip", "pair.hip", "dither.hip", "mac_sys.hip"]
HEADERS = ["kernels.h", "fft_lds.h", os.path.join("..", "..", "include", "bfir_hip.h")]
# -ffp-contract=on: fuse only inside one source expression (the stage kernels rely on it);
# -fno-slp-vectorize: packing the FFT butterflies into v_pk_* costs more moves than it saves
# (the MAC kernel asks for v_pk_fma_f32 explicitly).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-fno-slp-vectorize"]
FLAGS += os.environ.get("BFIR_EXTRA_FLAGS", "").split()   # tuning aid (A/B builds)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


FLAGS_FILE = os.path.join(LIBDIR, "flags.txt")


def _flags_changed():
    """The flags the objects in lib/ were built with are recorded beside them: a library built with
    other flags (a -DBFIR_TRACE tuning build, an A/B variant) is stale for this build."""
    try:
        return open(FLAGS_FILE).read() != " ".join(FLAGS)
    except OSError:
        return True


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link the shared library.

    hipcc cross-compiles without a GPU.  Returns the library path."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(LIBDIR, exist_ok=True)
    force = force or _flags_changed()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            jobs.append(cmd)
        objs.append(o)
    if jobs:   # the translation units are independent: compile them side by side (4 at most)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as pool:
            for r in pool.map(lambda c: subprocess.run(c, check=True), jobs):
                pass
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    with open(FLAGS_FILE, "w") as f:
        f.write(" ".join(FLAGS))
    return LIB


def build_variant(name, extra_flags, verbose=False, only=None):
    """Tuning aid: a second build of the library with extra hipcc flags (-D experiment switches) into
    lib/variant_<name>/ -> lib/libbfir_hip_<name>.so, leaving the product library alone.  Select it for
    one process with BFIR_LIB_OVERRIDE (A/B timing of two builds inside one gpurun call).
    only: recompile just these sources with the extra flags and link the product build's objects for the rest."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    vdir = os.path.join(LIBDIR, "variant_" + name)
    os.makedirs(vdir, exist_ok=True)
    flags = [f for f in FLAGS] + list(extra_flags)
    objs, jobs = [], []
    for src in SOURCES:
        if only is not None and src not in only:
            objs.append(os.path.join(LIBDIR, src.replace(".hip", ".o")))
            continue
        o = os.path.join(vdir, src.replace(".hip", ".o"))
        jobs.append([hipcc] + flags + ["-c", os.path.join(CSRC, src), "-o", o])
        objs.append(o)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=4) as pool:
        for r in pool.map(lambda c: subprocess.run(c, check=True), jobs):
            pass
    out = os.path.join(LIBDIR, "libbfir_hip_%s.so" % name)
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-o", out] + objs, check=True)
    return out


if __name__ == "__main__":
    print(build(verbose=True))
