"""Build libbfir_hip.so (the C-ABI library) for gfx950 with hipcc, in-tree."""
import json
import os
import re
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
USAGE_FILE = os.path.join(LIBDIR, "resource_usage.json")


def _compile(cmd, src):
    """One translation unit, with the compiler's per-kernel resource remarks (-Rpass-analysis=kernel-resource-usage:
    registers, scratch, occupancy, LDS) parsed into lib/<src>.usage.json -- tests/test_isa_audit.py holds the kernels whose
    design rests on a register budget to it (no scratch, the waves per SIMD they were built for)."""
    r = subprocess.run(cmd + ["-Rpass-analysis=kernel-resource-usage"], stderr=subprocess.PIPE, text=True)
    usage, cur = {}, None
    rest = []
    for ln in r.stderr.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", ln)
        if m:
            cur = usage.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass-analysis", ln)
        if m and cur is not None:
            v = m.group(2)
            cur[m.group(1).strip()] = int(v) if v.isdigit() else v
            continue
        if "-Rpass-analysis=kernel-resource-usage" not in ln and not re.match(r"^\s+(\d+ \|)|^\s+\|", ln):
            rest.append(ln)
    if r.returncode != 0:
        raise subprocess.CalledProcessError(r.returncode, cmd, stderr="\n".join(rest))
    with open(os.path.join(LIBDIR, os.path.basename(src) + ".usage.json"), "w") as f:
        json.dump(usage, f, indent=0, sort_keys=True)


def resource_usage():
    """{mangled kernel name: {"VGPRs": n, "ScratchSize": n, "Occupancy": n, ...}} of the objects in lib/ (build() first)."""
    out = {}
    for src in SOURCES:
        try:
            out.update(json.load(open(os.path.join(LIBDIR, src + ".usage.json"))))
        except OSError:
            pass
    return out


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
            jobs.append((cmd, src))
        objs.append(o)
    if jobs:   # the translation units are independent: compile them side by side (4 at most)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as pool:
            try:
                for r in pool.map(lambda c: _compile(*c), jobs):
                    pass
            except subprocess.CalledProcessError as e:
                print(e.stderr or "")
                raise
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
