"""The hand-counted `s_waitcnt vmcnt(N)` of the persistent pair kernels, checked in the emitted ISA (CPU; hipcc
cross-compiles without a GPU).  See scripts/audit_ps_isa.py for what is asserted and why (the scheme is
only right while there is no spill, exactly N stores follow the asm prefetch on every path to its wait, and nothing
touches the prefetch registers in between).  The auditor is first shown to catch each kind of violation on a doctored
listing, then run on the code the product is built from, with the product's own flags."""
import importlib.util
import os
import re
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("audit_ps_isa", os.path.join(ROOT, "scripts", "audit_ps_isa.py"))
audit_ps_isa = importlib.util.module_from_spec(spec)
spec.loader.exec_module(audit_ps_isa)


@pytest.fixture(scope="module")
def listing(tmp_path_factory):
    if not shutil.which("hipcc"):
        pytest.skip("no hipcc on this machine")
    return audit_ps_isa.compile_to_asm(str(tmp_path_factory.mktemp("isa") / "pair.s"))


def test_persistent_pair_kernels_pass_the_audit(listing):
    rep = audit_ps_isa.audit(listing)
    assert len(rep) == 20, sorted(rep)                       # forward and inverse, channel pairs and time pairs, N = 2^10 .. 2^14
    assert all(not errs for errs in rep.values()), {k: v for k, v in rep.items() if v}


def test_auditor_catches_violations(listing, tmp_path):
    text = open(listing).read()
    name = re.search(r"^(_ZN4bfir\S*k_fwd_pair_psILi13E\S*):", text, re.M).group(1)
    a = text.index(name + ":")
    b = text.index(".Lfunc_end", a)
    body = text[a:b]
    stores = [m.start() for m in re.finditer(r"\tbuffer_store_dwordx4 ", body)]
    assert len(stores) == 8

    def run(mutated_body, meta_edit=None):
        t = text[:a] + mutated_body + text[b:]
        if meta_edit:
            t = meta_edit(t)
        p = tmp_path / "m.s"
        p.write_text(t)
        return audit_ps_isa.audit(str(p))["k_fwd_pair_ps<13>"]

    assert run(body) == []
    # a ninth vector-memory operation inside the counted window (what a spill would be)
    extra = body[:stores[3]] + "\tscratch_store_dword off, v1, off\n" + body[stores[3]:]
    assert any("carries 9 vector-memory ops" in e for e in run(extra))
    # one store fewer
    line_end = body.index("\n", stores[0])
    assert any("carries 7" in e for e in run(body[:stores[0]] + body[line_end + 1:]))
    # a copy out of a prefetch destination register before its wait
    dest = re.search(r";;#ASMSTART\n(?:\ts_nop 4\n)?\tbuffer_load_dwordx2 v\[(\d+):", body).group(1)
    touched = body[:stores[0]] + "\tv_mov_b32_e32 v0, v%s\n" % dest + body[stores[0]:]
    assert any("touched before its wait" in e for e in run(touched))
    # a spill recorded in the kernel's metadata
    def spill(t):
        i = t.index(".name:           " + name)
        j = t.index(".vgpr_spill_count: 0", i)
        return t[:j] + ".vgpr_spill_count: 2" + t[j + len(".vgpr_spill_count: 0"):]
    assert any(".vgpr_spill_count = 2" in e for e in run(body, spill))


def test_register_budgets_of_the_persistent_kernels():
    """Kernels whose design rests on a register budget, held to it by the compiler's own per-kernel report (recorded at build
    time, _build.resource_usage): no scratch (a spill is a vector-memory operation -- in k_fwd_run / k_inv_run a reload waits
    behind the prefetch, in the *_ps kernels it breaks the counted wait) and the waves per SIMD each was shaped for."""
    if not shutil.which("hipcc"):
        pytest.skip("no hipcc on this machine")
    import sys
    sys.path.insert(0, ROOT)
    import importlib
    b = importlib.import_module("foo_dsp_bfir_amd._build")
    b.build()
    u = b.resource_usage()
    if not any("k_fwd_run" in k for k in u):                 # objects from a build that did not record it
        b.build(force=True)
        u = b.resource_usage()

    def pick(pattern):
        got = {k: v for k, v in u.items() if re.search(pattern, k)}
        assert got, pattern
        return got

    for pattern, min_occ, n in ((r"k_(fwd|inv)_run", 2, 28),                      # 1024 ... 8192 points x float / double frames x spectrum layouts (pairs up to 4096)
                                (r"k_(fwd|inv)_(pair|tp)_ps", 3, 20),
                                (r"k_mac_sysIfLb1ELi2ELi16ELi4E", 6, 1), (r"k_mac_sysIfLb1ELi4ELi16ELi4E", 6, 1),
                                (r"k_mac_sysIdLb[01]ELi4ELi16ELi6E", 3, 2), (r"k_mac_sysIdLb[01]ELi2ELi16ELi4E", 3, 2),   # fp64: both spectrum layouts
                                (r"k_mac_sysIdLb[01]ELi8ELi16ELi6E", 3, 2), (r"k_mac_sysIdLb[01]ELi16ELi16ELi6E", 3, 2),
                                (r"k_mac_sysIdLb[01]ELi(2|4|8|16)ELi12ELi4E", 4, 8), (r"k_mac_sysIfLb1ELi4ELi12ELi4E", 6, 1)):   # twelve partitions per stage
        ks = pick(pattern)
        assert n is None or len(ks) == n, (pattern, sorted(ks))
        for name, r in ks.items():
            assert r["ScratchSize"] == 0 and r["VGPRs Spill"] == 0 and r["Dynamic Stack"] == "False", (name, r)
            assert r["Occupancy"] >= min_occ, (name, r)
    # the headline's MAC: three waves per SIMD at 168 registers; its one spilled address pair is reloaded once per 32 output
    # blocks, outside the FMA stream (nothing is counted by hand in that kernel)
    hs = pick(r"k_mac_streamILi32ELi8ELb0E")
    assert len(hs) == 1 and all(r["Occupancy"] >= 3 and r["ScratchSize"] <= 32 for r in hs.values()), hs
