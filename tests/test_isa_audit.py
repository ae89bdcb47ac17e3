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
