#!/usr/bin/env python3
"""Machine check of the hand-counted vmcnt scheme of the persistent pair kernels (pair.hip).

k_fwd_pair_ps / k_inv_pair_ps (and their pairs-in-time siblings k_fwd_tp_ps / k_inv_tp_ps) issue the next block's buffer loads from an inline-asm statement the compiler's
s_waitcnt bookkeeping cannot see, and wait for them with a hand-written `s_waitcnt vmcnt(N)`, N = the vector-memory
stores issued after the prefetch.  That is only right while, in the code hipcc actually emits,

  1. the kernel has no scratch: vgpr/sgpr spill counts 0, private segment 0 (a spill is a VMEM op the count misses);
  2. on EVERY control-flow path from the prefetch statement to the counted in-loop wait there are exactly N
     vector-memory instructions, all of them stores, and N equals the wait's immediate;
  3. no instruction on any path from the prefetch to the wait that covers it (the counted one in the loop, the
     vmcnt(0) one behind it) reads or writes a prefetch destination register -- no copy, no reuse;
  4. there is exactly one prefetch statement, one counted wait and one closing vmcnt(0) wait per kernel.

hipcc cross-compiles without a GPU, so this runs in the CPU test suite (tests/test_isa_audit.py) and fails the build
when a compiler or flag change breaks an assumption.  BFIR_PAIR_PERSIST=0 is the fallback at run time.

    python scripts/audit_ps_isa.py [pair.s]      (without an argument: compiles csrc/pair.hip to assembly first)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VMEM = re.compile(r"^(buffer|global|flat|scratch)_(load|store|atomic)")
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def compile_to_asm(out_path):
    sys.path.insert(0, ROOT)
    from importlib import import_module
    b = import_module("foo_dsp_bfir_amd._build")
    src = os.path.join(b.CSRC, "pair.hip")
    cmd = ["hipcc"] + b.FLAGS + ["-S", "--cuda-device-only", "-o", out_path, src]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out_path


def vregs(text):
    s = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            s.add(int(m.group(1)))
        else:
            s.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return s


class Kernel:
    def __init__(self, name, lines):
        self.name = name
        self.ins = []          # (text, in_asm)
        self.labels = {}
        in_asm = False
        for ln in lines:
            t = ln.split(";")[0].strip() if not ln.strip().startswith(";;#") else ln.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t.startswith("."):
                m = re.match(r"^(\.LBB\d+_\d+):", t)
                if m:
                    self.labels[m.group(1)] = len(self.ins)
                continue
            self.ins.append((t, in_asm))

    def succ(self, i):
        t = self.ins[i][0]
        if t.startswith("s_endpgm"):
            return []
        m = re.match(r"^s_branch\s+(\S+)", t)
        if m:
            return [self.labels[m.group(1)]]
        m = re.match(r"^s_cbranch_\w+\s+(\S+)", t)
        if m:
            return [self.labels[m.group(1)], i + 1]
        return [i + 1] if i + 1 < len(self.ins) else []


def audit_kernel(k, meta):
    errs = []
    for key in (".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size"):
        if meta.get(key, None) != 0:
            errs.append("%s = %r (must be 0)" % (key, meta.get(key)))
    pre = [i for i, (t, a) in enumerate(k.ins) if a and t.startswith("buffer_load")]
    waits = [(i, int(re.search(r"vmcnt\((\d+)\)", t).group(1))) for i, (t, a) in enumerate(k.ins)
             if a and t.startswith("s_waitcnt vmcnt")]
    if not pre:
        return ["no inline-asm prefetch found"]
    # one statement = consecutive asm loads
    if pre[-1] - pre[0] != len(pre) - 1 or len(pre) not in (8, 16):
        errs.append("expected ONE prefetch statement of 8 (or, pairs in time, 16) loads, found asm loads at %s" % pre)
    counted = [w for w in waits if w[1] > 0]
    closing = [w for w in waits if w[1] == 0]
    if len(counted) != 1 or len(closing) != 1:
        errs.append("expected one counted and one closing asm wait, found %s" % waits)
        return errs
    dests = set()
    for i in pre:
        dests |= vregs(k.ins[i][0].split(",")[0])
    wait_at = {w[0]: w[1] for w in waits}
    # every path from behind the prefetch to an asm wait
    start = pre[-1] + 1
    seen, stack, ends = set(), [(start, 0, 0)], {}
    while stack:
        i, n_vmem, n_store = stack.pop()
        if (i, n_vmem) in seen or n_vmem > 64:
            continue
        seen.add((i, n_vmem))
        t, in_asm = k.ins[i]
        if i in wait_at:
            ends.setdefault(i, set()).add((n_vmem, n_store))
            continue
        if in_asm and t.startswith("buffer_load"):
            errs.append("a path reaches the prefetch again without passing a wait (instruction %d)" % i)
            continue
        if not in_asm and not t.startswith("s_"):
            hit = vregs(t) & dests
            if hit:
                errs.append("prefetch destination v%s touched before its wait: `%s`" % (sorted(hit), t))
        if VMEM.match(t):
            n_vmem += 1
            n_store += 1 if "_store" in t else 0
        nxt = k.succ(i)
        if not nxt:
            errs.append("a path from the prefetch ends the program without a wait (`%s`)" % t)
        for j in nxt:
            stack.append((j, n_vmem, n_store))
    ci, cn = counted[0]
    if ci not in ends:
        errs.append("no path from the prefetch reaches the counted wait")
    else:
        for n_vmem, n_store in ends[ci]:
            if n_vmem != cn or n_store != n_vmem:
                errs.append("path to `s_waitcnt vmcnt(%d)` carries %d vector-memory ops, %d of them stores"
                            % (cn, n_vmem, n_store))
    if closing[0][0] not in ends:
        errs.append("no path from the prefetch reaches the closing vmcnt(0) wait")
    return errs


def parse(asm_path):
    text = open(asm_path).read().splitlines()
    kernels, cur, name = {}, None, None
    for ln in text:
        m = re.match(r"^(_ZN4bfir\S*k_(?:fwd|inv)_(?:pair|tp)_psILi\d+E\S*):", ln)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if ln.startswith("\t.section") or ln.startswith(".Lfunc_end"):
                kernels[name] = Kernel(name, cur)
                cur = None
            else:
                cur.append(ln)
    meta, kname = {}, None
    for ln in text:
        m = re.match(r"^\s+(?:- )?\.name:\s+(\S+)", ln)
        if m:
            kname = m.group(1)
            meta.setdefault(kname, {})
        m = re.match(r"^\s+(?:- )?(\.\w+):\s+(\d+)\s*$", ln)
        if m and kname:
            meta[kname][m.group(1)] = int(m.group(2))
    return kernels, meta


def audit(asm_path):
    kernels, meta = parse(asm_path)
    report = {}
    for name, k in sorted(kernels.items()):
        short = re.search(r"k_(fwd|inv)_(pair|tp)_psILi(\d+)E", name)
        report["k_%s_%s_ps<%s>" % short.groups()] = audit_kernel(k, meta.get(name, {}))
    return report


def main():
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        path = compile_to_asm(os.path.join(tempfile.mkdtemp(prefix="bfir_isa_"), "pair.s"))
    rep = audit(path)
    bad = 0
    for k, errs in rep.items():
        print("%-24s %s" % (k, "ok" if not errs else "FAIL"))
        for e in errs:
            print("    " + e)
            bad += 1
    if not rep:
        print("no persistent pair kernels found in", path)
        return 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
