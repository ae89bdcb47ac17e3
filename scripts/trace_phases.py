"""Phase timing of k_fwd / k_inv / k_mac_lds from a -DBFIR_TRACE build (scripts/gpu_trace.sh).
Thread 0 of each workgroup stamps the 100 MHz wall clock at phase boundaries; this prints the
mean/median time per phase and the launch-level timeline (first start, last end, concurrency)."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import foo_dsp_bfir_amd as bf
from foo_dsp_bfir_amd import _lib

SLOTS = 24
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = _lib.load()
lib.bfir_debug_read_trace.restype = ctypes.c_int
lib.bfir_debug_read_trace.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
lib.bfir_debug_read_trace_pair.restype = ctypes.c_int
lib.bfir_debug_read_trace_pair.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
L, B, C = 4096, 32, 8
eng = bf.Brutefir(L, B, 4, C)
rng = np.random.default_rng(0)
eng.set_coeff([rng.standard_normal(L * B).astype(np.float32) * 0.01 for _ in range(C)])
eng.set_chunk(chunk)
n = chunk * 3
x = torch.randn(n * L, C, device="cuda", dtype=torch.float32)
y = torch.empty_like(x)
for _ in range(3):
    eng.run_device(x.data_ptr(), y.data_ptr(), n)
    eng.sync()
torch.cuda.synchronize()

names = {0: ("k_fwd_pair", {0: "start", 1: "loaded", 2: "pass0", 3: "xchg0", 4: "pass1", 5: "xchg1", 6: "pass2", 7: "xchg2", 8: "pass3", 10: "split+st"}),
         1: ("k_inv_pair", {0: "start", 9: "lds-in", 1: "build-z", 2: "pass0", 3: "xchg0", 4: "pass1", 5: "xchg1", 6: "pass2", 7: "xchg2", 8: "pass3", 10: "run-end", 11: "stored"}),
         2: ("k_mac_stream", {0: "start", 1: "h+queue", 2: "head", 3: "full", 4: "tail"})}
for kern, (nm, ph) in names.items():
    nw = 4096
    nw = min(nw, 4096)
    buf = np.zeros(nw * SLOTS, dtype=np.uint64)
    rc = (lib.bfir_debug_read_trace if kern == 2 else lib.bfir_debug_read_trace_pair)(kern, buf.ctypes.data, nw)
    assert rc == 0, rc
    t = buf.reshape(nw, SLOTS).astype(np.int64)
    order = list(ph.keys())
    t = t[(t[:, order[0]] > 0) & (t[:, order[-1]] >= t[:, order[0]])]
    t = t[t[:, order[0]] >= t[:, order[0]].max() - 100000]   # the last launch only (1 ms window)
    nw = len(t)
    if nw == 0:
        print('==', nm, ': no stamps'); continue
    order = list(ph.keys())
    t0 = t[:, order[0]].min()
    print(f"== {nm}: {nw} workgroups; launch span {(t[:, order[-1]].max() - t0) / 100:.1f} us; "
          f"wg lifetime mean {(t[:, order[-1]] - t[:, order[0]]).mean() / 100:.2f} us")
    for a, b in zip(order[:-1], order[1:]):
        d = (t[:, b] - t[:, a]) / 100.0
        print(f"   {ph[a]:>9} -> {ph[b]:<9} mean {d.mean():7.2f} us  median {np.median(d):7.2f}  p90 {np.percentile(d, 90):7.2f}")
    st = np.sort(t[:, order[0]] - t0) / 100.0
    en = np.sort(t[:, order[-1]] - t0) / 100.0
    print("   start times (us) pct 0/25/50/75/100:", [round(float(np.percentile(st, p)), 1) for p in (0, 25, 50, 75, 100)])
    print("   end   times (us) pct 0/25/50/75/100:", [round(float(np.percentile(en, p)), 1) for p in (0, 25, 50, 75, 100)])
