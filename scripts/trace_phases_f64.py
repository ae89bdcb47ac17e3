"""Phase timing of the general k_fwd / k_inv of an fp64 engine from a -DBFIR_TRACE build (BFIR_LIB_OVERRIDE = the trace build):
thread 0 of each workgroup stamps the 100 MHz wall clock at phase boundaries (the stamps drain the memory counters, so phases
do not overlap as they do in the product).    python scripts/trace_phases_f64.py L B C frames(f32|f64) [chunk]"""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import foo_dsp_bfir_amd as bf
from foo_dsp_bfir_amd import _lib

SLOTS = 24
L, B, C = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
fr = sys.argv[4]
chunk = int(sys.argv[5]) if len(sys.argv) > 5 else 1024
lib = _lib.load()
lib.bfir_debug_read_trace.restype = ctypes.c_int
lib.bfir_debug_read_trace.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
eng = bf.Brutefir(L, B, 8, C) if fr == "f64" else bf.Brutefir(L, B, 8, C, in_format=bf.SAMPLE_FORMAT_FLOAT_LE, out_format=bf.SAMPLE_FORMAT_FLOAT_LE)
rng = np.random.default_rng(0)
eng.set_coeff([rng.standard_normal(L * B) * 0.01 for _ in range(C)])
eng.set_chunk(chunk)
n = chunk * 3
x = torch.randn(n * L, C, device="cuda", dtype=torch.float64 if fr == "f64" else torch.float32)
y = torch.empty_like(x)
for _ in range(3):
    eng.run_device(x.data_ptr(), y.data_ptr(), n)
    eng.sync()
torch.cuda.synchronize()
names = {0: ("k_fwd", {0: "start", 1: "loaded", 2: "pass0", 4: "pass1", 6: "pass2", 9: "split", 10: "stored"}),
         1: ("k_inv", {0: "start", 9: "lds-in", 1: "build-z", 2: "pass0", 4: "pass1", 6: "pass2", 10: "stored"})}
for kern, (nm, ph) in names.items():
    nw = 4096
    buf = np.zeros(nw * SLOTS, dtype=np.uint64)
    assert lib.bfir_debug_read_trace(kern, buf.ctypes.data, nw) == 0
    t = buf.reshape(nw, SLOTS).astype(np.int64)
    order = [k for k in ph if (t[:, k] > 0).any()]
    t = t[(t[:, order[0]] > 0) & (t[:, order[-1]] >= t[:, order[0]])]
    t = t[t[:, order[0]] >= t[:, order[0]].max() - 200000]
    if len(t) == 0:
        print("==", nm, ": no stamps"); continue
    t0 = t[:, order[0]].min()
    print(f"== {nm} L={L} B={B} C={C} {fr}: {len(t)} workgroups stamped; wg lifetime mean {(t[:, order[-1]] - t[:, order[0]]).mean() / 100:.2f} us")
    for a, b in zip(order[:-1], order[1:]):
        d = (t[:, b] - t[:, a]) / 100.0
        print(f"   {ph[a]:>9} -> {ph[b]:<9} mean {d.mean():7.2f} us  median {np.median(d):7.2f}  p90 {np.percentile(d, 90):7.2f}")
