"""k_mac_sys (csrc/mac_sys.hip): the forward-walking form of the partition sums, S adjacent lanes per bin.

Every output is ONE fused multiply-add chain over the partitions in the reference's order (brutefir/brutefir.cpp:
288-299, fftw_convolver.cpp:1464-1525 / 2160-2220), exactly the chain of the default MAC kernels (k_mac_stream,
k_mac_lds on the pair layout, the fp64 LDS kernel) -- so it must agree with them BIT FOR BIT on every block, whatever
the launch geometry: run lengths that are no multiple of the slot group, runs shorter than the filter, the ring wrap
inside a run (the stages pass it PL + 1 slots apart), partition counts below S PL (zero partitions), ragged last
partitions, several engines, call-to-call continuation, two / four / eight / sixteen lanes per bin, fp32 (pairs layout) and fp64
(the reference's grouped layout).  The default kernels are checked against the oracle throughout the suite; two shapes
here go to the oracle directly."""
import numpy as np
import pytest

from test_launch_geometry_gpu import _check, _run_calls, _synth

pytestmark = pytest.mark.gpu

# name, L, B, C, n_eng, blocks resident, calls, chunk, extra env  (a leading "d_" = realsize 8)
SHAPES = [
    ("headline_wrap", 4096, 32, 8, 1, 700, [700, 37, 700, 1, 5], 700, {}),          # ring 1432 wraps in the third call
    ("headline_short_runs", 4096, 32, 8, 1, 300, [300, 300], 300, {"BFIR_MAC_RANGE": "33"}),
    ("runs_of_40", 1024, 32, 2, 1, 500, [500, 123, 500], 250, {"BFIR_MAC_RANGE": "40"}),
    ("runs_shorter_than_filter", 1024, 32, 2, 1, 200, [200, 200], 200, {"BFIR_MAC_RANGE": "7"}),
    ("B20_zero_partitions", 1024, 20, 4, 1, 400, [400, 400, 9], 400, {}),
    ("B17", 512, 17, 2, 3, 300, [300, 300], 300, {}),
    ("B9_PL8", 2048, 9, 2, 1, 300, [300, 77, 300], 150, {}),
    ("B16_PL8", 512, 16, 6, 1, 300, [300, 300], 300, {"BFIR_MAC_RANGE": "50"}),
    ("B3_PL4", 1024, 3, 2, 2, 200, [200, 200, 6], 100, {}),
    ("B8_PL4_L256", 256, 8, 2, 1, 500, [500, 500], 500, {"BFIR_MAC_RANGE": "21"}),
    # four and eight lanes per bin (fp32, more than 32 partitions: the default there is the LDS-shared kernel)
    ("B64_S4", 1024, 64, 2, 1, 400, [400, 77, 400], 400, {}),
    ("B40_S4_short_runs", 512, 40, 4, 1, 300, [300, 300], 150, {"BFIR_MAC_RANGE": "25"}),
    ("B100_S8", 1024, 100, 2, 1, 300, [300, 300, 5], 300, {}),
    # fp64, grouped layout: the plug-in's shipped precision (REALSIZE 8, common.h:17), cfg5's partition count
    ("d_B64_S4_plugin", 1024, 64, 2, 1, 300, [300, 40, 300], 300, {}),
    ("d_B64_S4_L4096_wrap", 4096, 64, 2, 1, 200, [200, 200, 200], 100, {"BFIR_MAC_RANGE": "37"}),
    ("d_B20_S2", 512, 20, 3, 2, 200, [200, 200], 200, {}),
    ("d_B9_S2_PL8", 256, 9, 2, 1, 300, [300, 300], 300, {"BFIR_MAC_RANGE": "11"}),
    ("d_B100_S8", 512, 100, 2, 1, 260, [260, 33, 260], 260, {}),
    ("d_B128_S8_wrap", 1024, 128, 1, 2, 300, [300, 300], 150, {"BFIR_MAC_RANGE": "45"}),
    # sixteen lanes per bin (a whole DPP row): 129 ... 256 partitions
    ("d_B129_S16", 1024, 129, 2, 1, 300, [300, 41, 300], 300, {}),
    ("d_B256_S16_wrap", 512, 256, 1, 2, 400, [400, 400], 200, {"BFIR_MAC_RANGE": "45"}),
    ("d_B200_S16_runs_shorter_than_filter", 1024, 200, 3, 1, 260, [260, 260, 3], 260, {"BFIR_MAC_RANGE": "19"}),
    ("B160_S16", 1024, 160, 2, 1, 300, [300, 300], 300, {}),
    # twelve partitions per stage: the quarter steps between the powers of two (24 / 48 / 96 / 192)
    ("d_B23_S2_PL12", 1024, 23, 2, 1, 300, [300, 300], 150, {"BFIR_MAC_RANGE": "29"}),
    ("d_B44_S4_PL12", 1024, 44, 2, 1, 300, [300, 55, 300], 300, {}),
    ("d_B90_S8_PL12_wrap", 512, 90, 3, 1, 300, [300, 300], 100, {"BFIR_MAC_RANGE": "31"}),
    ("d_B188_S16_PL12", 1024, 188, 2, 1, 300, [300, 300, 2], 300, {}),
    ("d_B192_S16_PL12_runs_shorter_than_filter", 512, 192, 1, 2, 260, [260, 260], 260, {"BFIR_MAC_RANGE": "17"}),
    ("B47_S4_PL12", 1024, 47, 2, 1, 400, [400, 77, 400], 200, {}),
]


@pytest.mark.parametrize("name,L,B,C,n_eng,nb,calls,chunk,env", SHAPES, ids=[g[0] for g in SHAPES])
def test_systolic_mac_is_bit_identical_to_the_streaming_mac(orc, bfir, name, L, B, C, n_eng, nb, calls, chunk, env):
    import torch
    s = 8 if name.startswith("d_") else 4
    dt = np.float32 if s == 4 else np.float64
    hs = _synth(orc, s, C, B * L - 37, n_eng, seed=len(name) + L)
    rng = np.random.default_rng(B + C)
    x_host = rng.random((n_eng, nb * L, C), dtype=dt)
    x_host *= 2.0; x_host -= 1.0
    d_in = torch.from_numpy(x_host).cuda()
    stream_env = {k: v for k, v in env.items() if k != "BFIR_MAC_RANGE"}
    ref = _run_calls(bfir, torch, L, B, s, C, hs, d_in, calls, chunk, dict(stream_env, BFIR_MAC_SYS="0"))
    got = _run_calls(bfir, torch, L, B, s, C, hs, d_in, calls, chunk, dict(env, BFIR_MAC_SYS="1"))
    for ci, (a, b) in enumerate(zip(got, ref)):
        if not torch.equal(a, b):
            diff = (a != b).any(dim=2).view(n_eng, -1, L).any(dim=2)           # [engine, block]
            bad = diff.nonzero()[:8].tolist()
            raise AssertionError("call %d: %d blocks differ, first (engine, block): %s" % (ci, int(diff.sum()), bad))
    if name in ("headline_wrap", "d_B64_S4_plugin"):
        worst, n_pts = _check(orc, torch, L, B, s, C, hs, x_host, calls, got, chunk, lambda tc: 0)
        print("oracle: %d sampled blocks, worst rel err %.3g" % (n_pts, worst))
