"""k_mac_sys (csrc/mac_sys.hip): the forward-walking, two-lanes-per-bin form of the partition sums.

Every output is ONE fused multiply-add chain over the partitions in the reference's order (brutefir/brutefir.cpp:
288-299, fftw_convolver.cpp:1464-1525), exactly the chain of k_mac_stream -- so the two kernels must agree BIT FOR
BIT on every block, whatever the launch geometry: run lengths that are no multiple of the slot group, runs shorter
than the filter, the ring wrap inside a run (the two lane halves pass it 2 PL / 2 + 1 slots apart), partition counts
below 2 PL (zero partitions), ragged last partitions, several engines, call-to-call continuation.  The streaming
kernel itself is checked against the oracle throughout the suite; one shape here goes to the oracle directly."""
import numpy as np
import pytest

from test_launch_geometry_gpu import _check, _run_calls, _synth

pytestmark = pytest.mark.gpu

# name, L, B, C, n_eng, blocks resident, calls, chunk, extra env
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
]


@pytest.mark.parametrize("name,L,B,C,n_eng,nb,calls,chunk,env", SHAPES, ids=[g[0] for g in SHAPES])
def test_systolic_mac_is_bit_identical_to_the_streaming_mac(orc, bfir, name, L, B, C, n_eng, nb, calls, chunk, env):
    import torch
    s = 4
    hs = _synth(orc, s, C, B * L - 37, n_eng, seed=len(name) + L)
    rng = np.random.default_rng(B + C)
    x_host = rng.random((n_eng, nb * L, C), dtype=np.float32)
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
    if name == "headline_wrap":
        worst, n_pts = _check(orc, torch, L, B, s, C, hs, x_host, calls, got, chunk, lambda tc: 0)
        print("oracle: %d sampled blocks, worst rel err %.3g" % (n_pts, worst))
