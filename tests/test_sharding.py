"""The N > 1 path on CPU: two gloo ranks split a batch of independent engines with
foo_dsp_bfir_amd.sharding, process their shares with no data exchange, and the union equals
the single-process result.  The per-engine compute here is the CPU oracle (there is no GPU in
this test); the sharding and the timing reduction are the product code under test."""
import os
import socket

import numpy as np
import pytest

from foo_dsp_bfir_amd import sharding


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 256, 1000):
        for world in (1, 2, 3, 8):
            got = [sharding.shard_range(n, r, world) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
            sizes = [hi - lo for lo, hi in got]
            assert max(sizes) - min(sizes) <= 1 and sizes == sharding.shard_sizes(n, world)
    assert sharding.shard_sizes(256, 8) == [32] * 8          # BASELINE configs[3]
    with pytest.raises(ValueError):
        sharding.shard_range(4, 2, 2)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, E, L, B, C, nb, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    lo, hi = sharding.shard_range(E, rank, world)
    out = {}
    for e in range(lo, hi):                      # this rank's engines only; nothing is exchanged
        rng = np.random.default_rng(1000 + e)    # stream e is the same whoever owns it
        h = O.synth_ir(rng, C, B * L, np.float32)
        x = O.synth_audio(rng, nb * L, C, np.float32)
        eng = O.Engine(L, B, 4, C); eng.set_coeff(h)
        out[e] = eng.run(x)[1]
    elapsed = 1.0 + rank                          # rank 1 is "slower"
    worst = sharding.max_over_ranks(elapsed)
    total = sharding.sum_over_ranks(hi - lo)
    dist.barrier()
    q.put((rank, (lo, hi), worst, total, {e: float(np.abs(v).sum()) for e, v in out.items()}))
    dist.destroy_process_group()


def test_two_rank_gloo_shards_cover_all_engines(orc):
    import torch.multiprocessing as mp
    E, L, B, C, nb, world = 5, 64, 3, 2, 6, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, E, L, B, C, nb, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort()
    assert [r[1] for r in results] == [(0, 3), (3, 5)]
    assert all(r[2] == 2.0 for r in results)        # MAX over ranks of the elapsed time
    assert all(r[3] == E for r in results)          # units all ranks processed
    sums = {}
    for r in results:
        sums.update(r[4])
    assert sorted(sums) == list(range(E))
    for e in range(E):                               # same answer as one process doing everything
        rng = np.random.default_rng(1000 + e)
        h = orc.synth_ir(rng, C, B * L, np.float32)
        x = orc.synth_audio(rng, nb * L, C, np.float32)
        eng = orc.Engine(L, B, 4, C); eng.set_coeff(h)
        assert float(np.abs(eng.run(x)[1]).sum()) == sums[e]
