"""bench.py --gpus N started WITHOUT a launcher spawns its own ranks (VERDICT r01 "What's missing" 2).

CPU part: the launcher plumbing alone (fresh child processes, gloo rendezvous on 127.0.0.1, the
MAX / SUM timing reductions, rank 0's single JSON line) via the hidden --spawn-check switch.
GPU part: two real ranks rehearsed on ONE GPU (gloo + --device 0), both sharding modes, the HIP
engine doing the work in each rank."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 3])
def test_gpus_flag_spawns_ranks_cpu(n):
    d = _bench("--gpus", str(n), "--spawn-check")
    assert d["spawn_check"] and d["n_gpus"] == n
    assert d["max_elapsed"] == float(n)            # rank r reported 1 + r
    assert d["sum_units"] == 10.0 * n
    assert d["channels_covered"] == 8.0            # shard_range(8, r, n) covers the 8 channels exactly once


def test_eight_ranks_rendezvous_and_shares_cpu():
    """The 8-GPU point of north_star's curve, rehearsed on the CPU (gloo): eight rank processes meet, the
    channel-sharded job gives every rank exactly ONE channel, configs[3]'s 256 streams come to 32 per rank."""
    d = _bench("--gpus", "8", "--spawn-check")
    assert d["n_gpus"] == 8 and d["max_elapsed"] == 8.0 and d["sum_units"] == 80.0
    assert d["channels_covered"] == 8.0 and d["channels_rank0"] == 1 and d["channels_max"] == 1.0
    assert d["streams_covered"] == 256.0 and d["streams_rank0"] == 32 and d["streams_max"] == 32.0


@pytest.mark.gpu
@pytest.mark.parametrize("shard,scaling,ch0,units", [("replicas", "weak", 8, 6), ("channels", "strong", 2, 1)])
def test_six_ranks_on_one_gpu(shard, scaling, ch0, units):
    """As many ranks as a GPU box admits on one card (six), 64-block steps so that six engines' buffers fit one
    device: every rank runs the HIP engine on its share (replicas: six independent 8-channel streams; channels:
    ONE stream, shares of 2, 2, 1, 1, 1, 1 channels -- the one-channel shares are what the 8-GPU point runs)."""
    d = _bench("--gpus", "6", "--dist-backend", "gloo", "--device", "0", "--shard", shard, "--blocks", "64",
               "--steps", "2", "--warmup", "1", "--no-extras", "--no-exclusive-pass", timeout=900)
    assert d["n_gpus"] == 6 and d["scaling"] == scaling and d["config"]["channels_rank0"] == ch0
    assert d["value"] > 0 and d["parity_rel_err_vs_oracle"] <= 1e-5
    per_step = 64 * 4096 * 8 * units
    assert abs(d["value"] * 1e6 * d["ms_per_step"] * 1e-3 - per_step) <= 1e-3 * per_step


@pytest.mark.gpu
def test_256_streams_shape_dealt_out_to_six_ranks_on_one_gpu():
    """configs[3]'s arrangement (independent stereo engines sharing launches, dealt out to the ranks) at 24 streams."""
    d = _bench("--gpus", "6", "--dist-backend", "gloo", "--device", "0", "--workload", "cfg4_stereo_65536tap_L4096_fp32",
               "--streams", "24", "--blocks", "64", "--steps", "2", "--warmup", "1", "--no-extras", "--no-exclusive-pass",
               timeout=900)
    assert d["n_gpus"] == 6 and d["scaling"] == "strong" and d["config"]["engines_rank0"] == 4
    assert d["config"]["streams_total"] == 24 and d["parity_rel_err_vs_oracle"] <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("shard,scaling,ch0", [("replicas", "weak", 8), ("channels", "strong", 4)])
def test_two_ranks_on_one_gpu(shard, scaling, ch0):
    d = _bench("--gpus", "2", "--dist-backend", "gloo", "--device", "0", "--shard", shard, "--blocks", "256",
               "--steps", "2", "--warmup", "1", "--no-extras")
    assert d["n_gpus"] == 2 and d["scaling"] == scaling
    assert d["config"]["channels_rank0"] == ch0
    assert d["value"] > 0 and d["parity_rel_err_vs_oracle"] <= 1e-5
    # whole-job samples: replicas = 2 streams of 8 channels, channels = 1 stream of 8 channels
    per_step = 256 * 4096 * 8 * (2 if shard == "replicas" else 1)
    assert abs(d["value"] * 1e6 * d["ms_per_step"] * 1e-3 - per_step) <= 1e-3 * per_step


@pytest.mark.gpu
def test_streams_dealt_out_to_two_ranks_on_one_gpu():
    d = _bench("--gpus", "2", "--dist-backend", "gloo", "--device", "0", "--workload",
               "cfg4_stereo_65536tap_L4096_fp32", "--streams", "6", "--blocks", "64", "--steps", "2",
               "--warmup", "1", "--no-extras")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["engines_rank0"] == 3
    assert d["config"]["streams_total"] == 6
