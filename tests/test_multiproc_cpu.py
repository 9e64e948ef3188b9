"""N > 1 path on CPU: world_size-2 gloo job launched exactly as the driver launches bench.py
(python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 ...).
Checks: contiguous game-id shards, barrier + MAX-over-ranks timing, SUM of work, and that sharded results equal
the single-process run of the same global ids (the property that makes 1/2/4/8-GPU runs agree)."""
import ctypes as C
import json
import os
import subprocess
import sys

from alphazeroforhnefatafl_amd import abi, dist as tdist
from alphazeroforhnefatafl_amd.abi import TaflMctsParams, TaflState
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_world_size_2_gloo(tmp_path):
    per_rank = 3
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "tests", "_dist_worker.py"), str(tmp_path), str(per_rank)]
    subprocess.run(cmd, check=True, env=env, cwd=ROOT, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    outs = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    assert [o["base"] for o in outs] == [0, per_rank]
    assert all(o["world"] == 2 for o in outs)
    assert all(o["elapsed_max"] == 2.0 for o in outs)                 # MAX over ranks of (1 + rank)
    assert all(o["sims_total"] == 2 * per_rank * 40 for o in outs)    # SUM over ranks
    merged = {}
    for o in outs:
        merged.update(o["results"])
    # single-process run of the same global ids
    lg = orc.GameLogic(abi.rules.BRANDUBH, 7)
    st = orc.GameState(abi.boards.BRANDUBH, abi.ATTACKER, 64).to_abi()
    n = 2 * per_rank
    states = (TaflState * n)(*[st] * n)
    kids, cnt, _ = orc.batch_mcts(lg, states, n, 64, TaflMctsParams(40, 64, 1.0, 9, 0, 0), 0, 64)
    for g in range(n):
        want = [[kids[g * 64 + j].action, kids[g * 64 + j].visits, kids[g * 64 + j].q.hex()] for j in range(cnt[g])]
        assert merged[str(g)] == want, g
    # different ids give different trees (the shards are not trivially identical)
    assert merged["0"] != merged[str(per_rank)]


def test_single_process_helpers():
    assert tdist.shard_base(3, 65536) == 196608
    assert tdist.max_over_ranks(1.5, 1) == 1.5 and tdist.sum_over_ranks(2.0, 1) == 2.0


def _bench(args, env_extra=None, timeout=600):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TAFL_BENCH_CHILD"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run(args, env=env, cwd=ROOT, timeout=timeout, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def test_bench_launches_its_own_ranks_dry_run():
    """`python bench.py --gpus N` invoked directly (the way the driver invokes N=1) starts N fresh rank processes itself: rendezvous
    over gloo, contiguous game-id shards (configs[3]: rank r owns ids r*65536 ..), MAX-over-ranks reduce.  No GPU needed for --dry-run."""
    r = _bench([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"])
    assert r.returncode == 0, r.stderr
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 4 and line["shard_bases"] == [0, 65536, 131072, 196608] and line["devices"] == [0, 1, 2, 3]
    assert line["max_over_ranks"] == 4.0 and line["games_total"] == 4 * 65536 and line["launcher"] == "bench.py child processes"
    # rehearsal on one GPU: every rank on device 0
    r = _bench([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--single-device", "--games", "1000"])
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["devices"] == [0, 0] and line["shard_bases"] == [0, 1000]


def test_bench_under_torchrun_dry_run_and_mismatch():
    """The driver's launch line (torch.distributed.run, one rank per GPU) keeps working, and --gpus != WORLD_SIZE is an error."""
    r = _bench([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29537", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"])
    assert r.returncode == 0, r.stderr
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["launcher"] == "torch.distributed.run" and line["shard_bases"] == [0, 65536]
    r = _bench([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], {"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "does not match WORLD_SIZE" in r.stderr


def test_bench_launcher_propagates_rank_failure():
    """Without a GPU every rank of a real run fails loudly (there is no CPU path) and the launcher reports it."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-only check")
    r = _bench([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-device", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs an MI355X") == 2
