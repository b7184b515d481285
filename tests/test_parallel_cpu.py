"""Multi-GPU batched mode exercised on CPU with gloo, world_size 2: static sharding of independent pairs,
barrier + max-over-ranks timing, gather of per-rank results to rank 0.  No compute kernels are involved (the shards are
independent by construction); the per-item work is a stand-in that records which items each rank processed."""
import os
import socket
import subprocess
import sys

import pytest

from opencv_contrib_amd import parallel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                seen += list(parallel.shard_range(n, world, r))
            assert seen == list(range(n))
    assert list(parallel.shard_range(512, 8, 3)) == list(range(192, 256))   # BASELINE configs[4]: 64 pairs per GPU
    with pytest.raises(ValueError):
        parallel.shard_range(4, 2, 2)


WORKER = r'''
import json, os, sys, time
sys.path.insert(0, %(root)r)
import torch
from opencv_contrib_amd import parallel
dist, rank, world, local = parallel.init_distributed("gloo")
assert world == 2 and dist is not None
done = []
def work(shard):
    done.append(list(shard))
    time.sleep(0.01 * (rank + 1))          # rank 1 is slower: the job time must be ITS time
wall, total = parallel.run_sharded(dist, rank, world, 7, work, warmup=1, steps=3)
res = torch.full((len(parallel.shard_range(7, world, rank)), 2), float(rank))
g = parallel.gather_to_rank0(dist, rank, world, res)
out = {"rank": rank, "wall": wall, "total": total, "calls": len(done), "shard": done[0],
       "gathered": g.tolist() if g is not None else None}
print("RESULT " + json.dumps(out), flush=True)
dist.destroy_process_group()
'''


def test_two_rank_gloo_job(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    res = sorted((json.loads(l.split("RESULT ", 1)[1]) for l in r.stdout.splitlines() if "RESULT " in l), key=lambda d: d["rank"])
    assert len(res) == 2
    assert res[0]["shard"] == [0, 1, 2, 3] and res[1]["shard"] == [4, 5, 6]
    assert res[0]["total"] == res[1]["total"] == 7 and res[0]["calls"] == 4
    assert abs(res[0]["wall"] - res[1]["wall"]) < 1e-9 and res[0]["wall"] >= 0.055      # 3 steps x 20 ms of the slow rank
    assert res[1]["gathered"] is None
    assert res[0]["gathered"] == [[0.0, 0.0]] * 4 + [[1.0, 1.0]] * 3
