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
# one write per rank (the two ranks share the launcher's stdout pipe: a line written in pieces can interleave with the other rank's)
os.write(1, ("RESULT " + json.dumps(out) + "\n").encode())
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


EXCHANGE_WORKER = r'''
import json, os, sys, time
sys.path.insert(0, %(root)r)
import torch
from opencv_contrib_amd import parallel
dist, rank, world, local = parallel.init_distributed("gloo")
B, H, W = 3, 4, 5
parts = [torch.arange(B * 2 * H * W, dtype=torch.float32).reshape(B, 2, H, W) + 1000 * r for r in range(world)] if rank == 0 else None
local_in = [torch.zeros(B, 2, H, W) for _ in range(2)]
local_out = [torch.zeros(B, H, W, 2) for _ in range(2)]
root_out = [[torch.zeros(B, H, W, 2) for _ in range(world)] for _ in range(2)] if rank == 0 else None
seen = []
def compute(inp, out):                      # stand-in for calc_batch: a per-rank function of both frames
    seen.append(float(inp[0, 0, 0, 0]))
    time.sleep(0.005 * (rank + 1))
    out[..., 0] = inp[:, 0] * 2 + rank
    out[..., 1] = inp[:, 1] - inp[:, 0]
el = parallel.run_exchange_pipeline(dist, rank, world, parts, local_in, local_out, root_out, compute, steps=5)
wall = parallel.max_over_ranks(dist, el)
ok = None
if rank == 0:
    ok = True
    for b in range(2):
        for r in range(world):
            exp0 = parts[r][:, 0] * 2 + r
            exp1 = parts[r][:, 1] - parts[r][:, 0]
            ok = ok and bool(torch.equal(root_out[b][r][..., 0], exp0)) and bool(torch.equal(root_out[b][r][..., 1], exp1))
os.write(1, ("RESULT " + json.dumps({"rank": rank, "wall": wall, "rounds": len(seen), "first": seen[0], "ok": ok}) + "\n").encode())   # one write per rank
dist.destroy_process_group()
'''


def test_two_rank_scatter_compute_gather_pipeline(tmp_path):
    """The exchange of the batched-frames mode (inputs on rank 0, flows back to rank 0) with a stand-in compute: every rank gets
    ITS part in every round, both output buffers of the double-buffered pipeline arrive intact on rank 0."""
    import json
    script = tmp_path / "exchange_worker.py"
    script.write_text(EXCHANGE_WORKER % {"root": ROOT})
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    res = sorted((json.loads(l.split("RESULT ", 1)[1]) for l in r.stdout.splitlines() if "RESULT " in l), key=lambda d: d["rank"])
    assert len(res) == 2 and res[0]["ok"] is True
    assert res[0]["rounds"] == res[1]["rounds"] == 5
    assert res[0]["first"] == 0.0 and res[1]["first"] == 1000.0            # each rank computed on its own part
    assert abs(res[0]["wall"] - res[1]["wall"]) < 1e-9 and res[0]["wall"] >= 0.05


def test_exchange_pipeline_single_process():
    import torch
    parts = [torch.arange(12, dtype=torch.float32).reshape(1, 2, 2, 3)]
    lin = [torch.zeros(1, 2, 2, 3) for _ in range(2)]
    lout = [torch.zeros(1, 2, 3, 2) for _ in range(2)]
    rout = [[torch.zeros(1, 2, 3, 2)] for _ in range(2)]

    def compute(inp, out):
        out[..., 0] = inp[:, 0]
        out[..., 1] = inp[:, 1]

    parallel.run_exchange_pipeline(None, 0, 1, parts, lin, lout, rout, compute, steps=3)
    for b in range(2):
        assert torch.equal(rout[b][0][..., 0], parts[0][:, 0]) and torch.equal(rout[b][0][..., 1], parts[0][:, 1])


BENCH_EXCHANGE_WORKER = r'''
import json, os, sys, types
sys.path.insert(0, %(root)r)
sys.argv = ["bench.py"]
import torch
import bench
from opencv_contrib_amd import parallel, cuda

class FakeAlg:                               # stand-in for the HIP TV-L1 object: deterministic function of the two frames
    def calc_batch(self, a, b, out):
        out[..., 0] = a * 2 + 1
        out[..., 1] = b - a

cuda.OpticalFlowDual_TVL1 = types.SimpleNamespace(create=lambda **kw: FakeAlg())
torch.cuda.synchronize = lambda: None
dist, rank, world, local = parallel.init_distributed("gloo")
args = types.SimpleNamespace(iterations=10, epsilon=0.0, exact_math=False, time_block=0, warmup=1, steps=4)
g = torch.Generator().manual_seed(5)          # rank 0 derives one DISTINCT shard per rank from its batch
I0 = torch.rand(3, 6, 8, generator=g) + rank   # other ranks hold different data of their own: it must not matter
I1 = torch.rand(3, 6, 8, generator=g) + rank
ref = torch.empty(3, 6, 8, 2)
if rank == 0:
    FakeAlg().calc_batch(I0, I1, ref)
res = bench.bench_exchange(args, parallel, dist, rank, world, "cpu", I0, I1, ref)
os.write(1, ("RESULT " + json.dumps({"rank": rank, "res": res}) + "\n").encode())   # one write per rank
dist.destroy_process_group()
'''


def test_bench_exchange_leg_two_ranks_with_a_stand_in_algorithm(tmp_path):
    """bench.py's "with_scatter_gather" leg end to end on 2 gloo ranks (the TV-L1 object replaced by a stand-in): rank 0 sends a
    distinct shard to every rank, the flows gathered from rank r equal rank 0's own computation of shard r, one throughput figure
    for the job."""
    import json
    script = tmp_path / "bench_exchange_worker.py"
    script.write_text(BENCH_EXCHANGE_WORKER % {"root": ROOT})
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    res = sorted((json.loads(l.split("RESULT ", 1)[1]) for l in r.stdout.splitlines() if "RESULT " in l), key=lambda d: d["rank"])
    assert len(res) == 2
    assert res[0]["res"]["gathered_flows_identical"] is True and "gathered_flows_identical" not in res[1]["res"]
    assert res[0]["res"]["value"] == res[1]["res"]["value"] > 0
    # the leg tiles the 3-pair batch up to configs[4]'s 64 distinct pairs per GPU (MIFLOW_BENCH_EXCHANGE_PAIRS)
    n = res[0]["res"]["pairs_per_gpu"]
    assert n == 64
    assert abs(res[0]["res"]["exchange_GB_per_step"] - (n * 2 * 6 * 8 + n * 6 * 8 * 2) * 4 / 1e9) < 1e-15


def test_multi_device_state_machine_on_a_fake_device_table(tmp_path):
    """tests/cpp/multi_sm_test.cpp: the worker state machine of mi_tvl1_multi_* (csrc/tvl1_multi_sm.h, the source the HIP backend
    instantiates) against a recording fake with DISTINCT device ids {5, 3, 6}: per-thread device currency of every object, peer
    access enabled both ways, randomised lazy stream scheduling with real byte movement (a missing event wait corrupts a flow),
    sharding, up-front validation, a failing worker (error names the device, both streams drained, machine stays usable),
    construction failures, release of everything."""
    exe = str(tmp_path / "multi_sm_test")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-pthread", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "opencv_contrib_amd", "csrc"), os.path.join(ROOT, "tests", "cpp", "multi_sm_test.cpp"),
                        "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "multi_sm_test: ok" in r.stdout, r.stderr


def _bench(argv, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)


def test_bench_gpus_flag_is_the_rank_count_without_an_external_launcher():
    """`python bench.py --gpus 2`, started bare, must BE a 2-rank job (it re-launches itself under
    torch.distributed.run) -- here over gloo and without touching a GPU (--dry-run); the one JSON line names both ranks."""
    import json
    r = _bench(["--gpus", "2", "--dry-run"], {"MIFLOW_BENCH_BACKEND": "gloo", "MIFLOW_BENCH_DEVICE": "0"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout            # only rank 0 owns stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["gpus_arg"] == 2
    assert out["rccl_ranks"]["world_size"] == 2 and [d["rank"] for d in out["rccl_ranks"]["ranks"]] == [0, 1]
    assert len({d["pid"] for d in out["rccl_ranks"]["ranks"]}) == 2


def test_bench_refuses_a_rank_count_that_is_not_what_gpus_asked_for():
    """Under a launcher the flag and WORLD_SIZE must agree: a 1-rank group with --gpus 2 fails loudly instead of printing n_gpus 1."""
    r = _bench(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 2" in r.stderr and not any(l.startswith("{") for l in r.stdout.splitlines())
    r = _bench(["--gpus", "0", "--dry-run"])
    assert r.returncode != 0
    r = _bench(["--gpus", "2", "--workload", "stereobm"])
    assert r.returncode != 0 and "single-GPU" in r.stderr
