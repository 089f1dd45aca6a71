"""The N>1 path on CPU: world_size-2 gloo. Independent streams shard across ranks
with no data-path collective; only the bench's barrier + max-over-ranks timing
use torch.distributed. Each rank encodes its own stream with the CPU oracle here
(the sharding / seeding / reduction logic is what is under test, not the kernels)."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import hashlib, json, os, sys, time
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from tests import oracle_lib as ol
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo")
    w, h, streams_per_rank, frames = 64, 48, 2, 4
    hashes = []
    dist.barrier()
    t0 = time.perf_counter()
    for s in range(streams_per_rank):
        gs = rank * streams_per_rank + s            # global stream id -> seed offset gs<<8
        enc = ol.OracleEncoder(ol.default_config(w, h, qp=24, keyint=25))
        hh = hashlib.sha1()
        for f in range(frames):
            hh.update(enc.encode(*ol.gen_frame("B", w, h, f, gs << 8)))
        hashes.append((gs, hh.hexdigest()))
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)       # max-over-ranks timing
    gathered = [None] * world
    dist.all_gather_object(gathered, hashes)        # test-only: collect what every rank made
    if rank == 0:
        print("RESULT " + json.dumps({"max_dt": float(dt.item()), "hashes": gathered}))
    dist.destroy_process_group()
""")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_shard_streams_without_collectives(tmp_path):
    import hashlib
    import json
    from tests import oracle_lib as ol

    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][0]
    res = json.loads(line[7:])
    got = {gs: h for rank_hashes in res["hashes"] for gs, h in rank_hashes}
    assert sorted(got) == [0, 1, 2, 3]            # every global stream encoded exactly once
    for gs in range(4):                            # and equals a stream encoded alone
        enc = ol.OracleEncoder(ol.default_config(64, 48, qp=24, keyint=25))
        hh = hashlib.sha1()
        for f in range(4):
            hh.update(enc.encode(*ol.gen_frame("B", 64, 48, f, gs << 8)))
        assert got[gs] == hh.hexdigest()
    assert len(set(got.values())) == 4            # different seeds -> different streams
    assert res["max_dt"] > 0
