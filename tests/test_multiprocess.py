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


# ---- two encoder PROCESSES on one GPU: only one of them may pipeline its frames on the device ----
TWO_PROC_WORKER = textwrap.dedent("""
    import hashlib, json, os, sys, time
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    sys.path.insert(0, %r)
    import torch  # noqa: F401  (first: one HIP runtime per process)
    import cedarx_h264_encoder_amd as ce
    me, other, want = sys.argv[1], sys.argv[2], json.load(open(sys.argv[3]))
    w, h, n = 1920, 1080, 25
    frames = [ce.gen_pattern("B", w, h, f) for f in range(n)]
    enc = ce.CedarEncoder(width=w, height=h, qp=24, keyframe_interval=25, fps_num=30, fps_den=1)
    open(me, "w").write("configured")
    t0 = time.time()
    while not os.path.exists(other) and time.time() - t0 < 120:   # both sessions exist before either encodes
        time.sleep(0.01)
    ok = os.path.exists(other)
    pipelined = enc.stats()["frames_pipelined"]
    for rep in range(3):
        got, q = [], 0
        for y, uv in frames:
            if q == enc.pipeline_depth:
                enc.collect(); got.append(enc.output(0)); q -= 1
            enc.set_input(y, uv)
            enc.submit(); q += 1
        while q:
            enc.collect(); got.append(enc.output(0)); q -= 1
        # keyint 25 and 25 frames per repetition: every repetition is the same GOP (idr_pic_id alternates)
        if rep == 0:
            ok = ok and [hashlib.sha1(a).hexdigest() for a in got] == want
        else:
            ok = ok and [len(a) for a in got[1:]] == [len(a) for a in first[1:]] and got[1:] == first[1:]
        first = got if rep == 0 else first
    enc.close()
    print("RESULT " + json.dumps({"ok": bool(ok), "pipelined": bool(pipelined)}))
""")


import pytest  # noqa: E402


@pytest.mark.gpu
def test_two_processes_on_one_gpu_only_one_pipelines(tmp_path):
    """A pipelined session's deblock workgroups are resident and poll; enc_plan.h budgets them against the CUs of the
    device, which only holds if no OTHER process does the same on that device. The session that pipelines holds the
    device's lock file (enc_config.cpp, pipe_lock_acquire); the second process finds it taken and runs its frames back
    to back. Both emit the golden bytes of configs[2] (tests/golden/indep_fullsize.json), side by side on one GPU."""
    import json
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "indep_fullsize.json")))["cases"][1]
    assert want["case"][:6] == ["configs[2]", 1920, 1080, 24, 25, 25]
    (tmp_path / "want.json").write_text(json.dumps(want["au_sha1"]))
    script = tmp_path / "worker.py"
    script.write_text(TWO_PROC_WORKER % ROOT)
    env = dict(os.environ, CEDAR_LOCK_DIR=str(tmp_path))
    procs = [subprocess.Popen([sys.executable, str(script), str(tmp_path / ("flag%d" % i)), str(tmp_path / ("flag%d" % (1 - i))),
                               str(tmp_path / "want.json")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env)
             for i in range(2)]
    res = []
    for pr in procs:
        out, err = pr.communicate(timeout=280)
        assert pr.returncode == 0, err[-2000:]
        res.append(json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][0][7:]))
    assert all(r["ok"] for r in res), res
    assert sorted(r["pipelined"] for r in res) == [False, True], res
