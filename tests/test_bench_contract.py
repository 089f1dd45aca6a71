"""bench.py's contract on the GPU box: the one-line JSON at N=1, and the N>1 launch path
(torchrun, barrier + max-over-ranks timing, streams sharded over ranks with no data-path
collective). A one-GPU box has one card, so the two ranks of the second test share device 0
and rendezvous over gloo; on the 8-GPU node the driver runs the same code with one rank per GPU
over RCCL."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
        "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.gpu
def test_bench_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "25", "--warmup", "25",
                        "--cpu-frames", "2"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_line(r.stdout)
    assert KEYS <= set(j)
    assert j["n_gpus"] == 1 and j["steps"] == 25 and j["warmup"] == 25 and j["scaling"] == "weak"
    assert j["unit"] == "macroblocks/s" and j["dtype"] == "u8" and j["higher_is_better"] is True
    assert j["vs_baseline"] is None and "workload" in j["config"] and "model" not in j["config"]
    assert abs(j["value"] - 8160 * 25 / (j["ms_per_step"] * 25e-3)) < 1e-3 * j["value"]
    assert len(j["region_runs"]) == 5 and j["region_runs"][0] == j["value"] and min(j["region_runs"]) > 0  # the spread of the number
    rf, cb = j["roofline"], j["cpu_baseline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 * rf["frac"] and rf["achieved"] > 0  # achieved is printed rounded
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0
    assert cb["all_cores"]["cores"] >= 1 and cb["all_cores"]["value"] > 0  # 1 thread and all cores
    assert j["bitexact_vs_cpu"] is True  # the GPU bytes of the sample equal the oracle's
    # BASELINE configs[1] and configs[3] (one-GPU forms) ride on the same line; the traffic figure says where it is from
    assert [c["macroblocks_per_frame"] for c in j["configs"]] == [54 * 30, 240 * 135, 240 * 135] and all(c["value"] > 0 for c in j["configs"])
    assert rf["traffic"] is None or "static" in rf["traffic_source"]
    ki = [k for k in rf["kernels"] if k["kernel"] == "k_inter"][0]
    assert ki["true_bound"]["valu_slots_per_macroblock"] > 0 and 0 < ki["true_bound"]["frac"] < 1


@pytest.mark.gpu
def test_bench_two_ranks_shard_streams():
    env = dict(os.environ, BENCH_FORCE_DEVICE="0", BENCH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "25", "--warmup", "25"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_line(r.stdout)                       # rank 0 prints, once
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["value"] > 0
    # whole-job value: both ranks' macroblocks over the slowest rank's time
    assert abs(j["value"] - 2 * 8160 * 25 / (j["ms_per_step"] * 25e-3)) < 1e-3 * j["value"]
    assert "no collective" in j["config"]["parallelism"]


@pytest.mark.gpu
def test_bench_slices_across_ranks_concatenate():
    """BASELINE configs[3] harness: rank k codes slice k of one picture, the host concatenates the slice
    NALs; the run itself checks that the concatenation IS the one-session stream. Two ranks on the one
    card of this box over gloo; on the 8-GPU node the same code runs one rank per GPU."""
    env = dict(os.environ, BENCH_FORCE_DEVICE="0", BENCH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--slices-across-gpus", "--width", "1280", "--height", "720",
           "--steps", "12", "--warmup", "3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["scaling"] == "strong"
    assert j["config"]["slices_concatenate_to_one_session_stream"] is True
    assert abs(j["value"] - 3600 * 12 / (j["ms_per_step"] * 12e-3)) < 1e-3 * j["value"]   # ONE picture per step
    assert "slices x2" in j["config"]["parallelism"] and j["config"]["idr_in_timed_region"] == 1


@pytest.mark.gpu
def test_bench_gop_shards_across_ranks_concatenate():
    """SURVEY 8(e) row 2 as a harness: ONE stream, GOP g on rank g mod N; the run checks that the shards'
    bytes, in GOP order, are the serial stream. Two ranks on the one card of this box over gloo."""
    env = dict(os.environ, BENCH_FORCE_DEVICE="0", BENCH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--gops-across-gpus", "--width", "640", "--height", "368",
           "--keyint", "5", "--steps", "4", "--warmup", "5"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["config"]["gop_shards_concatenate_to_the_serial_stream"] is True
    assert abs(j["value"] - 2 * 5 * 4 * 920 / (j["ms_per_step"] * 4e-3)) < 1e-3 * j["value"]
