"""bench.py's stdout contract: ONE compact JSON line (< 2 KB... hard limit 4 KB) that carries `roofline` and `cpu_baseline`.
Round 5's line had grown to 27 KB and the driver's record of it came out `parsed: null`; nothing checked the line against
its consumer.  These tests format a worst-case result without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def fake_full():
    long = "x" * 20000  # prose that must stay in the extras file
    return {
        "metric": "iLQR iterations/sec (batch x T timesteps/sec), acrobot T=500 batch=4096",
        "value": 3653489123.4567890123, "unit": "trajectory-timesteps/s", "n_gpus": 8, "steps": 20, "warmup": 5,
        "ms_per_step": 0.55912345678901, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "acrobot n=4 m=1 T=499 (500 knots), box-QP limits active, FD derivatives + backward/box-QP + 11-alpha search + accept, fixed work",
                   "batch_per_gpu": 4096, "global_batch": 32768, "T": 499, "u_limit": 1.5, "parallelism": "batch shards x8, one all_gather of costs",
                   "prose": long},
        "roofline": {"bound": "hbm", "kernel": "k_solve_hex", "achieved": 1148.123456789, "peak": 8000.0, "unit": "GB/s", "frac": 0.143515432,
                     "limiter": "valu_issue", "bound_frac": 0.4812345678, "traffic": 18160000000.123, "traffic_source": long,
                     "algorithmic_bytes_per_launch": 12753960960.0, "avg_launch_ms": 11.1091234, "limiter_note": long},
        "roofline_issue": {"source": long},
        "cpu_baseline": {"value": 5651234.5678, "unit": "trajectory-timesteps/s", "cores": 256, "kind": "port", "spread": 0.031234,
                         "sample": "median of 3 runs of 1024 trajectories x 40 fixed-work iterations of this workload (4.1 s each), oracle/liboracle_ilqr.so, OpenMP on all host threads",
                         "backward_only_value": 1.0e7},
        "rccl_ranks": 8, "backward_only_timesteps_per_s": 4.2e9, "stages": {"solve": {"note": long}},
        "configs": {"lq": {"note": long}}, "extras": "gpurun_out/bench_extras.json",
    }


def test_compact_record_is_small_and_complete():
    line = bench.compact_record(fake_full())
    assert "\n" not in line
    assert len(line) < bench.COMPACT_LIMIT <= 2048 < 4096
    rec = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in rec, k
    assert rec["config"]["workload"] and "model" not in rec["config"]
    assert rec["roofline"]["bound"] in ("hbm", "mfma")
    for k in ("achieved", "peak", "unit", "frac", "traffic"):
        assert k in rec["roofline"], k
    assert abs(rec["roofline"]["frac"] - rec["roofline"]["achieved"] / rec["roofline"]["peak"]) < 1e-5
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in rec["cpu_baseline"], k
    assert rec["cpu_baseline"]["kind"] in ("reference", "port")
    assert abs(rec["value"] / 3653489123.4567890123 - 1) < 1e-5


def test_compact_record_refuses_to_grow():
    full = fake_full()
    full["config"]["workload"] = "y" * 3000
    try:
        bench.compact_record(full)
    except AssertionError as e:
        assert "extras" in str(e)
    else:
        raise AssertionError("a 3 KB workload string went to stdout")


def test_bench_without_a_gpu_fails_loudly_and_prints_no_record():
    """No CPU fallback: on a box without a HIP device bench.py exits non-zero and stdout carries no JSON line."""
    import torch
    if torch.cuda.is_available():
        return
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert p.stdout.strip() == ""
    assert "needs a GPU" in p.stderr


import pytest  # noqa: E402


@pytest.mark.gpu
def test_default_bench_invocation_prints_one_parsable_line(tmp_path):
    """The driver's own invocation (no flags beyond steps / warmup): stdout is exactly one JSON line under the limit,
    with roofline and cpu_baseline, and the full record lands in the extras file."""
    extras = tmp_path / "extras.json"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--extras-out", str(extras)],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    assert len(lines[0]) < bench.COMPACT_LIMIT
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["steps"] == 5 and rec["warmup"] == 2
    assert rec["config"]["batch_per_gpu"] == 4096 and rec["config"]["T"] == 499
    assert rec["roofline"]["bound"] == "hbm" and rec["roofline"]["kernel"] == "k_solve_hex"
    assert 0 < rec["roofline"]["frac"] < 1 and rec["roofline"]["avg_launch_ms"] > 0
    assert rec["cpu_baseline"]["kind"] == "port" and rec["cpu_baseline"]["value"] > 0
    assert abs(rec["value"] - 4096 * 499 / (rec["ms_per_step"] * 1e-3)) / rec["value"] < 1e-4
    full = json.load(open(extras))
    assert "stages" in full and "roofline_issue" in full and full["configs"] == {}


@pytest.mark.gpu
def test_two_rank_bench_invocation_as_the_driver_launches_it(tmp_path):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 ... bench.py --gpus 2 ...` -- the driver's command for N > 1 -- on the
    one GPU of the test box: --test-backend gloo puts both ranks on device 0 and carries the one collective over gloo, everything else
    is the code an 8-GPU node runs (env parsing, the shard of the global batch, barriers, max-over-ranks timing, rank 0 alone printing).
    One compact line, n_gpus = 2, value = the whole job's timesteps over the slowest rank's time, no cpu_baseline (N = 1 only)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    extras = tmp_path / "extras.json"
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "512", "--test-backend", "gloo", "--extras-out", str(extras)],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["scaling"] == "weak" and "cpu_baseline" not in rec
    assert rec["config"]["batch_per_gpu"] == 512 and rec["config"]["global_batch"] == 1024
    assert abs(rec["value"] - 2 * 512 * 499 / (rec["ms_per_step"] * 1e-3)) / rec["value"] < 1e-4
    assert json.load(open(extras))["collective_backend"] == "gloo"


def test_multi_gpu_request_on_a_box_without_them_fails_loudly():
    """bench.py --gpus 8 started plainly on a box with fewer devices says so instead of measuring one GPU (no GPU at all: the no-fallback message)."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        return  # (an 8-GPU node would simply run the job)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and p.stdout.strip() == ""
    assert ("HIP device(s) visible" in p.stderr) or ("needs a GPU" in p.stderr)
