"""bench.py's multi-rank path, rehearsed on ONE device: two ranks share the GPU, the all-reduce goes over gloo
(HF_BENCH_BACKEND=gloo; the measured configuration is nccl = RCCL, one rank per GPU, launched by the driver).
The two-rank job must trace exactly the rays of the one wavefront (tile partition, BASELINE configs[3]) and its
reduced gradient must be the single-rank gradient; the row band it restricts the all-reduce to must lie inside the
texture; both the overlapped (headline) and the serial step time are reported."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZE = ["--grid", "1024", "--film", "512", "--spp", "16", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0"]


def _run(cmd, extra_env):
    env = dict(os.environ, HF_BENCH_EXTRAS="0", **extra_env)
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def _free_port():
    import socket
    with socket.socket() as s:   # bind to port 0: the kernel picks a free one (a hard-coded port fails when it is busy)
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_over_gloo_equal_one_rank(hf):
    one = _run([sys.executable, "bench.py", "--gpus", "1"] + SIZE, {})
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2"] + SIZE,
               {"HF_BENCH_BACKEND": "gloo"})
    # the line describes the collective it ran: backend and world size as the process group reports them, one entry per rank
    col = two["collective"]
    assert col["backend"] == "gloo" and col["world_size"] == 2 and [r["rank"] for r in col["ranks"]] == [0, 1]
    assert all(r["host"] and r["name"] and r["device"] is not None for r in col["ranks"])
    assert one["collective"] is None
    for line in (one, two):   # which library the numbers come from
        assert line["library"]["path"].endswith("libhf.so") and len(line["library"]["sha256_16"]) == 16
        assert line["library"]["hf_lib_override"] == bool(os.environ.get("HF_LIB"))
    R = 512 * 512 * 16
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert one["config"]["rays_per_step"] == R and two["config"]["rays_per_step"] == R     # sum over ranks = the one wavefront
    assert two["config"]["rays_rank0"] == R // 2                                            # 32x32-pixel tiles, interleaved
    for key in ("grad_l2", "grad_checksum"):
        assert abs(two[key] - one[key]) <= 1e-5 * abs(one[key]), (key, one[key], two[key])  # float-atomic order only
    lo, hi = two["allreduce"]["rows"]
    assert 0 <= lo < hi <= 1024 and two["allreduce"]["bytes"] == (hi - lo) * 1024 * 4
    assert one["allreduce"]["rows"] == [lo, hi]                 # the union over ranks is the single rank's band
    assert two["allreduce"]["serial_ms_per_step"] > 0 and two["allreduce"]["serial_value"] > 0   # reported (two steps time nothing)
    assert one["allreduce"]["serial_ms_per_step"] is None
