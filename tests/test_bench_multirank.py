"""bench.py's N > 1 code path, executed end to end on the ONE GPU of the test box: ranks launched by
torch.distributed.run exactly as the driver launches them, all on cuda:0 (DIINN_BENCH_ONE_DEVICE=1), exchanging over
gloo with host-staged messages because RCCL refuses two ranks on one device.  Everything but the wire is what the
8-GPU run executes: the band plan, the side-stream hand-off overlapped with rank 0's band, the checksum broadcast, the
max-over-ranks timing, the oracle check of every rank's rows, the strong-scaling legs with their same-run one-GPU
reference, the gather.  The timings of such a run mean nothing and are not looked at."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_bench(nproc, *extra):
    env = dict(os.environ, DIINN_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--backend", "gloo", "--steps", "3", "--warmup", "1",
           *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_two_ranks_weak_c1_with_strong_leg():
    res = _run_bench(2, "--workload", "c1", "--strong-legs", "c1", "--strong-steps", "2")
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["config"]["hr"] == [192, 96]
    assert res["checked"]["ok"] and res["checked"]["handoff_exact"]
    assert res["checked"]["max_err"] <= 1e-4
    (leg,) = res["strong"]
    assert leg["workload"] == "c1" and leg["n_gpus"] == 2 and leg["checked_ok"]
    assert leg["ms_1gpu"] > 0 and leg["ms_Ngpu"] > 0 and leg["handoff_ms"] >= 0
    assert "x2 on 48^2 LR" in res["metric"]                      # the metric names the workload that ran
    assert "cpu_baseline" not in res and "target_shape" not in res


def test_four_ranks_strong_with_gather_and_bcast():
    res = _run_bench(4, "--workload", "c1", "--scaling", "strong", "--gather", "--no-strong")
    assert res["n_gpus"] == 4 and res["scaling"] == "strong" and res["config"]["hr"] == [96, 96]
    assert res["checked"]["ok"] and res["checked"]["handoff_exact"] and res["gather_ms"] > 0
    res = _run_bench(2, "--workload", "c1", "--scaling", "strong", "--dist-mode", "bcast", "--no-strong")
    assert res["checked"]["ok"] and res["checked"]["handoff_exact"]


def test_two_ranks_default_workload_bf16():
    """The driver's own command line shape (default workload c2, weak) with the optional bf16 arithmetic and the
    default strong legs replaced by a small one (tgt / c3 would only burn test time on a shared GPU)."""
    res = _run_bench(2, "--compute", "bf16_full", "--strong-legs", "c1", "--strong-steps", "2")
    assert res["config"]["name"] == "c2" and res["config"]["hr"] == [2048, 1024] and res["dtype"] == "bf16_full"
    assert res["checked"]["ok"] and res["strong"][0]["checked_ok"]
