"""bench.py's bare N > 1 form on a box WITHOUT a GPU: the parent (standard library only, no torch) starts the ranks as a
child ``torch.distributed.run``; here every rank stops at "needs a ROCm GPU", and the parent must relay that and exit with
the child's non-zero code -- the launch form itself is what is tested (the GPU legs: tests/test_bench_multirank.py)."""
import os
import subprocess
import sys

from conftest import ROOT


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


def test_bare_gpus_2_starts_a_child_launcher_and_returns_its_code():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present: the bare form is covered by tests/test_bench_multirank.py")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    assert r.returncode not in (0, 124), (r.returncode, r.stderr[-2000:])
    assert "starting the ranks as a child" in r.stderr and "--nproc-per-node=2" in r.stderr
    assert r.stderr.count("bench.py needs a ROCm GPU") == 2          # both ranks were started and both said why they stopped
    assert r.stdout == ""                                            # nothing but a JSON line may ever reach stdout


def test_parent_of_the_bare_form_never_imports_torch():
    """The self-launch decision is taken before ``import torch`` (nothing in the parent can initialise HIP)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index("_self_launch_if_bare()\n") < src.index("\nimport torch")
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '1', '--help']\n"
            "import builtins; real = builtins.__import__\n"
            "def imp(name, *a, **k):\n"
            "    if name.split('.')[0] == 'torch': print('TORCH_IMPORT'); raise SystemExit(0)\n"
            "    return real(name, *a, **k)\n"
            "builtins.__import__ = imp\n"
            f"runpy.run_path({os.path.join(ROOT, 'bench.py')!r}, run_name='__main__')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=_env(), cwd=ROOT)
    assert "TORCH_IMPORT" in r.stdout                               # N = 1: falls through to the normal path (imports torch)
    code2 = code.replace("'--gpus', '1', '--help'", "'--gpus', '2', '--watchdog', '1'")
    r = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=120, env=_env(), cwd=ROOT)
    assert "TORCH_IMPORT" not in r.stdout and r.returncode == 124, (r.returncode, r.stdout, r.stderr[-500:])
