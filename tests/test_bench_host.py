"""Host-side logic of bench.py that needs no GPU: the in-run HBM-traffic probe (exercised against a stand-in profiler
executable), the leg tables, the small helpers."""
import importlib.util
import os
import stat
import sys
import types

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_FAKE = r'''#!/usr/bin/env python3
# stand-in for rocprofv3: writes the counter_collection.csv the real tool writes for `--pmc <counter>`
import os, sys
a = sys.argv[1:]
counter = a[a.index("--pmc") + 1]
out = a[a.index("-d") + 1]
mode = os.environ.get("FAKE_PROF_MODE", "ok")
if mode == "fail":
    sys.stderr.write("boom\n"); sys.exit(3)
os.makedirs(os.path.join(out, "host", "123"), exist_ok=True)
rows = ["Correlation_Id,Dispatch_Id,Agent_Id,Kernel_Name,Counter_Name,Counter_Value"]
val = {"FETCH_SIZE": 138150.0, "WRITE_SIZE": 12288.0}[counter]
if mode != "norows":
    for i in range(4):
        rows.append(f'{i},{i},0,"void decode_kernel<2, true, false>(DecodeParams)",{counter},{val}')
rows.append(f'9,9,0,"precompute_P_wino_kernel(PWinoParams)",{counter},777.0')
open(os.path.join(out, "host", "123", "1_counter_collection.csv"), "w").write("\n".join(rows) + "\n")
assert "--" in a and a[a.index("--") + 1].endswith("python3") or "python" in a[a.index("--") + 1]
assert "--no-traffic" in a and "--no-side-legs" in a          # the child must not recurse or run the extra legs
'''


def _args(**kw):
    d = dict(workload="c2", compute="f32", sin="default")
    d.update(kw)
    return types.SimpleNamespace(**d)


def test_traffic_probe_parses_and_corrects_the_counters(bench, tmp_path, monkeypatch):
    """bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 over the dominant kernel's dispatches only (MI355X_MICROARCH.md:
    gfx950 tallies wide reads at half their bytes; separate passes per counter)."""
    fake = tmp_path / "rocprofv3"
    fake.write_text(_FAKE)
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", f"{tmp_path}:{os.environ['PATH']}")
    monkeypatch.setenv("FAKE_PROF_MODE", "ok")
    got, how = bench.measure_traffic(_args())
    assert got == int((2 * 138150.0 + 12288.0) * 1024)
    assert "measured by this run" in how and "FETCH_SIZE 138150" in how
    # a profiler that fails, or reports nothing for the kernel, yields (None, reason) -- never an exception
    monkeypatch.setenv("FAKE_PROF_MODE", "fail")
    got, how = bench.measure_traffic(_args())
    assert got is None and "exited 3" in how
    monkeypatch.setenv("FAKE_PROF_MODE", "norows")
    got, how = bench.measure_traffic(_args(sin="hw"))
    assert got is None and "no rows" in how


def test_leg_tables_and_helpers(bench):
    assert [leg[:2] for leg in bench.SIDE_LEGS] == [("c5", "f32"), ("c5", "bf16_full"), ("c5", "bf16"), ("c1", "f32")]
    assert bench.STRONG_LEGS == {2: ["tgt", "c3"], 4: ["tgt", "c3"], 8: ["tgt", "c4"]}
    assert set(bench.WORKLOADS) == {"c1", "c2", "c3", "tgt", "c4", "c5"} and set(bench.METRIC) == set(bench.WORKLOADS)
    assert bench.WORKLOADS["c2"][:2] == ((256, 256), (1024, 1024))           # BASELINE.json's metric is quoted on c2
    assert bench.pct([3.0, 1.0, 2.0], 0.5) == 2.0 and bench.pct([], 0.5) is None
    assert bench.effective_cores() >= 1
    assert bench.CHECK_TOL["f32"] == (1e-4, True) and bench.CHECK_TOL["bf16x3"] == (1e-4, True)
