"""bench.py's in-run traffic measurement (`live_traffic`): the parsing of rocprofv3's counter CSV, the unit / gfx950 correction
(2 x FETCH_SIZE + WRITE_SIZE, both in KB) and the fall-backs, against a stand-in `rocprofv3` executable (no GPU, no profiler)."""
import importlib.util
import os
import stat
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

FAKE = r'''#!/usr/bin/env python3
import os, sys
a = sys.argv[1:]
ctr, d, o = a[a.index("--pmc") + 1], a[a.index("-d") + 1], a[a.index("-o") + 1]
assert "--kernel-trace" in a and "--sys-trace" not in a and a[a.index("--") + 2].endswith("prof_match.py")
assert os.environ["VFM_RECORDS"] == "8"
if os.environ.get("FAKE_FAIL"):
    sys.exit(3)
os.makedirs(os.path.join(d, "host"), exist_ok=True)
val = {"FETCH_SIZE": [68000.0, 69000.0, 68500.0], "WRITE_SIZE": [100.0, 110.0, 120.0]}[ctr]
with open(os.path.join(d, "host", o + "_counter_collection.csv"), "w") as f:
    f.write('"Correlation_Id","Kernel_Name","Counter_Name","Counter_Value"\n')
    if not os.environ.get("FAKE_EMPTY"):
        for i, v in enumerate(val):
            f.write(f'{i},"void vfmm::(anonymous namespace)::match_coarse_mx6q2_kernel<3, 2, false, 6, 4, 4>(Args)","{ctr}",{v}\n')
    f.write(f'9,"prep_stream_kernel<384, true>(Args)","{ctr}",123456.0\n')
'''


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_live_traffic_parses_the_counter_passes(tmp_path, monkeypatch):
    exe = tmp_path / "rocprofv3"
    exe.write_text(FAKE.replace("#!/usr/bin/env python3", "#!" + sys.executable))
    exe.chmod(exe.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    b = _bench()
    r = b.live_traffic(8)
    assert r is not None and r["launches"] == 3 and "match_coarse_mx6q2_kernel<3, 2, false, 6, 4, 4>" in r["kernel"]
    assert r["FETCH_SIZE_KB"] == 68500.0 and r["WRITE_SIZE_KB"] == 110.0
    assert r["hbm_bytes_per_launch"] == (2 * 68500.0 + 110.0) * 1024.0     # other kernels' rows are not counted
    monkeypatch.setenv("FAKE_EMPTY", "1")      # the profiler ran but saw no coarse kernel: fall back
    assert b.live_traffic(8) is None
    monkeypatch.delenv("FAKE_EMPTY")
    monkeypatch.setenv("FAKE_FAIL", "1")       # the profiler failed: fall back, never raise
    assert b.live_traffic(8) is None


def test_live_traffic_without_a_profiler(tmp_path, monkeypatch):
    monkeypatch.setenv("PATH", str(tmp_path))
    assert _bench().live_traffic(8) is None
