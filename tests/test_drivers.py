"""The example drivers keep the reference's command line and CSV (examples/choleskyTest01.cpp:74-85,
:273-277; examples/triangularTest02.cpp:188-271)."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
DRV = ROOT / "parsy_bench_amd" / "drivers"


def _inputs(tmp_path, name="small3d"):
    sys.path.insert(0, str(ROOT / "tools"))
    import make_mtx
    return make_mtx.write(name, str(tmp_path / name))


def test_drivers_are_built_and_usage_is_reported():
    from parsy_bench_amd.build import build_native
    build_native()
    for exe in ("choleskyTest.bin", "triangularTest.bin"):
        r = subprocess.run([str(DRV / exe)], capture_output=True, text=True)
        assert r.returncode != 0 and "input args are missing" in r.stdout


def test_reader_rejects_unsorted_or_upper_input(tmp_path):
    bad = tmp_path / "bad.mtx"
    bad.write_text("%%MatrixMarket matrix coordinate real symmetric\n2 2 3\n1 1 4.0\n1 2 -1.0\n2 2 4.0\n")
    r = subprocess.run([str(DRV / "choleskyTest.bin"), str(bad), "1", "1", "1", "0", "1", "2"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "lower triangle" in r.stderr


@pytest.mark.gpu
def test_choleskyTest_csv(tmp_path):
    mtx, order = _inputs(tmp_path)
    r = subprocess.run([str(DRV / "choleskyTest.bin"), mtx, "4", "1", "4", "0", "1", "2", order],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    fields = r.stdout.strip().split(",")
    assert fields[0] == mtx and fields[1:7] == ["4", "1", "4", "0", "1", "2"]
    total, par, root, symbolic, ordering = (float(v) for v in fields[7:12])
    assert total > 0 and par > 0 and root == 0 and symbolic > 0
    assert fields[12] == ""  # the reference's line ends with a comma


@pytest.mark.gpu
def test_triangularTest_output(tmp_path):
    mtx, order = _inputs(tmp_path)
    r = subprocess.run([str(DRV / "triangularTest.bin"), mtx, "4", "1", "4", "0", "1", "2", order],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = r.stdout.strip()
    assert "failed" not in out
    groups = out.split("*:")
    assert len(groups) == 5  # header+serial, H1, H2, H2-peeled, trailing empty
    head = groups[0].split(",")
    assert head[0] == mtx and int(head[3]) == 1000
    for g in groups[1:4]:
        assert len([v for v in g.split(",") if v]) == 5
