"""The committed fixtures under tests/golden/ are exactly what the committed generator produces from the reference.

The reference ships no tests, so tests/golden/*.npz is the only pin of the oracle.  Whenever the reference checkout
is present (the build container; never the GPU box) this re-runs tests/golden/gen_golden.py against it into a temporary directory
and compares key sets and arrays bit for bit with the committed files (`--check`).  The two config-dim fixtures (2.6 GB of weights
each) are left to `python tests/golden/gen_golden.py --check` without --skip-full."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(ROOT, "tests", "golden", "gen_golden.py")


@pytest.mark.skipif(not os.path.isdir("/root/reference/diffusion"), reason="needs the reference checkout (build container only)")
def test_committed_fixtures_are_what_the_generator_produces():
    r = subprocess.run([sys.executable, GEN, "--check", "--skip-full"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "identical" in r.stdout and "DIFF" not in r.stdout


def test_check_mode_notices_a_drifted_fixture(tmp_path):
    """compare_trees itself: a changed value, a changed dtype and a missing key are each reported (no reference needed)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_golden", GEN)
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    a, b = tmp_path / "fresh", tmp_path / "committed"
    a.mkdir(); b.mkdir()
    x = np.arange(6, dtype=np.float32).reshape(2, 3)
    np.savez_compressed(a / "same.npz", x=x, t=np.array([1, 2]))
    np.savez_compressed(b / "same.npz", x=x.copy(), t=np.array([1, 2]))
    assert gen.compare_trees(str(a), str(b)) == []
    y = x.copy(); y[1, 2] = np.nextafter(y[1, 2], np.float32(10))           # one ulp
    np.savez_compressed(b / "same.npz", x=y, t=np.array([1, 2]))
    assert any("same.npz[x]" in d for d in gen.compare_trees(str(a), str(b)))
    np.savez_compressed(b / "same.npz", x=x.astype(np.float64), t=np.array([1, 2]))
    assert any("same.npz[x]" in d for d in gen.compare_trees(str(a), str(b)))
    np.savez_compressed(b / "same.npz", x=x)                                  # key the generator emits is missing
    assert any("key sets differ" in d for d in gen.compare_trees(str(a), str(b)))
    np.savez_compressed(a / "new.npz", x=x)
    assert any("not committed" in d for d in gen.compare_trees(str(a), str(b)))


def test_every_sampler_fixture_carries_the_per_row_timestep_case():
    for s in ("s0", "s1", "s2", "s3", "s4", "s5"):
        z = np.load(os.path.join(ROOT, "tests", "golden", f"sampler_{s}.npz"))
        assert "t_rows" in z.files and "eps_rows" in z.files, s
        assert z["eps_rows"].shape == (int(z["dims"][5]), int(z["dims"][3])), s
