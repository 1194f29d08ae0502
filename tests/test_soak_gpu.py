"""Seeded 60-second slices of the two soak tools, as child processes under a time limit (the hang of
round 3 was found by tools/soak_search.py in minutes and by none of 133 green tests).  Search: a fresh index per case, several
searches per index lifetime, value ranges up to FLT_MAX and NaN rows, the prefilter path bit-equal to the exact fp32 kernels.
Encoder: a long-lived handle bit-equal to a fresh one on every random batch (sizes 1-700, lengths 16-512, both GEMM families).
A hang shows up as the time limit, a device-detected invariant slip as HAC_ERR_INTERNAL, a mismatch as exit code 1."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_soak(tool, *args, limit):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)] + [str(a) for a in args], cwd=ROOT,
                       capture_output=True, text=True, timeout=limit)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-2500:])
    assert "soak ok:" in p.stdout, p.stdout[-1500:]
    return p.stdout


@pytest.mark.parametrize("seed", [20261004])
def test_search_soak_slice(seed):
    out = run_soak("soak_search.py", seed, 60, limit=420)
    n_cases = int(out.rsplit("soak ok:", 1)[1].split()[0])
    assert n_cases >= 5, out[-500:]


@pytest.mark.parametrize("seed", [20261004])
def test_encoder_soak_slice(seed):
    out = run_soak("soak_encoder.py", seed, 60, 2, limit=420)
    assert int(out.rsplit("soak ok:", 1)[1].split()[0]) >= 10, out[-500:]
