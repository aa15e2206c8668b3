"""GPU: the pass ladder once more on the bounds-checked build of the library (sgc_kernels.h SGC_CHECK: every pool block id, block
fill, pool write and miss-run slot is compared with its bound; a violation is reported by sgc_sample_finish instead of faulting).
A re-arranged k_partition / k_count_slices loop cannot walk off its scratch silently."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_partitioned_pass_stays_inside_its_scratch():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_check_run.py")], capture_output=True, timeout=1200)
    out = p.stdout.decode()
    assert p.returncode == 0 and "CHECKED BUILD OK" in out, (out[-1500:], p.stderr.decode()[-3000:])
