"""The experiment build (experiments/libkzhip_exp.so) against the product kernels and the oracle: tests/exp_cases.py in a
child pytest process with KZ_LIB_PATH set, so that this process (and every other test module) keeps the product library."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP_LIB = os.path.join(REPO, "experiments", "libkzhip_exp.so")


@pytest.mark.gpu
def test_experiment_kernels_agree_with_the_product_kernels():
    if not os.path.exists(EXP_LIB):  # (the experiment library is built best effort: __graft_entry__.build())
        pytest.skip("libkzhip_exp.so is not built: experiments/build.sh")
    env = dict(os.environ, KZ_LIB_PATH=EXP_LIB)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join("tests", "exp_cases.py"), "-x", "-q", "-m", "gpu",
                        "-p", "no:cacheprovider"], cwd=REPO, env=env, capture_output=True, text=True, timeout=1500)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-40:])
    assert r.returncode == 0, tail
    print(tail)
