"""The selection kernel (k nearest chosen inside the grouped gather for groups in which many queries meet k photons)
and the radius bound in front of it.  By default a gather goes that way only when thousands of groups qualify; here
the thresholds are forced down through the environment (read once per process, hence the child process), so that
every gather of the kNN parity tests - goldens from the reference, random sets against the oracle, photons tied at
one distance, whole iterations - is decided by the selection kernel wherever a query meets k photons."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle_py as O
from pathtracer_amd.tracer import PhotonMap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def test_parity_suites_with_every_group_selected():
    env = dict(os.environ, TKHIP_SEL_MIN="1", TKHIP_SEL_GROUPS="1")
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
           "tests/test_parity_gpu.py", "tests/test_ref_pinned_gpu.py",
           "-k", "(knn or gather or iterations or trace_tile or odd_image) and not full_size"]   # (the full-size ones take 20 s)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


def test_large_radii_thousands_of_groups(gpu_ctx):
    """The default thresholds: 160 k queries whose radii hold about ten times k photons (what the first iteration of
    a render looks like).  The radii are cut, nearly every group goes through the selection kernel; found, radius and
    index sets against the oracle on a sample, properties on all."""
    rng = np.random.default_rng(5)
    n, k, nq = 120000, 100, 160000
    pos = rng.random((n, 3)).astype(np.float32) * np.array([4, 4, 0.02], np.float32)      # a slab: a surface with some depth
    power = rng.random((n, 3)).astype(np.float32)
    th = rng.integers(0, 256, n).astype(np.uint8)
    ph = rng.integers(0, 256, n).astype(np.uint8)
    qp = rng.random((nq, 3)).astype(np.float32) * np.array([4, 4, 0.02], np.float32)
    qn = np.tile(np.array([[0, 0, 1]], np.float32), (nq, 1))
    qwo = qn + rng.normal(scale=0.3, size=(nq, 3)).astype(np.float32)
    qR = rng.random((nq, 3)).astype(np.float32)
    # n / 16 photons per unit area: a disc of r2 = 0.04 holds about 940, one of 0.006 about 140
    r2 = np.where(rng.random(nq) < 0.8, 0.04, 0.006).astype(np.float32)
    gm = PhotonMap(gpu_ctx, n, k)
    gm.store(pos, power, th, ph)
    gm.build()
    flux, found, r2o, idx = gm.gather(qp, qn, qwo, qR, r2, mode=0)
    assert (found == k).mean() > 0.95
    assert ((found < k) == (r2o == r2)).all() and (r2o <= r2).all()
    assert ((idx >= 0).sum(axis=1) == found).all()
    assert np.isfinite(flux).all() and (flux >= 0).all()
    om = O.OracleMap(n, k)
    om.store(pos, power, th, ph)
    om.build(1)
    pick = rng.choice(nq, 3000, replace=False)
    of, ofound, or2, oidx = om.gather(qp[pick], qn[pick], qwo[pick], qR[pick], r2[pick])
    assert (found[pick] == ofound).all() and (r2o[pick] == or2).all() and (idx[pick] == oidx).all()
    assert (np.abs(flux[pick] - of) <= 1e-5 * np.abs(of) + 1e-10).all()
    # a second gather on the same map (the history of the first now says: many met k) and the per-query kernel agree
    f2, n2, r2b, i2 = gm.gather(qp, qn, qwo, qR, r2, mode=0)
    assert (n2 == found).all() and (r2b == r2o).all() and (i2 == idx).all()
    f1, n1, r1, i1 = gm.gather(qp[:20000], qn[:20000], qwo[:20000], qR[:20000], r2[:20000], mode=1)
    assert (n1 == found[:20000]).all() and (r1 == r2o[:20000]).all() and (i1 == idx[:20000]).all()
    gm.close()
    om.close()
