import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import _pkgload  # noqa: E402

pkg = _pkgload.load()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def mts():
    """the product package (mitsuba-renderer_amd)"""
    return pkg


@pytest.fixture(scope="session")
def orc():
    """the CPU oracle (test infrastructure)"""
    import orc as _orc
    _orc.lib()
    return _orc


@pytest.fixture(scope="session")
def gpu_lib(mts):
    """libmtsgpu.so must be the thing that runs on the GPU box: fail loudly if it is missing"""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu test selected but no GPU is visible")
    return mts.lib()


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def chord_rays(n, centre, radius, seed):
    """test_kd.cpp:103-121 method: chords between two uniform points on a bounding sphere"""
    rng = np.random.RandomState(seed)
    def sphere(k):
        z = 1 - 2 * rng.rand(k)
        r = np.sqrt(np.maximum(0, 1 - z * z))
        phi = 2 * np.pi * rng.rand(k)
        return np.stack([r * np.cos(phi), r * np.sin(phi), z], axis=1)
    c = np.asarray(centre, dtype=np.float64)
    p1 = c + sphere(n) * radius
    p2 = c + sphere(n) * radius
    d = p2 - p1
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), dtype=np.float32)
    rays[:, 0:3] = p1
    rays[:, 3] = 1e-4   # Epsilon -> adaptive epsilon branch (skdtree.cpp:116-119)
    rays[:, 4:7] = d
    rays[:, 7] = np.inf
    return rays
