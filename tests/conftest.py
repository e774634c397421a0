import gc
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _no_cyclic_collection_inside_a_gpu_test(request):
    """A torch.cuda.CUDAGraph kept alive only by a reference cycle (pytest.raises' ExceptionInfo -> traceback -> frame -> locals) is destroyed
    whenever the cyclic collector happens to run; when that is inside ANOTHER test's stream capture, hipGraphExecDestroy under the global capture
    mode aborts the process ("Fatal Python error: Aborted ... Garbage-collecting", seen once in round 5).  Collect between GPU tests, never inside one."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    gc.collect()
    gc.disable()
    try:
        yield
    finally:
        gc.enable()
        gc.collect()


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def ctx():
    """Default GPU context (gpu-marked tests only).  The HIP library must load and a
    device must be present: there is no CPU fallback to hide behind."""
    import bitnuc_amd
    from bitnuc_amd import build
    build.ensure_built()  # a fresh checkout has no .so (git-ignored): compile the product, never substitute it
    c = bitnuc_amd.Context(0)
    # every GPU parity test exercises the KERNELS: single words and tiny inputs run as batches of one on the device
    # (the library's size dispatch would otherwise keep them on the host: tests/test_host_path.py covers that side)
    c.set_variant("force_gpu", 1)
    yield c
    c.close()


@pytest.fixture(scope="session")
def sweep_ctx():
    """Context on the evidence build (all 47 codec variants + the ballot formulation)."""
    import bitnuc_amd
    from bitnuc_amd import build
    build.ensure_built(sweep=True)
    c = bitnuc_amd.Context(0, lib_path=build.LIB_SWEEP)
    c.set_variant("force_gpu", 1)
    yield c
    c.close()
