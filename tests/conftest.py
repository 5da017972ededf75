import os
import sys
import pytest
# torch first: it bundles its own libamdhip64 under the same SONAME the library links from /opt/rocm, and whichever is loaded first serves
# both.  When libhobbit_hip.so came first (a partial run, e.g. `pytest tests/test_gpu_parity.py -k shard`), a later `import torch` in a
# test found "No HIP GPUs are available"; a full run imports torch at collection time anyway (tests/test_dist_gloo.py).
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """HOBBIT_TEST_ORDER=reverse / a number (shuffle seed): run the tests in another order, to catch tests that only pass
    because an earlier one warmed something up (context, libc generator, oracle globals)."""
    order = os.environ.get("HOBBIT_TEST_ORDER")
    if not order:
        return
    if order == "reverse":
        items.reverse()
    else:
        import random
        random.Random(int(order)).shuffle(items)


@pytest.fixture(scope="session")
def oracle():
    """The plain-C CPU restatement (oracle/hobbit_oracle.c); built on demand with gcc."""
    from oracle import pyoracle
    pyoracle.build_oracle()
    return pyoracle.Oracle()


@pytest.fixture(scope="session")
def ref():
    """The real reference (oracle/_ref); only where it has been built (needs /root/reference)."""
    from oracle import pyoracle
    if not pyoracle.ref_available():
        if os.path.isdir("/root/reference/src"):
            pyoracle.build_ref()
        else:
            pytest.skip("oracle/_ref not built and /root/reference absent")
    return pyoracle.Ref()
