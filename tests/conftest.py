import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: builds a 113 M-row real BWT first (~10 min of host time): deselected unless "
                                       "MOVI_SLOW_TESTS=1 (tools/r05_real_bwt.sh runs them; the log is committed under profiles/)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if os.environ.get("MOVI_SLOW_TESTS") != "1":            # (deselected, not skipped: they are not part of the default suites)
        slow = [it for it in items if "slow" in it.keywords]
        if slow:
            config.hook.pytest_deselected(items=slow)
            items[:] = [it for it in items if "slow" not in it.keywords]
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def built_lib():
    """libmovi_hip.so, (re)built if stale.  hipcc cross-compiles without a GPU."""
    so = os.path.join(ROOT, "movi_amd", "lib", "libmovi_hip.so")
    csrc = os.path.join(ROOT, "movi_amd", "csrc")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".hpp", ".cpp"))] + [os.path.join(ROOT, "include", "movi_hip.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-j8", "-C", csrc], stdout=subprocess.DEVNULL)
    return so


@pytest.fixture(scope="session")
def golden_image():
    def get(mode):
        name = {6: "index_regular-thresholds", 8: "index_blocked-thresholds"}[mode]
        with open(os.path.join(GOLDEN, name, "index.movi"), "rb") as f:
            return f.read()
    return get


def read_fastx(path):
    """[(id, seq bytes)] with the reference's id rule (src/batch_loader.cpp:117-119)."""
    out = []
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    if not lines or not lines[0]:
        return out
    if lines[0][:1] == b"@":
        for i in range(0, len(lines) - 3, 4):
            if lines[i][:1] == b"@":
                out.append((_read_id(lines[i]), lines[i + 1].rstrip()))
    else:
        cur, seq = None, []
        for ln in lines:
            if ln[:1] == b">":
                if cur is not None:
                    out.append((cur, b"".join(seq)))
                cur, seq = _read_id(ln), []
            elif cur is not None:
                seq.append(ln.rstrip())
        if cur is not None:
            out.append((cur, b"".join(seq)))
    return out


def _read_id(header):
    k = len(header)
    for i in range(1, len(header)):
        if header[i:i + 1] in (b" ", b"\t", b"\r"):
            k = i
            break
    return header[1:k + 1] if k < len(header) else header[1:]


def golden_sorted_pmls():
    with open(os.path.join(GOLDEN, "sample.fastq.pmls.sorted")) as f:
        lines = f.read().split("\n")
    return sorted(l for l in lines if l and not l.startswith(">")), sorted(l for l in lines if l.startswith(">"))


def stdout_line(pml_emission_order):
    """`--stdout` text of one read: values in read order, each followed by a space
    (include/move_query.hpp:33-37 + src/utils.cpp:214-219)."""
    return "".join("%d " % v for v in pml_emission_order[::-1])


def classify_py(pml, thr, bin_width=150):
    """Classifier::classify, src/classifier.cpp:99-143: (found, avg max, bins above, bins below)."""
    n, start, above, below, s, bins = len(pml), 0, 0, 0, 0, 0
    while start < n:
        end = start + bin_width if start + bin_width < n else n
        if n - end < bin_width:
            end = n
        mx = int(max(pml[start:end]))
        above += mx >= thr
        below += mx < thr
        s += mx
        bins += 1
        start = end
    found = above / (above + below + 0.0) > 0.5
    return found, s / bins, above, below


def vector_kernel(name):
    """A walk kernel's name with the reset-mask instantiation (RING = 2: the vector through masks, movi_pml_device's default on batches
    of short reads) read as the one that writes the vector itself (RING = 0): for asserts that are about the table layout the launch
    policy picked, not about how the PMLs left the kernel."""
    return name[:-2] + "0>" if name.startswith("pml_kernel_flatp<") and name.endswith(", 2>") else name


@pytest.fixture
def packer_paths():
    """Handles created inside the test write PML vectors straight from the walk ("pml_via_mask" 0: register packer / LDS ring) --
    for the tests that are about those output paths, or that assert their kernels by name."""
    old = os.environ.get("MOVI_PML_VIA_MASK")
    os.environ["MOVI_PML_VIA_MASK"] = "0"
    yield
    if old is None:
        os.environ.pop("MOVI_PML_VIA_MASK", None)
    else:
        os.environ["MOVI_PML_VIA_MASK"] = old
