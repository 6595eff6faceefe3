import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = os.path.join(GOLDEN, "cases")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """SD_TEST_ORDER=reverse | shuffle:<seed> runs the collected tests in another order (developer switch: the hipHostFree hang
    of rounds 1-5 only showed when two tests ran in an order the suite does not use; see DESIGN section 0, "found on the way")."""
    order = os.environ.get("SD_TEST_ORDER", "")
    if order == "reverse":
        items.reverse()
    elif order.startswith("shuffle:"):
        import random
        random.Random(int(order.split(":", 1)[1])).shuffle(items)


def case_names(include_errors=False, include_edthr=False):
    out = []
    for n in sorted(os.listdir(CASES)):
        if n.startswith("err_") and not include_errors:
            continue
        if "edthr" in n and not include_edthr:
            continue
        out.append(n)
    return out


def load_case(name):
    d = os.path.join(CASES, name)
    with open(os.path.join(d, "params.json")) as f:
        meta = json.load(f)
    meta["reads"] = os.path.join(GOLDEN, meta["inputs"][0])
    meta["monomers"] = os.path.join(GOLDEN, meta["inputs"][1])
    with open(os.path.join(d, "raw.tsv"), "rb") as f:
        meta["raw"] = f.read()
    meta["name"] = name
    return meta


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.build()
    return binding
