"""Repository rules: the oracle is test infrastructure; the product never imports it."""
import os
import re

from tests.conftest import REPO


def _py_files(root):
    for d, _, fs in os.walk(root):
        if "__pycache__" in d:
            continue
        for f in fs:
            if f.endswith(".py"):
                yield os.path.join(d, f)


def test_product_never_imports_the_oracle_or_the_reference():
    bad = []
    for path in _py_files(os.path.join(REPO, "syntalker_amd")):
        src = open(path).read()
        if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "/root/reference" in src:
            bad.append(path)
    assert not bad, bad


def test_only_checkers_import_the_oracle():
    users = set()
    for path in _py_files(REPO):
        rel = os.path.relpath(path, REPO)
        if rel.startswith(("oracle", ".git", "gpurun_out")):
            continue
        if re.search(r"^\s*(from|import)\s+oracle\b", open(path).read(), flags=re.M):
            users.add(rel.split(os.sep)[0])
    assert users <= {"tests", "bench.py", "__graft_entry__.py", "scripts"}, users


def test_gpu_paths_never_read_the_reference_tree():
    for rel in ("bench.py", "__graft_entry__.py"):
        assert "/root/reference" not in open(os.path.join(REPO, rel)).read()
    for path in _py_files(os.path.join(REPO, "tests")):
        if path.endswith(("make_golden.py", "make_vq_golden.py", "make_longform_golden.py", "make_train_golden.py", "make_frechet_golden.py", "test_layout.py")):   # golden generators run in the build container only
            continue
        if path.endswith("test_dropin_reference_driver.py"):
            # CPU-only, skipped where the tree does not exist (the GPU box): the reference's own driver code run against this build
            src = open(path).read()
            assert src.count("/root/reference") == 1 and "skipif(not os.path.isdir(REF_TREE)" in src and "pytest.mark.gpu" not in src
            continue
        if path.endswith("test_config.py"):
            # one CPU-only test there reads the reference's own YAML files where the tree exists (the build container) and is skipped
            # elsewhere; its -m gpu tests carry their configurations inline
            src = open(path).read()
            assert src.count("/root/reference") == 1 and "skipif(not os.path.isdir(REF_CONFIGS)" in src
            continue
        assert "/root/reference" not in open(path).read(), path
