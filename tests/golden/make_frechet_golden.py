#!/usr/bin/env python3
"""Known answers for the Fréchet distance, produced by EXECUTING the reference's own `FIDCalculator.frechet_distance` /
`calculate_frechet_distance` (dataloaders/data_tools.py:1615-1685).  The module cannot be imported here (lmdb, fasttext, pymo's plotting
stack), so the two static methods are lifted out of the file with `ast` and compiled as they stand from /root/reference at run time; nothing
of them is stored.  Inputs are seeded (the tests regenerate them); stored: the distances.
    python tests/golden/make_frechet_golden.py
"""
import ast
import os

import numpy as np
from scipy import linalg

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def cases():
    """name -> (samples_A, samples_B), float64, from seeds alone."""
    r = np.random.RandomState(7)
    mix = r.randn(24, 24)
    a = r.randn(300, 24) @ mix
    b = (r.randn(280, 24) * 1.1) @ mix + 0.3
    out = {"gaussians_24d": (a, b), "same_set": (a, a.copy()),
           "one_dim": (r.randn(50, 1), r.randn(60, 1) + 2.0),
           "rank_deficient": (r.randn(10, 16), r.randn(12, 16)),                 # fewer samples than dimensions: singular covariances
           "latent_like_240d": (r.randn(512, 240) * 0.2, r.randn(512, 240) * 0.2 + 0.01)}
    return out


def main():
    path = os.path.join(REF, "dataloaders", "data_tools.py")
    tree = ast.parse(open(path).read(), filename=path)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "FIDCalculator")
    cls.body = [f for f in cls.body if isinstance(f, ast.FunctionDef) and f.name in ("frechet_distance", "calculate_frechet_distance")]
    assert len(cls.body) == 2
    cls.bases = []
    ns = {"np": np, "linalg": linalg}
    exec(compile(ast.Module(body=[cls], type_ignores=[]), path, "exec"), ns)
    fid = ns["FIDCalculator"]
    out = {}
    for name, (a, b) in cases().items():
        out[name] = np.float64(fid.frechet_distance(a, b))
        print(name, out[name])
    np.savez(os.path.join(HERE, "frechet_reference.npz"), **out)


if __name__ == "__main__":
    main()
