"""Fixture container: a list of dict cases (numpy arrays / ints) <-> one compressed .npz."""
from __future__ import annotations

import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def save_cases(name: str, cases: list[dict]) -> str:
    flat = {"__n__": np.int64(len(cases))}
    for i, c in enumerate(cases):
        for k, v in c.items():
            flat[f"c{i}__{k}"] = np.asarray(v)
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    np.savez_compressed(path, **flat)
    return path


def load_cases(name: str) -> list[dict]:
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    n = int(z["__n__"])
    cases: list[dict] = [dict() for _ in range(n)]
    for key in z.files:
        if key == "__n__":
            continue
        idx, k = key.split("__", 1)
        v = z[key]
        cases[int(idx[1:])][k] = v if v.ndim else v.item()
    return cases
