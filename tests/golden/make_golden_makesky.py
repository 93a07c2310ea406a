#!/usr/bin/env python3
"""Golden vectors for the cora-makesky driver row: outputs of the reference's own
``FreqState`` (cora/scripts/makesky.py:44-92) for every channelisation mode, obtained by importing the
reference in this container (it needs only click + numpy).  Commits data only: tests/golden/makesky_vectors.npz.

    python tests/golden/make_golden_makesky.py [/root/reference]
"""
import importlib.util
import json
import os
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

CASES = {
    "default": {},
    "centre_400_800_8": {"freq": (400.0, 800.0, 8), "freq_mode": "centre"},
    "nyquist_400_800_9": {"freq": (400.0, 800.0, 9), "freq_mode": "centre_nyquist"},
    "edge_400_800_8": {"freq": (400.0, 800.0, 8), "freq_mode": "edge"},
    "chime_descending_bin4": {"freq": (800.0, 400.0, 1024), "freq_mode": "centre", "channel_bin": 4},
    "edge_range": {"freq": (400.0, 800.0, 32), "freq_mode": "edge", "channel_range": (4, 12)},
    "centre_list_bin2": {"freq": (800.0, 400.0, 64), "freq_mode": "centre", "channel_bin": 2, "channel_list": [0, 3, 31]},
}


def main():
    spec = importlib.util.spec_from_file_location("ref_makesky", os.path.join(REF, "cora", "scripts", "makesky.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = {"cases_json": np.array(json.dumps(CASES))}
    for name, kw in CASES.items():
        fs = mod.FreqState()
        for k, v in kw.items():
            setattr(fs, k, v)
        g[name + "__frequencies"] = np.asarray(fs.frequencies, dtype=np.float64)
        g[name + "__freq_width"] = np.asarray(fs.freq_width, dtype=np.float64)
    path = os.path.join(OUT, "makesky_vectors.npz")
    np.savez_compressed(path, **g)
    print("wrote", path, len(g), "arrays")


if __name__ == "__main__":
    main()
