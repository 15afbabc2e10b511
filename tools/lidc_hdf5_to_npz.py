#!/usr/bin/env python3
"""Write an .npz mirror of data_lidc.hdf5 (needs h5py) for hosts without h5py:

    python tools/lidc_hdf5_to_npz.py data_lidc.hdf5 data_lidc.npz [split ...]      (default split: test)

The mirror holds '<split>/images' [n,128,128] float and '<split>/labels' [n,4,128,128] integer — the layout
`evaluation.TestLIDC` (reference: datasets/lidc.py:177-198) reads; name it as `dataset_path` in params_eval.yml."""
import sys

import numpy as np


def main(argv):
    if len(argv) < 3:
        print(__doc__)
        return 2
    import h5py
    splits = argv[3:] or ["test"]
    out = {}
    with h5py.File(argv[1], "r") as h:
        for sp in splits:
            out[f"{sp}/images"] = np.asarray(h[sp]["images"])
            out[f"{sp}/labels"] = np.asarray(h[sp]["labels"])
    np.savez(argv[2], **out)
    print("wrote", argv[2], {k: v.shape for k, v in out.items()})
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
