"""Test-infrastructure tooling: {path with '|' separators: array} .npz -> HDF5 file written by the real HDF5 library (h5py under
/opt/conda in the build container), one dataset per entry, groups created like keras' H5IOStore does (default h5py settings).

    /opt/conda/bin/python3.9 oracle/npz_to_h5.py src.npz dst.h5
"""
import sys

import h5py
import numpy as np

src, dst = sys.argv[1], sys.argv[2]
z = np.load(src)
with h5py.File(dst, "w") as f:
    f.create_group("vars")
    for k in z.files:
        path = k.replace("|", "/")
        g = f.require_group(path.rsplit("/", 1)[0])
        g[path.rsplit("/", 1)[1]] = z[k]
    f.require_group("optimizer/vars")["0"] = np.int64(7)
print(dst, len(z.files), "datasets")
