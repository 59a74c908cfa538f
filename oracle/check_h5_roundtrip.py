"""Test-infrastructure tooling: check a file written by tensorflowasr_amd.h5lite.write_h5 with the REAL HDF5 library (h5py under
/opt/conda in the build container): every dataset of the companion .npz must read back identically, and the library must be able
to extend the file (create groups / datasets in it), i.e. heaps, B-trees and symbol nodes are well-formed.

    /opt/conda/bin/python3.9 oracle/check_h5_roundtrip.py file.h5 file.npz
"""
import shutil
import sys

import h5py
import numpy as np

path, npz = sys.argv[1], sys.argv[2]
z = np.load(npz)
want = {k.replace("|", "/"): z[k] for k in z.files}
with h5py.File(path, "r") as f:
    names = []
    f.visititems(lambda n, o: names.append(n) if isinstance(o, h5py.Dataset) else None)
    assert sorted(names) == sorted(want), (sorted(set(names) ^ set(want))[:10])
    for k, a in want.items():
        got = f[k][()]
        assert got.dtype == a.dtype and got.shape == a.shape, (k, got.dtype, a.dtype, got.shape, a.shape)
        np.testing.assert_array_equal(got, a)
tmp = path + ".extend"
shutil.copy(path, tmp)
with h5py.File(tmp, "a") as f:
    g = f[sorted(want)[0].rsplit("/", 1)[0]] if "/" in sorted(want)[0] else f
    g["added_by_libhdf5"] = np.arange(5)
    f.create_group("new_group_by_libhdf5")["x"] = np.ones(3, np.float32)
with h5py.File(tmp, "r") as f:
    assert f["new_group_by_libhdf5/x"][()].sum() == 3
print("ok:", len(want), "datasets read back identically by libhdf5", h5py.version.hdf5_version, "and the file can be extended")
