"""CPU: the pure-Python HDF5 reader against files written by the REAL HDF5 library (h5py 3.3 / HDF5 1.10.6, oracle/gen_h5_fixture.py
and oracle/npz_to_h5.py, run under /opt/conda in the build container), and the `.weights.h5` -> model-variable resolver on two
path spellings (attribute-walk paths and layer-name paths)."""
import os

import numpy as np
import pytest

from tensorflowasr_amd import checkpoint as C
from tensorflowasr_amd.h5lite import H5Error, H5File


def test_reader_matches_real_library_output(golden_dir):
    z = np.load(os.path.join(golden_dir, "h5lite_fixture.npz"))
    want = {k.replace("|", "/"): z[k] for k in z.files}
    with H5File(os.path.join(golden_dir, "h5lite_fixture.h5")) as f:
        got = f.datasets()
        assert sorted(got) == sorted(want)
        for k, a in want.items():
            assert got[k].shape == a.shape and got[k].dtype == a.dtype.newbyteorder("="), k
            np.testing.assert_array_equal(got[k], a)
        np.testing.assert_array_equal(f["layers/misc/chunked"], want["layers/misc/chunked"])
        with pytest.raises(KeyError):
            f["layers/nope"]


def test_not_hdf5_fails_loudly(tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"PK\x03\x04 not hdf5" * 10)
    with pytest.raises(H5Error):
        H5File(str(p))


@pytest.mark.parametrize("fixture", ["weights_h5_tiny_attrpaths.weights.h5", "weights_h5_tiny_layernames.weights.h5"])
def test_weights_h5_resolver_maps_every_variable(golden_dir, fixture):
    z = np.load(os.path.join(golden_dir, "weights_h5_tiny_expected.npz"))
    expected = {k.replace("|", "/"): z[k] for k in z.files}
    with H5File(os.path.join(golden_dir, fixture)) as f:
        datasets = f.datasets()
    got, unused = C.from_weights_h5(datasets, expected)
    assert sorted(got) == sorted(expected)
    for k, a in expected.items():
        np.testing.assert_array_equal(got[k], a, err_msg=k)
    assert unused == ["optimizer/vars/0"]
    # an ambiguous / incomplete file fails loudly instead of guessing
    broken = {p: a for p, a in datasets.items() if "ffn1" not in p and "dense_1" not in p}
    with pytest.raises(KeyError):
        C.from_weights_h5(broken, expected)
