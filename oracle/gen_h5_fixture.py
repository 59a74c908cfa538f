"""Test-infrastructure tooling: writes tests/golden/h5lite_fixture.h5 (+ .npz with the same arrays) with the REAL HDF5 library
(h5py 3.3 / HDF5 1.10.6 under /opt/conda in the build container; the product image's python has no h5py) in the group / dataset
shapes keras' H5IOStore produces for `.weights.h5` (<path>/vars/<i> datasets, default h5py settings), so that the pure-Python
reader tensorflowasr_amd/h5lite.py is validated against genuine library output.

    /opt/conda/bin/python3.9 oracle/gen_h5_fixture.py
"""
import os

import h5py
import numpy as np

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    rng = np.random.default_rng(7)
    arrays = {}

    def put(f, path, arr, **kw):
        g = f.require_group(os.path.dirname(path))
        if kw:
            g.create_dataset(os.path.basename(path), data=arr, **kw)
        else:
            g[os.path.basename(path)] = arr  # what keras' H5Entry.__setitem__ does
        arrays[path] = np.asarray(arr)

    p = os.path.join(OUT, "h5lite_fixture.h5")
    with h5py.File(p, "w") as f:
        f.create_group("vars")  # keras: the model's own (empty) variable group
        # 20 sibling groups: forces several symbol nodes / B-tree entries in one group
        for i in range(20):
            base = f"layers/conformer_encoder/conformer_blocks/conformer_block{'' if i == 0 else '_' + str(i)}"
            put(f, base + "/ffm1/dense_1/vars/0", rng.standard_normal((6, 8)).astype(np.float32))
            put(f, base + "/ffm1/dense_1/vars/1", rng.standard_normal(8).astype(np.float32))
            put(f, base + "/convm/dw_norm/vars/0", rng.standard_normal(6).astype(np.float32))
        put(f, "layers/transducer_prediction/lstm/cell/vars/0", rng.standard_normal((5, 12)).astype(np.float32))
        put(f, "layers/misc/f64", rng.standard_normal((3, 2, 2)))
        put(f, "layers/misc/i32", rng.integers(-5, 5, (4, 3)).astype(np.int32))
        put(f, "layers/misc/u8", rng.integers(0, 255, 17).astype(np.uint8))
        put(f, "layers/misc/scalar", np.float32(3.25))
        put(f, "layers/misc/big_endian", rng.standard_normal(9).astype(">f4"))
        put(f, "layers/misc/chunked", rng.standard_normal((10, 7)).astype(np.float32), chunks=(4, 3))
        put(f, "layers/misc/compact", rng.standard_normal(5).astype(np.float32))
        put(f, "optimizer/vars/0", np.int64(1234))
    np.savez_compressed(os.path.join(OUT, "h5lite_fixture.npz"), **{k.replace("/", "|"): v for k, v in arrays.items()})
    print(p, os.path.getsize(p), "bytes,", len(arrays), "datasets")


if __name__ == "__main__":
    main()
