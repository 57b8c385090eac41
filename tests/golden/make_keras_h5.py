#!/opt/conda/bin/python3.9
"""Writes tests/golden/keras_model_k8.h5 + keras_model_k8_expected.json (shape / crc32 / sum of every array written).

Run in the build container with the interpreter that has h5py (a REAL HDF5 library; the
product's reader is pure Python):

    /opt/conda/bin/python3.9 tests/golden/make_keras_h5.py

The file has exactly the structure tf.keras 2.4 / TF 2.2 `model.save(path.h5)` produces for
the reference's create_nn model (training_pipeline.py:59-114,186-191; h5py default settings:
libver 'earliest', contiguous float32 datasets, fixed-length string array attributes
`layer_names` / `weight_names`, variable-length string attributes for the configs):

    /                      attrs keras_version, backend, model_config, training_config
    /model_weights         attrs layer_names, backend, keras_version
    /model_weights/<layer> attrs weight_names; datasets <layer>/kernel:0, <layer>/bias:0, ...
    /optimizer_weights     Adam slots (ignored by the importer)

with NUM_KERNELS = 8 so that the fixture stays small.  The layer names carry the offset a
second create_nn call in the same Keras session gives them (conv2d_10 ...), which the importer
must handle.  Keras and TensorFlow themselves are not available here: the weight VALUES are
seeded random numbers, not a trained model; what this fixture pins is the container format and
the layout conventions (Conv2D kernels H,W,I,O; Dense kernels in,out; BatchNormalization gamma,
beta, moving_mean, moving_variance).
"""
import json
import os

import sys

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "checkers-mcts_amd"))
import keras_h5 as product_h5            # numpy-only at import time: the config generators of the product's exporter
K = 8
OFFSET = 10          # layer-name counter offset of a second model built in the same session


def q(a):
    """float32 values on a 2^-10 grid: the fixture compresses well in git without changing what it tests."""
    return (np.round(np.asarray(a) * 1024.0) / 1024.0).astype(np.float32)


def suffix(i):
    return "" if i == 0 else "_%d" % i


def main():
    rng = np.random.RandomState(20260929)
    layers = []      # (name, [(weight name, array)])

    def conv(i, kh, cin, cout):
        name = "conv2d" + suffix(i + OFFSET)
        return (name, [(name + "/kernel:0", q(rng.randn(kh, kh, cin, cout) * 0.2)),
                       (name + "/bias:0", (rng.randn(cout) * 0.1).astype(np.float32))])

    def bn(i, c):
        name = "batch_normalization" + suffix(i + OFFSET + 1)
        return (name, [(name + "/gamma:0", (1 + 0.2 * rng.randn(c)).astype(np.float32)),
                       (name + "/beta:0", (0.1 * rng.randn(c)).astype(np.float32)),
                       (name + "/moving_mean:0", (0.1 * rng.randn(c)).astype(np.float32)),
                       (name + "/moving_variance:0", (0.5 + rng.rand(c)).astype(np.float32))])

    def dense(name, cin, cout):
        return (name, [(name + "/kernel:0", q(rng.randn(cin, cout) * 0.1)),
                       (name + "/bias:0", (rng.randn(cout) * 0.1).astype(np.float32))])

    # model.layers order of the functional model (topological, as Keras lists them)
    layers.append(("input_2", []))
    for i in range(7):
        layers.append(conv(i, 3, 14 if i == 0 else K, K))
        layers.append(bn(i, K))
    layers.append(conv(7, 3, K, K)); layers.append(conv(9, 1, K, 1))          # policy_conv1, value_conv1
    layers.append(bn(7, K)); layers.append(bn(9, 1))
    layers.append(conv(8, 1, K, 8)); layers.append(("flatten_3", []))          # policy_conv2, value flatten
    layers.append(bn(8, 8)); layers.append(dense("dense_1", 64, 64))
    layers.append(("flatten_2", [])); layers.append(bn(10, 64))
    layers.append(dense("policy_head", 512, 512)); layers.append(dense("value_head", 64, 1))

    path = os.path.join(HERE, "keras_model_k8.h5")
    with h5py.File(path, "w") as f:
        f.attrs["keras_version"] = "2.3.0-tf"                # python str -> variable-length string (global heap), as h5py stores it
        f.attrs["backend"] = "tensorflow"
        # tf.keras 2.2 encodes the two configs to utf-8 bytes (hdf5_format.save_model_to_hdf5): fixed-length string attributes of
        # their full size -- 17 KB for this model -- which do not fit the root object header's first chunk and go to a
        # continuation block: the path a genuine model.save() file takes through the reader
        f.attrs["model_config"] = json.dumps(product_h5.keras_model_config(K, 0.001, 0.001)).encode("utf8")
        f.attrs["training_config"] = json.dumps(product_h5.keras_training_config(1.0, 1.0)).encode("utf8")
        g = f.create_group("model_weights")
        g.attrs["layer_names"] = np.array([n.encode("utf8") for n, _ in layers])     # fixed-length strings, as h5py 2.10 wrote lists of bytes
        g.attrs["backend"] = "tensorflow".encode("utf8")
        g.attrs["keras_version"] = "2.4.0".encode("utf8")
        for name, weights in layers:
            lg = g.create_group(name)
            lg.attrs["weight_names"] = np.array([w.encode("utf8") for w, _ in weights], dtype="S64") if weights else np.zeros((0,), "S1")
            for wname, val in weights:
                d = lg.create_dataset(wname, val.shape, dtype=val.dtype)
                d[:] = val
        og = f.create_group("optimizer_weights")
        og.attrs["weight_names"] = [b"Adam/iter:0"]                                      # h5py 3: variable-length strings
        og.create_dataset("Adam/iter:0", data=np.int64(1234))
    import zlib
    expected = {}
    for name, weights in layers:
        for wname, val in weights:
            expected[wname] = dict(shape=list(val.shape), crc32=zlib.crc32(np.ascontiguousarray(val).tobytes()),
                                   sum=float(val.astype(np.float64).sum()), first=float(val.reshape(-1)[0]))
    with open(os.path.join(HERE, "keras_model_k8_expected.json"), "w") as fp:
        json.dump(dict(layer_order=[n for n, _ in layers], arrays=expected), fp, indent=0, sort_keys=True)
    print(path, os.path.getsize(path), "bytes;", len(expected), "arrays")


if __name__ == "__main__":
    main()
