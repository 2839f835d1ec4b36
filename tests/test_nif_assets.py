"""NIF asset loading: nif_metadata.txt + Keras-H5 weights, as IpuScene::loadNifModel reads
them (src/IpuScene.cpp:174-187, src/neural_networks/NifMetaData.cpp:11-71, src/keras/Hdf5Model.cpp:8-133).

The HDF5 fixture tests/golden/nif_tiny/converted.hdf5 is written by tests/golden/make_nif_h5_fixture.c with
libhdf5 itself (Keras layout: model_config attribute, /model_weights/<l>/<l>/kernel:0, binary16 and binary32
datasets); its weight values follow a closed formula that is recomputed here."""
import ctypes as C
import json
import shutil
import struct
from pathlib import Path

import numpy as np
import pytest

import ipu_ray_lib_amd as irl

GOLDEN = Path(__file__).resolve().parent / "golden" / "nif_tiny"
needs_h5 = pytest.mark.skipif(not (irl.PKG_DIR / "libmi_nif_h5.so").exists(), reason="HDF5 plugin not built (no libhdf5 on this machine)")


def _expected_layer(l, rows, cols, has_bias):
    i = np.arange(rows * cols)
    k = ((((i * 7 + l * 13) % 61) - 30) / 256.0).astype(np.float32).reshape(rows, cols)
    j = np.arange(cols)
    b = ((((j * 5 + l * 3) % 17) - 8) / 64.0).astype(np.float32) if has_bias else None
    return k, b


@needs_h5
def test_keras_h5_fixture_loads_exactly():
    a = irl.NifAssets(GOLDEN)
    assert a.source.endswith("converted.hdf5") and a.name == "nif_tiny"
    assert a.embedding_dimension == 4 and a.hidden_size == 32
    assert a.max_value == 2.5 and a.log_tonemap and a.weights_are_half
    # eps folded into the mean in float arithmetic (NifMetaData.cpp:48-53)
    want_mean = np.array([-1.25, -1.0, -0.75], np.float32) - np.float32(1e-8)
    assert a.mean.tobytes() == want_mean.tobytes()
    shapes = [(16, 32, True), (32, 32, True), (48, 32, False), (32, 3, True)]
    assert a.relu == [True, True, True, False]                  # "linear" -> none (NifModel.cpp:75-77)
    for l, (r, c, hb) in enumerate(shapes):
        k, b = _expected_layer(l, r, c, hb)
        assert a.kernels[l].shape == (r, c) and a.kernels[l].tobytes() == k.tobytes(), l   # binary16 widened exactly
        assert (a.biases[l] is None) == (not hb)
        if hb:
            assert a.biases[l].tobytes() == b.tobytes()


def test_flat_weights_fallback_and_metadata_errors(tmp_path):
    # no converted.hdf5 -> nif_weights.bin
    d = tmp_path / "flat"; d.mkdir()
    shutil.copy(GOLDEN / "nif_metadata.txt", d / "nif_metadata.txt")
    rng = np.random.default_rng(1)
    ks = [rng.standard_normal((16, 8)).astype(np.float32), rng.standard_normal((8, 3)).astype(np.float32)]
    bs = [rng.standard_normal(8).astype(np.float32), None]
    with open(d / "nif_weights.bin", "wb") as f:
        f.write(struct.pack("<I", 2))
        for k, b, relu in zip(ks, bs, (1, 0)):
            f.write(struct.pack("<IIBB", k.shape[0], k.shape[1], relu, 0 if b is None else 1))
            f.write(k.tobytes())
            if b is not None:
                f.write(b.tobytes())
    a = irl.NifAssets(d)
    assert a.source.endswith("nif_weights.bin") and not a.weights_are_half
    assert a.kernels[0].tobytes() == ks[0].tobytes() and a.kernels[1].tobytes() == ks[1].tobytes()
    assert a.biases[0].tobytes() == bs[0].tobytes() and a.biases[1] is None and a.relu == [True, False]

    # truncated dump
    raw = (d / "nif_weights.bin").read_bytes()
    (d / "nif_weights.bin").write_bytes(raw[:-5])
    with pytest.raises(irl.RaylibError, match="truncated"):
        irl.NifAssets(d)

    # metadata problems are reported with the property and the file, like the reference does (:66-70)
    e = tmp_path / "bad"; e.mkdir()
    with pytest.raises(irl.RaylibError, match="nif_metadata.txt"):
        irl.NifAssets(e)
    meta = json.loads((GOLDEN / "nif_metadata.txt").read_text())
    del meta["encode_params"]["max"]
    (e / "nif_metadata.txt").write_text(json.dumps(meta))
    with pytest.raises(irl.RaylibError, match="Error reading property.*max"):
        irl.NifAssets(e)
    (e / "nif_metadata.txt").write_text(json.dumps(json.loads((GOLDEN / "nif_metadata.txt").read_text())))
    with pytest.raises(irl.RaylibError, match="neither converted.hdf5 nor nif_weights.bin"):
        irl.NifAssets(e)


@needs_h5
def test_h5_errors_are_reported(tmp_path):
    d = tmp_path / "broken"; d.mkdir()
    shutil.copy(GOLDEN / "nif_metadata.txt", d / "nif_metadata.txt")
    (d / "converted.hdf5").write_bytes(b"this is not an HDF5 file")
    with pytest.raises(irl.RaylibError, match="cannot open HDF5 file"):
        irl.NifAssets(d)


@pytest.mark.skipif(not Path("/root/reference/assets/nif").is_dir(), reason="reference checkout not present")
def test_reference_metadata_file_parses(tmp_path):
    """The reference ships the metadata (not the weights) of its trained model: parse that very file."""
    src = Path("/root/reference/assets/nif/urban_alley_01_4k_fp16_yuv/assets.extra/nif_metadata.txt")
    d = tmp_path / "ref"; d.mkdir()
    (d / "nif_metadata.txt").write_bytes(src.read_bytes())
    k = np.zeros((48, 3), np.float32)
    with open(d / "nif_weights.bin", "wb") as f:
        f.write(struct.pack("<IIIBB", 1, 48, 3, 0, 0)); f.write(k.tobytes())
    a = irl.NifAssets(d)
    assert a.embedding_dimension == 12 and a.hidden_size == 320 and a.log_tonemap
    assert a.max_value == np.float32(3.4299468994140625)
    want = np.array([-2.3514461517333984, -2.2660605907440186, -1.9648972749710083], np.float32) - np.float32(1e-8)
    assert a.mean.tobytes() == want.tobytes()


def test_host_library_exports_nif_symbols():
    lib = irl.host_lib()
    for name in ("mi_host_nif_load", "mi_host_nif_describe", "mi_host_nif_destroy", "mi_host_nif_last_error"):
        assert hasattr(lib, name)
    h = C.c_void_p()
    assert lib.mi_host_nif_load(None, C.byref(h)) != 0
