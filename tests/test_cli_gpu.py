"""The C++ `trace` CLI end to end on the GPU: BASELINE config[0] (built-in scene, shadow-trace,
--visualise normal, 512x512) and a small path-trace; its EXR output is read back and compared with the
oracle's AOVs bit for bit."""
import struct
import subprocess
from pathlib import Path

import numpy as np
import pytest

import ipu_ray_lib_amd as irl
import oracle_lib as ol

ROOT = Path(__file__).resolve().parent.parent
TRACE = ROOT / "ipu_ray_lib_amd" / "trace"


def read_exr_bgr(path: Path):
    """Reader for the uncompressed scanline EXR files the CLI writes (channels B, G, R float32)."""
    b = path.read_bytes()
    assert struct.unpack_from("<I", b, 0)[0] == 20000630
    p = 8
    attrs = {}
    while b[p] != 0:
        e = b.index(0, p); name = b[p:e].decode(); p = e + 1
        e = b.index(0, p); typ = b[p:e].decode(); p = e + 1
        size = struct.unpack_from("<i", b, p)[0]; p += 4
        attrs[name] = (typ, b[p:p + size]); p += size
    p += 1
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    assert attrs["compression"][1] == b"\x00"
    offs = struct.unpack_from(f"<{h}Q", b, p)
    img = np.zeros((h, w, 3), np.float32)
    for y in range(h):
        q = offs[y]
        yy, size = struct.unpack_from("<ii", b, q); q += 8
        row = np.frombuffer(b, dtype="<f4", count=3 * w, offset=q).reshape(3, w)
        img[yy] = row.T
    return img


@pytest.mark.gpu
def test_cli_config0_shadow_trace_normals(tmp_path):
    assert TRACE.exists(), "trace CLI is not built (python -c 'import __graft_entry__ as g; g.build()')"
    prefix = tmp_path / "cfg0"
    r = subprocess.run([str(TRACE), "--scene", "box", "--render-mode", "shadow-trace", "--visualise", "normal", "-w", "512", "-h", "512",
                        "--mesh-file", str(irl.DEFAULT_MESH), "-o", str(prefix)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "GPU rays per second" in r.stderr
    img = read_exr_bgr(Path(str(prefix) + "_normal_gpu.exr"))
    s = irl.HostScene.builtin("box"); s.desc.set_image(512, 512); s.desc.path_trace = 0
    want = s.init_ray_stream(); ol.shadow_trace(s.desc, want, 16)
    hit = (want["h"]["geomID"] != irl.INVALID_GEOM).reshape(512, 512)
    n = want["h"]["normal"]
    ref = np.stack([n["z"], n["y"], n["x"]], -1).reshape(512, 512, 3) * hit[..., None]   # cv::Vec3f(n.z, n.y, n.x)
    assert np.array_equal(img.view(np.uint32), np.ascontiguousarray(ref, dtype=np.float32).view(np.uint32))
    # PFM twin of the same image exists
    assert Path(str(prefix) + "_normal_gpu.pfm").stat().st_size > 512 * 512 * 12


@pytest.mark.gpu
def test_cli_path_trace_rgb_and_flag_validation(tmp_path):
    prefix = tmp_path / "pt"
    r = subprocess.run([str(TRACE), "--scene", "box", "-w", "96", "-h", "64", "--samples", "8", "--seed", "77", "--crop", "40x30+10+20",
                        "--mesh-file", str(irl.DEFAULT_MESH), "-o", str(prefix), "--log-level", "debug"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    img = read_exr_bgr(Path(str(prefix) + "_rgb_gpu.exr"))
    s = irl.HostScene.builtin("box"); d = s.desc
    d.set_image(96, 64, (40, 30, 10, 20)); d.samples_per_pixel = 8; d.rng_seed = 77
    want = s.init_ray_stream(); ol.path_trace_pixel_rng(d, want, 16)
    irl.host_lib().mi_scale_rgb(want.ctypes.data, want.size, 1.0 / 8)
    ref = np.zeros((64, 96, 3), np.float32)
    rows = want["u"].astype(int); cols = want["v"].astype(int)
    ref[rows, cols] = np.stack([want["rgb"]["z"], want["rgb"]["y"], want["rgb"]["x"]], -1)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    bad = subprocess.run([str(TRACE), "--visualise", "normal"], capture_output=True, text=True)   # path-trace needs visualise=rgb
    assert bad.returncode != 0 and "not advised" in bad.stderr


@pytest.mark.gpu
def test_cli_nif_hdri_from_keras_h5(tmp_path):
    """--nif-hdri <assets.extra>: the CLI loads nif_metadata.txt + converted.hdf5 (committed tiny fixture) and
    the NIF lights the open 'spheres' scene, as in the reference's notebook recipe; rgb equals the in-process
    render with the same model (same library, same per-pixel RNG) bit for bit."""
    golden = ROOT / "tests" / "golden" / "nif_tiny"
    if not (irl.PKG_DIR / "libmi_nif_h5.so").exists():
        pytest.skip("HDF5 plugin not built")
    prefix = tmp_path / "nif"
    r = subprocess.run([str(TRACE), "--scene", "spheres", "-w", "64", "-h", "48", "--samples", "4", "--nif-hdri", str(golden),
                        "--hdri-rotation", "30", "-o", str(prefix)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Loaded NIF model 'nif_tiny'" in r.stderr
    img = read_exr_bgr(Path(str(prefix) + "_rgb_gpu.exr"))
    s = irl.HostScene.builtin("spheres"); d = s.desc
    d.set_image(64, 48); d.samples_per_pixel = 4
    dev = irl.IpuScene(d)
    assert dev.loadNifModel(golden)
    dev.setHdriRotation(30.0)
    rays = s.init_ray_stream(); dev.run(rays, irl.MODE_PATH_TRACE)
    irl.host_lib().mi_scale_rgb(rays.ctypes.data, rays.size, 1.0 / 4)
    ref = np.stack([rays["rgb"]["z"], rays["rgb"]["y"], rays["rgb"]["x"]], -1).reshape(48, 64, 3)
    assert np.array_equal(img.view(np.uint32), np.ascontiguousarray(ref).view(np.uint32))
    assert ref.sum() > 0          # the environment is the only light in this scene
    missing = subprocess.run([str(TRACE), "--scene", "spheres", "-w", "16", "-h", "16", "--samples", "1", "--nif-hdri", str(tmp_path),
                              "-o", str(prefix)], capture_output=True, text=True, timeout=300)
    assert missing.returncode == 0 and "Could not load NIF model" in missing.stderr   # logged, render goes on (trace.cpp:309-312)


def test_cli_rejects_bad_flags_without_gpu():
    if not TRACE.exists():
        pytest.skip("trace CLI not built")
    for argv, msg in ((["--visualise", "bogus"], "visualise"), (["--render-mode", "x"], "render-mode"), (["--load-normals"], "load-normals"),
                      (["--log-level", "loud"], "log-level"), (["--nope"], "unrecognised")):
        r = subprocess.run([str(TRACE)] + argv, capture_output=True, text=True)
        assert r.returncode != 0 and msg in r.stderr


def _render_exr(tmp_path, tag, *extra):
    prefix = tmp_path / tag
    r = subprocess.run([str(TRACE), "--scene", "box", "-w", "200", "-h", "144", "--samples", "12", "--mesh-file", str(irl.DEFAULT_MESH),
                        "-o", str(prefix), *extra], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return Path(str(prefix) + "_rgb_gpu.exr").read_bytes(), r.stderr


@pytest.mark.gpu
def test_cli_replicas_give_the_byte_identical_exr(tmp_path):
    """`trace --replicas R` (the C++ multi-GPU path of mi::IpuScene: bands dealt to R scene replicas, one RCCL gather;
    on a one-GPU box the replicas share GPU 0 and the RCCL peers are the root's own rank) must write the same EXR,
    byte for byte, as the single-scene render; so must `--gpus N` on a box that has N GPUs."""
    import torch
    one, _ = _render_exr(tmp_path, "one")
    two, log = _render_exr(tmp_path, "two", "--replicas", "2", "--gather", "rccl")
    assert two == one and "1 RCCL send/recv pairs" in log, log
    three, log = _render_exr(tmp_path, "three", "--replicas", "3", "--gather", "copy", "--ipu-ray-callback")
    # with the callback the stream goes through in batches of 8 640 rays (1440 tiles x 6 workers x 1 ray, src/IpuScene.cpp:360-361),
    # each dealt, gathered and brought home on its own: 28 800 rays = 3 batches of three 4 096-ray bands + one of one band
    assert three == one and "10 bands dealt" in log and "6 peer copies" in log, log
    n_dev = torch.cuda.device_count()
    if n_dev >= 2:
        many, log = _render_exr(tmp_path, "many", "--gpus", str(n_dev))
        assert many == one and f"{n_dev - 1} RCCL send/recv pairs" in log, log


@pytest.mark.gpu
def test_cli_replicas_with_nif_give_the_byte_identical_exr(tmp_path):
    """`trace --replicas 2 --nif-hdri <assets.extra>`: mi::IpuScene::configure sets the NIF model on EVERY replica of the
    group (the reference streams the weights to every replica, src/IpuScene.cpp:535), each replica runs the trace -> uv ->
    MLP -> env-add loop on its bands, one RCCL group call gathers them: the EXR must equal, byte for byte, the one the
    single-scene render writes (a ray's MLP result does not depend on which rays share its tile)."""
    golden = ROOT / "tests" / "golden" / "nif_tiny"
    if not (irl.PKG_DIR / "libmi_nif_h5.so").exists():
        pytest.skip("HDF5 plugin not built")

    def render(tag, *extra):
        prefix = tmp_path / tag
        r = subprocess.run([str(TRACE), "--scene", "spheres", "-w", "200", "-h", "144", "--samples", "12", "--nif-hdri", str(golden),
                            "--hdri-rotation", "30", "-o", str(prefix), *extra], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert "Loaded NIF model 'nif_tiny'" in r.stderr
        return Path(str(prefix) + "_rgb_gpu.exr").read_bytes(), r.stderr

    one, _ = render("one")
    two, log = render("two", "--replicas", "2", "--gather", "rccl")
    assert two == one and "1 RCCL send/recv pairs" in log, log
    three, log = render("three", "--replicas", "3", "--gather", "copy", "--ipu-ray-callback")
    assert three == one and "peer copies" in log, log
    img = read_exr_bgr(tmp_path / "two_rgb_gpu.exr")
    assert img.sum() > 0          # the environment is the only light in this scene
