"""Tier-2 statistical parity: the oracle's literal restatement of renderCPU — ONE shared
xoroshiro consumed sequentially, libstdc++ normal_distribution jitter (trace.cpp:236-245) — against the
per-pixel-stream scheme the GPU uses. Different random numbers, same estimator: images must agree like two
independent renders do, which is the reference's own acceptance method (notebook cells 18-19)."""
import numpy as np

import ipu_ray_lib_amd as irl
import oracle_lib as ol


def _rgb(r, spp):
    return np.stack([r["rgb"]["x"], r["rgb"]["y"], r["rgb"]["z"]], 1) / spp


def test_shared_rng_and_pixel_rng_agree_statistically():
    s = irl.HostScene.builtin("box")
    d = s.desc
    d.set_image(40, 40)
    imgs = {}
    for spp in (16, 64):
        d.samples_per_pixel = spp
        a = s.init_ray_stream(); ol.path_trace_pixel_rng(d, a, 8)
        b = s.init_ray_stream(); ol.path_trace_shared_rng(d, b)
        d.rng_seed = 99
        c = s.init_ray_stream(); ol.path_trace_pixel_rng(d, c, 8)      # an independent render of the same scheme
        d.rng_seed = 1442
        imgs[spp] = (_rgb(a, spp), _rgb(b, spp), _rgb(c, spp))
    # Channel means. One 40x40 render's mean has a relative standard deviation of 3-4.6 % at 16 spp and 2-2.5 % at
    # 64 spp (measured over 12 seeds), so single renders only agree to ~2 sigma = 10 %; the means of SIX seeds of
    # each scheme differ by sigma * sqrt(2/6): the bounds below are 3.5 of those.
    for spp, tol in ((16, 0.09), (64, 0.05)):
        d.samples_per_pixel = spp
        pix, shared = [], []
        for seed in range(1, 7):
            d.rng_seed = seed
            r = s.init_ray_stream(); ol.path_trace_pixel_rng(d, r, 8); pix.append(_rgb(r, spp).mean(0))
            r = s.init_ray_stream(); ol.path_trace_shared_rng(d, r); shared.append(_rgb(r, spp).mean(0))
        d.rng_seed = 1442
        assert np.allclose(np.mean(pix, 0), np.mean(shared, 0), rtol=tol), (spp, np.mean(pix, 0), np.mean(shared, 0))
    for spp, (a, b, c) in imgs.items():
        # the cross-scheme MSE is the same size as the MSE between two seeds of one scheme
        mse_ab = np.mean((a - b) ** 2); mse_ac = np.mean((a - c) ** 2)
        assert 0.4 < mse_ab / mse_ac < 2.5, (spp, mse_ab, mse_ac)
    # and it falls like 1/spp (x4 samples -> ~x4 lower; accept x2..x8)
    r = np.mean((imgs[16][0] - imgs[16][1]) ** 2) / np.mean((imgs[64][0] - imgs[64][1]) ** 2)
    assert 2.0 < r < 8.0, r
    # the shared stream really is sequential: its first pixel's jitter comes from the first two draws
    d.samples_per_pixel = 1
    b1 = s.init_ray_stream(); ol.path_trace_shared_rng(d, b1)
    b2 = s.init_ray_stream(); ol.path_trace_shared_rng(d, b2)
    assert b1.tobytes() == b2.tobytes()


def test_segment_length_rule_is_the_same_in_the_product_header_and_the_oracle(tmp_path):
    """The tier-1 stream definition cuts a pixel's samples into segments whose length depends on the sample count
    (DESIGN.md §4): ceil(spp / 16) rounded up to a power of two, within 4 ... 64. The product header (ray_math.h,
    compiled here for the host), the oracle and this restatement must agree for every sample count."""
    import ctypes as C
    import subprocess
    src = tmp_path / "seg.cpp"
    src.write_text('#include <cstdio>\n#include "ray_math.h"\nint main() { for (unsigned s = 0; s <= 5000; ++s) std::printf("%u %u\\n", mi::segment_samples(s), mi::segment_shift(s)); }\n')
    exe = tmp_path / "seg"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", str(ol.ROOT / "ipu_ray_lib_amd" / "csrc"), "-I", str(ol.ROOT / "include"), "-o", str(exe), str(src)], check=True)
    rows = [tuple(map(int, line.split())) for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines()]
    lib = ol.lib()
    lib.o_segment_samples.restype = C.c_uint32; lib.o_segment_samples.argtypes = [C.c_uint32]
    for spp, (length, shift) in enumerate(rows):
        want = 4
        while want < -(-spp // 16) and want < 64:
            want *= 2
        assert length == want == lib.o_segment_samples(spp) and length == 1 << shift, (spp, length, shift, want)
    assert [rows[s][0] for s in (1, 64, 65, 128, 129, 256, 257, 512, 513, 1000, 4000)] == [4, 4, 8, 8, 16, 16, 32, 32, 64, 64, 64]
