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
