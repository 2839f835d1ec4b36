#!/usr/bin/env python3
"""What is the drain at the end of a frame worth, and does handing the expensive pixels out FIRST remove it?

    python tools/order_probe.py [scene=box] [spp=1000]

The work order of the persistent kernel is the order of the stream (any order gives the same image: every pixel owns its RNG streams).
1. cost map: casts per pixel of every 32 x 32 block of the 1440^2 frame, from the library's cast counter over a 16-spp render of the block;
2. the frame as a stream of 8 x 8 tiles in row-major tile order (what the kernel's own tile walk does), tiles = 0;
3. the same tiles, those of the most expensive blocks first.
Prints the three frame times (and casts/s)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import ipu_ray_lib_amd as irl


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "box"
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    E = 1440
    s = irl.HostScene.builtin(name); d = s.desc
    d.set_image(E, E); d.path_trace = 1
    base = s.init_ray_stream()                      # row-major pixels
    st = torch.cuda.current_stream().cuda_stream

    def dev_stream(order):
        r = base[order]
        return torch.from_numpy(r.view(np.uint8).reshape(r.size, irl.TRACE_RESULT.itemsize).copy()).cuda()

    # 1. cost map
    d.samples_per_pixel = 16
    dev = irl.IpuScene(d)
    B = 32; nb = E // B
    rows, cols = np.divmod(np.arange(E * E), E)
    block_of = (rows // B) * nb + cols // B
    cost = np.zeros(nb * nb)
    order_by_block = np.argsort(block_of, kind="stable")
    t = dev_stream(order_by_block)
    per = B * B
    t0 = time.perf_counter()
    for b in range(nb * nb):
        dev.reset_counters()
        dev.run_device(t.data_ptr() + b * per * irl.TRACE_RESULT.itemsize, per, irl.MODE_PATH_TRACE, st)
        torch.cuda.synchronize()
        cost[b] = dev.counters()["casts"] / (per * 16.0)
    print(f"cost map: {nb * nb} blocks in {time.perf_counter() - t0:.1f} s; casts per path min {cost.min():.2f} mean {cost.mean():.2f} max {cost.max():.2f}", flush=True)
    dev.close()

    # 2./3. tile streams
    d.samples_per_pixel = spp
    tile_of = (rows // 8) * (E // 8) + cols // 8
    within = (rows % 8) * 8 + cols % 8
    natural = np.lexsort((within, tile_of))                     # tiles row-major, pixels row-major inside a tile
    tile_cost = cost[block_of]                                  # per pixel: its block's cost
    heavy_first = np.lexsort((within, tile_of, -tile_cost))
    light_first = np.lexsort((within, tile_of, tile_cost))
    cases = [("kernel's own tile walk", np.arange(E * E), 1, 0), ("tiles in row-major order, as a stream", natural, 0, 0),
             ("expensive blocks first", heavy_first, 0, 0), ("cheap blocks first", light_first, 0, 0)]
    # (a fourth form - the most expensive 2 ... 40 % of the blocks at the head of the stream with ALL their work units handed out before any
    # other, through a two-region work index in the kernel - was measured once and removed again: profiles/r05_k1w_drain_ab.txt)
    cases.append(("tiles in row-major order, as a stream", natural, 0, 0))
    for label, order, opt, first in cases:
        dev = irl.IpuScene(d).set_option("tiles", opt)
        t = dev_stream(order)
        dev.run_device(t.data_ptr(), E * E, irl.MODE_PATH_TRACE, st); torch.cuda.synchronize(); dev.reset_counters()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            dev.run_device(t.data_ptr(), E * E, irl.MODE_PATH_TRACE, st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        c = dev.counters()
        print(f"{label:62s} {ms:8.2f} ms   {c['casts'] / 3 / (ms * 1e-3) / 1e9:.3f}e9 casts/s", flush=True)
        dev.close()


if __name__ == "__main__":
    main()
