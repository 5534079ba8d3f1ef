"""Worker of the two-process gloo tests: the multi-GPU partitioning logic with the CPU oracle standing in
for the render kernel (test infrastructure; the product path never renders on the CPU)."""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

W, H, DEPTH, STEPS = 64, 48, 4, 3


def _setup():
    import oracle_binding as ob

    pta = importlib.import_module("path-tracer_amd")
    wd = pta.work_dispatcher
    scene = pta.scenes.load_fixture_scene("sphere_scene1", os.path.join(HERE, "golden"))
    packed = wd.pack_scene(scene, ob.OracleBvhBuilder(), 8)
    return ob, pta, wd, scene, packed


def render_samples(ob, wd, scene, packed, sample_ids, window=None, first_frame=0):
    """Progressive render of the given global sample indices (frame = first_frame + position in the batch)."""
    img = np.zeros((H, W, 4), np.float32)
    for pos, k in enumerate(sample_ids):
        win = window or wd.PTRenderWindow(render_width=W, render_height=H, max_ray_depth=DEPTH)
        win.frame = first_frame + pos
        ob.render(packed, wd.push_constant_bytes(scene.camera, win, wd.sample_time(k)), img,
                  win.work_group_width, win.work_group_height, nthreads=1)
    return img


def worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist

    mg = importlib.import_module("path-tracer_amd.multi_gpu")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        ob, pta, wd, scene, packed = _setup()
        # partitioning by sample batches: every rank renders its own samples, one reduce at the end
        mine = [mg.sample_index(rank, STEPS, k) for k in range(STEPS)]
        acc = torch.from_numpy(render_samples(ob, wd, scene, packed, mine, first_frame=mg.batch_frame(0)))
        mg.reduce_batch_means(acc, dst=0)
        # partitioning by tile rows: every rank renders its band of one sample, one gather at the end
        base = wd.PTRenderWindow(render_width=W, render_height=H, max_ray_depth=DEPTH)
        band = mg.band_window(base, rank, world)
        tile_img = torch.from_numpy(render_samples(ob, wd, scene, packed, [1], window=band))
        mg.gather_bands(tile_img, base.work_group_height, dst=0)
        if rank == 0:
            np.save(os.path.join(out_dir, "batch_mean.npy"), acc.numpy())
            np.save(os.path.join(out_dir, "tiles.npy"), tile_img.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()
