"""N>1 path on CPU: world_size-2 gloo.  Each rank fills a full-frame film for its ImageBlock tiles
(the oracle stands in for the GPU kernel here: there is no GPU in this tier), the films are summed with
filmreduce.reduce_film and the result must equal the unsharded film bit for bit.  The product's own world > 1
branch (bench.py: set_tiles, the shared film buffer, the reduce) runs in tests/test_gpu_round2.py on the GPU box."""
import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, W, H, out_path):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    import _pkgload, orc
    pkg = _pkgload.load()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sd = pkg.scenes.cornell_c1()
    fs = orc.FlatScene(sd)
    cam = orc.make_camera(sd, W, H)
    prm = orc.render_params(4, sampler=pkg.abi.SAMPLER_LD_KEYED, spp=8, seed=3, n_threads=2)
    film = np.zeros((H, W, 5), dtype=np.float32)
    bs = 32
    tx = (W + bs - 1) // bs
    keys = pkg.filmreduce.tiles_of_rank(W, H, bs, rank, world)
    tiles = sorted(set(((k // W) // bs) * tx + ((k % W) // bs) for k in keys.tolist()))
    assert all(pkg.filmreduce.tile_morton(t % tx, t // tx) % world == rank for t in tiles)
    for t in tiles:
        x0, y0 = (t % tx) * bs, (t // tx) * bs
        part, _ = orc.render(fs.scene, cam, prm, rect=(x0, y0, min(x0 + bs, W), min(y0 + bs, H)))
        film += part
    t = torch.from_numpy(film)
    pkg.filmreduce.reduce_film(t, dst=0)
    if rank == 0:
        np.save(out_path, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_film_reduce_is_exact(tmp_path, mts, orc):
    import torch.multiprocessing as mp
    W, H = 80, 48
    out = str(tmp_path / "film.npy")
    import socket
    with socket.socket() as sk:                  # a free port, not one derived from the pid (parallel test runs collide)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(2, port, W, H, out), nprocs=2, join=True)
    got = np.load(out)
    sd = mts.scenes.cornell_c1()
    fs = orc.FlatScene(sd)
    full, _ = orc.render(fs.scene, orc.make_camera(sd, W, H),
                         orc.render_params(4, sampler=mts.abi.SAMPLER_LD_KEYED, spp=8, seed=3))
    assert np.array_equal(got.view(np.uint32), full.view(np.uint32))
