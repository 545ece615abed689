"""N > 1 path on CPU: world_size-2 gloo.  Each rank takes its contiguous world
shard (scenes.shard_range), renders it -- with the CPU oracle standing in for
the GPU, which this container lacks -- and the slabs are all-gathered
(sharding.gather_slabs).  The gathered tensor must equal a single-process
render of the whole job, for equal and for ragged shards."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from madrona_renderer_amd import scenes, sharding
from tests.conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world_size, port, num_worlds, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from oracle import oracle
    lo, hi = scenes.shard_range(num_worlds, rank, world_size)
    # a rank builds only its own worlds (first_world = global offset) ...
    mine = scenes.synthetic_scene(hi - lo, with_wall=True, first_world=lo)
    # ... which is the same as slicing the whole job's description
    whole = scenes.synthetic_scene(num_worlds, with_wall=True)
    n_inst = whole.worlds[0][0]
    assert mine.instances == whole.instances[lo * n_inst:hi * n_inst]
    assert mine.cameras == whole.cameras[lo:hi]
    assert [w[0::2] for w in mine.worlds] == [w[0::2] for w in whole.shard(rank, world_size).worlds]
    o = oracle.FlatScene(mine).render(num_threads=2)
    counts = [b - a for a, b in sharding.view_ranges(whole.worlds, world_size)]
    rgb = sharding.gather_slabs(torch.from_numpy(o["rgb"]), counts)
    dep = sharding.gather_slabs(torch.from_numpy(o["depth"]), counts)
    # a per-step gather reuses its buffers: the global tensor and (ragged) the padded scratch
    keep = {}
    again = sharding.gather_slabs(torch.from_numpy(o["rgb"]), counts, out=rgb, scratch=keep)
    assert again.data_ptr() == rgb.data_ptr()
    if len(set(counts)) > 1:
        recv = keep["recv"].data_ptr()
        third = sharding.gather_slabs(torch.from_numpy(o["rgb"]), counts, out=again, scratch=keep)
        assert third.data_ptr() == rgb.data_ptr() and keep["recv"].data_ptr() == recv
        with pytest.raises(ValueError):
            sharding.gather_slabs(torch.from_numpy(o["rgb"]), counts + [1])
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == world_size
    if rank == 0:
        np.save(os.path.join(out_dir, "rgb.npy"), rgb.numpy())
        np.save(os.path.join(out_dir, "depth.npy"), dep.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("num_worlds", [8, 7])
def test_two_rank_shard_and_gather_equals_single_process(oracle_mod, tmp_path, num_worlds):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, num_worlds, str(tmp_path)), nprocs=2, join=True)
    whole = scenes.synthetic_scene(num_worlds, with_wall=True)
    ref = oracle_mod.FlatScene(whole).render(num_threads=2)
    assert np.array_equal(np.load(tmp_path / "rgb.npy"), ref["rgb"])
    assert np.array_equal(np.load(tmp_path / "depth.npy"), ref["depth"])


def test_world_size_one_gather_is_a_noop():
    t = torch.arange(12).reshape(3, 4)
    assert sharding.gather_slabs(t) is t


def test_view_ranges_follow_camera_prefix_sums():
    worlds = [(1, 0, 2, 0), (1, 0, 0, 0), (1, 0, 3, 0), (1, 0, 1, 0), (1, 0, 1, 0)]
    assert sharding.view_ranges(worlds, 2) == [(0, 5), (5, 7)]
    assert sharding.view_ranges(worlds, 1) == [(0, 7)]
