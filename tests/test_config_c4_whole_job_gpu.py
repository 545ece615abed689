"""BASELINE configs[3] as ONE job: 16384 worlds x 64x64 in one renderer of
eight shards -- the one-Manager form every reference caller uses
(/root/reference/scripts/test.py:112-130: one constructor, one step(); src/mgr.hpp:50 has a
single gpuID, mrx_config.device_ids widens it).  The box has one device, so all eight shards
sit on device 0: each is what rank i of 8 renders on a node.

Every shard's slab is compared with the oracle's render of its world range (all eight), and the
slabs put end to end with a single-shard render of the whole job, byte for byte."""
import ctypes

import numpy as np
import pytest

from madrona_renderer_amd import scenes
from tests.util import assert_parity, make_product, render_oracle

pytestmark = pytest.mark.gpu

WORLDS, SHARDS = 16384, 8


def _slabs(r, n, ids):
    import torch
    r.sync()
    out = {"rgb": [r.rgb_tensor(shard=i).to_torch() for i in range(n)],
           "depth": [r.depth_tensor(shard=i).to_torch() for i in range(n)]}
    if ids:
        out["tri_id"] = [r.visibility_tensor(shard=i).to_torch() for i in range(n)]
    return out


@pytest.fixture(scope="module")
def job(native):
    desc = scenes.synthetic_scene(WORLDS)
    ref = render_oracle(desc)                    # rgb, depth, tri_id of all 16384 views
    return desc, ref


@pytest.mark.parametrize("ids", [False, True], ids=["benchmarked-kernel", "with-visibility-ids"])
def test_configs3_every_shard_against_the_oracle_and_the_whole_against_one_render(native, job, ids, monkeypatch):
    import torch
    desc, ref = job
    # a host thread per shard, as on a node of eight devices (by default shards that share a
    # device share a thread: mrx_api.cpp startShardWorkers)
    monkeypatch.setenv("MRX_SHARD_THREADS", "2")
    monkeypatch.setenv("MADRONA_MI355_VISIBILITY", "1" if ids else "0")
    many = scenes.make_renderer(desc, device_ids=[0] * SHARDS)
    one = scenes.make_renderer(desc)
    assert many.num_shards == SHARDS and one.num_shards == 1
    for _ in range(3):                           # steady state, not only the constructor's frame
        many.step()
    got = _slabs(many, SHARDS, ids)
    for i in range(SHARDS):
        lo, hi = scenes.shard_range(WORLDS, i, SHARDS)
        assert (many.shard_first_world(i), many.shard_first_world(i + 1)) == (lo, hi)
        assert hi - lo == WORLDS // SHARDS
        slab = {"rgb": got["rgb"][i].cpu().numpy(),
                "depth": got["depth"][i].cpu().numpy().reshape(hi - lo, 64, 64)}
        if ids:
            slab["tri_id"] = got["tri_id"][i].cpu().numpy()
        want = {k: v[lo:hi] for k, v in ref.items() if k in slab}
        assert_parity(slab, want)
    # the global tensors = the slabs end to end (what rank-ordered all_gather_into_tensor yields)
    one.step()
    one.sync()
    assert torch.equal(torch.cat(got["rgb"]), one.rgb_tensor().to_torch())
    assert torch.equal(torch.cat(got["depth"]).view(torch.int32), one.depth_tensor().to_torch().view(torch.int32))
    if ids:
        assert torch.equal(torch.cat(got["tri_id"]), one.visibility_tensor().to_torch())
    assert many.bytes_per_step() == one.bytes_per_step() == WORLDS * (64 * 64 * (12 if ids else 8) + 2 * 44 + 28)


def test_serial_and_threaded_multi_device_steps_agree(native, monkeypatch):
    # MRX_SHARD_THREADS=0 launches every shard from the calling thread (the round-3 form), 2 gives
    # every shard a host thread (what a node of eight devices gets by default), the default 1 one
    # thread per device -- here one device, so the calling thread again.  Same bytes; poses written between steps are seen by
    # both (the write is on the shard's stream, the worker's launch is ordered behind it
    # because step() returns only after every shard's launch is enqueued).
    import torch
    desc = scenes.synthetic_scene(1003)          # ragged: 1003 = 8 * 125 + 3
    rs = []
    for threads in ("0", "2", None, "async"):
        monkeypatch.delenv("MRX_SHARD_ASYNC", raising=False)
        if threads is None:
            monkeypatch.delenv("MRX_SHARD_THREADS")
        elif threads == "async":
            # opt-in: step() only posts the render to the device threads; every other call joins them first
            # (here: the tensor getters ahead of the next pose write, sync() at the end)
            monkeypatch.setenv("MRX_SHARD_THREADS", "2")
            monkeypatch.setenv("MRX_SHARD_ASYNC", "1")
        else:
            monkeypatch.setenv("MRX_SHARD_THREADS", threads)
        rs.append(scenes.make_renderer(desc, device_ids=[0] * SHARDS))
    for step in range(4):
        for r in rs:
            for i in range(SHARDS):
                r.instance_position_tensor(shard=i).to_torch()[1::2, 2] += 0.25
            r.step()
    a, b, c, e = (_slabs(r, SHARDS, False) for r in rs)
    for k in a:
        for x, y, z, u in zip(a[k], b[k], c[k], e[k]):
            assert torch.equal(x, y) and torch.equal(x, z) and torch.equal(x, u), k
    moved = scenes.synthetic_scene(1003)
    inst = list(moved.instances)
    for row in range(1, len(inst), 2):
        z = np.float32(inst[row][0][2])
        for _ in range(4):
            z = np.float32(z + np.float32(0.25))
        inst[row] = ((inst[row][0][0], inst[row][0][1], float(z)),) + tuple(inst[row][1:])
    moved.instances = inst
    ref = render_oracle(moved, want_ids=False)
    rgb = torch.cat(a["rgb"]).cpu().numpy()
    depth = torch.cat(a["depth"]).cpu().numpy().reshape(1003, 64, 64)
    assert_parity({"rgb": rgb, "depth": depth}, ref)
    assert rs[1].time_steps_host(20) > 0 and rs[0].time_steps_host(20) > 0 and rs[3].time_steps_host(300) > 0
    # (300 posted steps in a row: more than the device threads enqueue in the meantime -- the count catches up at the join)
    after = _slabs(rs[3], SHARDS, False)
    for k in a:
        for x, u in zip(a[k], after[k]):
            assert torch.equal(x, u), k


def test_mrx_info_keeps_the_abi2_size_and_the_sized_call_gives_the_rest(native):
    # mrx_info_t grew by num_shards; the old entry point must not write past the
    # struct an ABI-2 caller allocated
    lib = native.load_capi()
    r = scenes.make_renderer(scenes.synthetic_scene(6), device_ids=[0, 0, 0])
    h = ctypes.c_void_p(r.native_handle())
    buf = (ctypes.c_uint8 * 96)(*([0xAB] * 96))
    assert lib.mrx_info(h, ctypes.byref(buf)) == 0
    raw = bytes(buf)
    assert raw[72:] == b"\xab" * 24              # nothing beyond the 72 bytes of ABI 2
    assert int.from_bytes(raw[0:4], "little") == 6      # num_worlds
    lib.mrx_info_sized.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    buf2 = (ctypes.c_uint8 * 96)(*([0xAB] * 96))
    assert lib.mrx_info_sized(h, ctypes.byref(buf2), 80) == 0
    raw2 = bytes(buf2)
    assert raw2[:72] == raw[:72] and int.from_bytes(raw2[72:76], "little") == 3 and raw2[80:] == b"\xab" * 16
    assert lib.mrx_info_sized(h, ctypes.byref(buf2), 64) != 0      # smaller than ABI 2: refused
    assert lib.mrx_info_sized(h, ctypes.byref(buf2), 4096) == 0    # a future, larger struct: only what exists


def test_more_devices_than_worlds_is_refused(native):
    with pytest.raises(RuntimeError, match="more devices"):
        scenes.make_renderer(scenes.synthetic_scene(2), device_ids=[0, 0, 0])


def test_shard_threads_survive_bursts_idle_gaps_and_interleaved_calls(native, monkeypatch):
    # the shard workers spin for MRX_SHARD_SPIN_US after a command and then sleep in a futex: bursts of
    # steps, gaps longer than the spin (so that every wake-up path is taken), syncs, timed renders and
    # pose writes in between, with a short spin and with none -- the pictures stay those of one renderer
    import time
    import torch
    desc = scenes.synthetic_scene(257)
    for spin, async_steps in (("50", "0"), ("0", "0"), ("50", "1")):
        monkeypatch.setenv("MRX_SHARD_THREADS", "2")
        monkeypatch.setenv("MRX_SHARD_SPIN_US", spin)
        monkeypatch.setenv("MRX_SHARD_ASYNC", async_steps)
        many = scenes.make_renderer(desc, device_ids=[0] * 5)
        one = scenes.make_renderer(desc)
        streams = [torch.cuda.Stream() for _ in range(5)]
        for i, s in enumerate(streams):
            many.set_stream(s.cuda_stream, shard=i)
        rng = np.random.default_rng(int(spin) + 1 + 7 * int(async_steps))
        pos1 = one.instance_position_tensor().to_torch()
        for burst in range(60):
            dz = float(np.float32(rng.uniform(-0.05, 0.05)))
            for i in range(5):
                with torch.cuda.stream(streams[i]):
                    many.instance_position_tensor(shard=i).to_torch()[1::2, 2] += dz
            pos1[1::2, 2] += dz
            for _ in range(int(rng.integers(1, 40))):
                many.step()
            one.step()
            if burst % 7 == 0:
                many.sync()
                time.sleep(0.002)                    # longer than any spin: the workers go to sleep
            if burst % 11 == 0:
                assert many.time_renders(5) > 0
        got = _slabs(many, 5, False)
        one.sync()
        assert torch.equal(torch.cat(got["rgb"]), one.rgb_tensor().to_torch())
        assert torch.equal(torch.cat(got["depth"]).view(torch.int32), one.depth_tensor().to_torch().view(torch.int32))
        del many, one
