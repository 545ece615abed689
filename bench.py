#!/usr/bin/env python3
"""Headline benchmark: rendered views/sec of the batch-render hot path
(Manager::step) on N MI355X, one process per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
         --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one render of every view of the rank's worlds into the contiguous
[views,H,W,4] RGBA8 + [views,H,W,1] f32 depth tensors; worlds shard across
ranks with no data-path collective (weak scaling: --worlds per GPU).  Rank 0
prints ONE JSON line.

Other workloads of BASELINE.json are selected with flags (the default is the
north-star configuration, 4096 worlds x 64x64 cube+plane):
  --worlds 1024                                    configs[1]
  --worlds 4096 --width 128 --height 128 --wall    configs[2]
  --worlds 2048 [--first-world 14336]              one rank's shard of configs[3]
  --worlds 4096 --width 256 --height 256 --textured --mode Raytracer [--variant 2]   configs[4]
  --cubes 40 --worlds 1024                         many-instance worlds (BVH path)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
# untimed renders before the warm-up (clocks leave idle only under load); reported as `settle_s`
SETTLE_S = float(os.environ.get("MRX_BENCH_SETTLE_S", "0.25"))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--worlds", type=int, default=4096, help="worlds per GPU")
    ap.add_argument("--width", type=int, default=64)
    ap.add_argument("--height", type=int, default=64)
    ap.add_argument("--wall", action="store_true", help="add wall_render.obj (config C3)")
    ap.add_argument("--textured", action="store_true", help="cube textured with cube.png (config C5)")
    ap.add_argument("--mode", default="Rasterizer", choices=["Rasterizer", "Raytracer"])
    ap.add_argument("--cubes", type=int, default=0,
                    help="many-instance worlds: this many cubes + plane per world (tests/meshes.py)")
    ap.add_argument("--variant", type=int, default=0,
                    help="mrx_config.kernel_variant: 0 default dispatch, 2 BVH path, 3 raster kernels")
    ap.add_argument("--first-world", type=int, default=None,
                    help="global id of this rank's first world (default: rank * worlds)")
    ap.add_argument("--gather", action="store_true",
                    help="also time an RCCL all-gather of the output slabs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the secondary 1024-world measurement")
    ap.add_argument("--cpu-views", type=int, default=4096)
    ap.add_argument("--cpu-threads", type=int, default=16)
    return ap.parse_args()


def timed_steps(r, steps, barrier):
    """K back-to-back step() calls bracketed by a barrier + device sync on both
    sides.  Every rank starts its clock when it leaves the opening barrier and
    stops it when its own K steps have completed on the device; the closing
    barrier follows (the caller takes the MAX over ranks: the job is done when
    the slowest rank is).  The collective itself is therefore not inside the
    timed region -- at K = 20 a 50 us RCCL barrier would be a tenth of it.
    Returns (wall seconds, device ms between HIP events on the launch stream)."""
    import torch
    barrier()
    torch.cuda.synchronize()
    r.mark(0)                      # (the opening event is recorded ahead of the clock: two event records
    t0 = time.perf_counter()       # inside a K = 20 region cost 0.4 us per step, scripts/k20_wall_probe.py)
    for _ in range(steps):
        r.step()
    r.mark(1)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    barrier()
    return t1 - t0, r.elapsed_ms()


def workload_tag(a, n_gpus):
    """Key of profiles/pmc_latest.json: the full configuration, so that a
    counter figure is only ever attached to the run it was measured on."""
    tag = "%dx%dx%d" % (a.worlds, a.width, a.height)
    if a.wall:
        tag += "+wall"
    if a.textured:
        tag += "+tex"
    if a.mode != "Rasterizer":
        tag += "+rt"
    if a.cubes:
        tag += "+cubes%d" % a.cubes
    if a.variant:
        tag += "+variant%d" % a.variant
    if n_gpus != 1:
        tag += "+gpus%d" % n_gpus
    return tag


def pmc_traffic(tag):
    """(HBM bytes per launch, where the figure comes from) from the committed
    rocprofv3 --pmc summaries of this same command (profiles/pmc_latest.json),
    or (None, None): the counters are not collected by this run."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except Exception:
        return None, None
    ent = table.get(tag)
    if ent is None:
        return None, None
    if isinstance(ent, dict):
        return ent.get("bytes"), "static: profiles/pmc_latest.json <- %s" % ent.get("source", "?")
    return ent, "static: profiles/pmc_latest.json"


def make_scene(a, first_world, worlds=None, scenes=None):
    worlds = a.worlds if worlds is None else worlds
    if a.cubes:
        from tests import meshes
        return meshes.cube_field(worlds, a.cubes, width=a.width, height=a.height, mode=a.mode,
                                 textured=a.textured, first_world=first_world)
    return scenes.synthetic_scene(worlds, width=a.width, height=a.height, with_wall=a.wall,
                                  textured=a.textured, render_mode=a.mode, first_world=first_world)


def main():
    a = parse()
    import torch
    from madrona_renderer_amd import scenes

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # MRX_BENCH_REHEARSAL=1: every rank on cuda:0 with gloo for the barriers --
    # rehearses the N>1 control flow on a one-GPU box (the numbers mean nothing)
    rehearsal = os.environ.get("MRX_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    # under torch.distributed.run (RANK is set) the process group is RCCL even
    # for one rank, so that init, device binding and the collectives run for real
    if world > 1 or "RANK" in os.environ:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    n_gpus = world if world > 1 else 1
    if a.gpus != n_gpus and rank == 0:
        print(f"note: --gpus {a.gpus} but WORLD_SIZE={world}; using {n_gpus}",
              file=sys.stderr)
    torch.cuda.set_device(local)

    def barrier():
        if dist is not None:
            dist.barrier()

    if a.variant:
        os.environ["MADRONA_MI355_KERNEL"] = str(a.variant)
    first = rank * a.worlds if a.first_world is None else a.first_world
    desc = make_scene(a, first, scenes=scenes)
    r = scenes.make_renderer(desc, gpu_id=local)
    views = desc.num_views
    # The card leaves its idle power state only after some tens of milliseconds
    # of load (a 25 us step runs ~10 % slower until then), so the W warm-up
    # steps are preceded by a quarter second of untimed renders (`settle_s`).
    t_settle = time.perf_counter()
    settle_renders = 0
    while time.perf_counter() - t_settle < SETTLE_S:
        r.time_renders(100)
        settle_renders += 100
    for _ in range(a.warmup):
        r.step()
    barrier()                                   # (first use sets the communicator up)
    wall, dev_ms = timed_steps(r, a.steps, barrier)
    t = torch.tensor([wall], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = float(t.item())
    total_views = views * n_gpus
    value = total_views * a.steps / wall

    bytes_per_launch = int(r.bytes_per_step())
    kern_us = dev_ms * 1000.0 / a.steps
    achieved = bytes_per_launch / (kern_us * 1e-6) / 1e9
    achieved_wall = bytes_per_launch / (wall / a.steps) / 1e9
    traffic, traffic_source = pmc_traffic(workload_tag(a, n_gpus))
    scene_txt = ("%d cubes + plane" % a.cubes) if a.cubes else \
        "cube+plane%s%s" % ("+wall" if a.wall else "", ", textured" if a.textured else "")
    out = {
        "metric": "rendered views/sec (whole node), N worlds x %dx%d RGB+depth" % (a.width, a.height),
        "value": value,
        "unit": "views/s",
        "n_gpus": n_gpus,
        "steps": a.steps,
        "warmup": a.warmup,
        "settle_s": SETTLE_S,
        "settle_renders": settle_renders,
        "ms_per_step": wall * 1000.0 / a.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "%d worlds/GPU x %dx%d RGBA8 + f32 depth, %s, %s mode, one camera/world" % (
                a.worlds, a.width, a.height, scene_txt, a.mode),
            "worlds_per_gpu": a.worlds, "views_total": total_views,
            "width": a.width, "height": a.height,
            "kernel_variant": a.variant,
            "parallelism": "worlds sharded x%d, no collective" % n_gpus,
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            # frac_kernel: from the kernel's average launch duration (HIP events on the
            # launch stream around the K launches); frac_wall: from the host clock
            # around the same K steps (what `value` is computed from)
            "frac_kernel": achieved / HBM_PEAK_GBPS,
            "frac_wall": achieved_wall / HBM_PEAK_GBPS,
            "traffic": traffic, "traffic_source": traffic_source,
            "kernel": "mrx %s (one launch per step)" % r.render_path(),
            "kernel_us": kern_us, "bytes_per_launch": bytes_per_launch,
        },
    }

    if a.gather and dist is not None:
        rgb = r.rgb_tensor().to_torch()
        dep = r.depth_tensor().to_torch()
        g_rgb = torch.empty((n_gpus,) + tuple(rgb.shape), dtype=rgb.dtype, device="cuda")
        g_dep = torch.empty((n_gpus,) + tuple(dep.shape), dtype=dep.dtype, device="cuda")
        for _ in range(3):
            dist.all_gather_into_tensor(g_rgb, rgb)
            dist.all_gather_into_tensor(g_dep, dep)
        barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            r.step()
            dist.all_gather_into_tensor(g_rgb, rgb)
            dist.all_gather_into_tensor(g_dep, dep)
        torch.cuda.synchronize(); barrier()
        gw = time.perf_counter() - t0
        # the slab of this rank inside the gathered tensor is what it rendered
        same = bool(torch.equal(g_rgb[rank], rgb)) and bool(torch.equal(g_dep[rank], dep))
        out["with_gather"] = {"value": total_views * a.steps / gw, "unit": "views/s",
                              "ms_per_step": gw * 1000.0 / a.steps,
                              "collective": "RCCL all_gather_into_tensor rgb+depth",
                              "backend": dist.get_backend(), "own_slab_intact": same}

    if rank == 0 and n_gpus == 1 and not a.no_extra and a.worlds != 1024:
        # BASELINE.json configs[1]: 1024 worlds, same scene, reported beside it
        d2 = make_scene(a, 0, worlds=1024, scenes=scenes)
        r2 = scenes.make_renderer(d2, gpu_id=local)
        # (a new renderer's first launches carry the XCC report and cold instruction caches:
        # the same kind of untimed settle as above, 0.1 s, then the W warm-up steps)
        t_settle = time.perf_counter()
        settle2 = 0
        while time.perf_counter() - t_settle < 0.1:
            r2.time_renders(100)
            settle2 += 100
        for _ in range(a.warmup):
            r2.step()
        w2, ms2 = timed_steps(r2, a.steps, lambda: None)
        out["also"] = {"workload": "1024 worlds (BASELINE configs[1])",
                       "value": 1024 * a.steps / w2, "unit": "views/s",
                       "ms_per_step": w2 * 1000.0 / a.steps,
                       "kernel_us": ms2 * 1000.0 / a.steps,
                       "settle_s": 0.1, "settle_renders": settle2}
        del r2

    if rank == 0 and n_gpus == 1 and not a.no_cpu_baseline:
        # The reference has no CPU renderer (mgr.cpp:195-197); the baseline is
        # this repo's scalar oracle on the host cores of this box.
        from oracle import oracle
        nv = min(a.cpu_views, views)
        dcpu = make_scene(a, 0, worlds=nv, scenes=scenes)
        fs = oracle.FlatScene(dcpu)
        # a 1-GPU box owns a 16-core share of its host; stay inside it
        nthr = max(1, min(len(os.sched_getaffinity(0)), a.cpu_threads))
        fs.render(view_end=min(nv, 64), want_ids=False, num_threads=nthr)   # warm up
        # about 20 s of CPU work: whole-batch renders until 1.5 s of wall time
        # have passed (at least 3), median per render
        times = []
        t_all = time.perf_counter()
        while len(times) < 3 or time.perf_counter() - t_all < 1.5:
            t0 = time.perf_counter()
            res = fs.render(want_ids=False, num_threads=nthr)
            times.append(time.perf_counter() - t0)
        times.sort()
        med = times[len(times) // 2]
        out["cpu_baseline"] = {
            "value": nv / med, "unit": "views/s", "cores": int(res["threads"]),
            "kind": "port",
            "sample": "%d views of the same scene per render, %d renders (%.1f s wall, "
                      "%.0f core-seconds), median, OpenMP over views"
                      % (nv, len(times), sum(times), sum(times) * int(res["threads"])),
        }

    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
