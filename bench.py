#!/usr/bin/env python3
"""Headline benchmark: rendered views/sec of the batch-render hot path
(Manager::step) on N MI355X, one process per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
         --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Both forms work for N > 1: started plainly (no RANK in the environment),
``--gpus N`` makes this process a launcher that starts the N ranks itself --
before it has imported torch or touched HIP -- waits for them and relays rank
0's line (the reference has a single ``gpuID``, /root/reference/src/mgr.hpp:50;
one process per GPU is this build's multi-GPU form).  Under
``torch.distributed.run`` (RANK is set) the process is one of the ranks.

A "step" is one render of every view of the rank's worlds into the contiguous
[views,H,W,4] RGBA8 + [views,H,W,1] f32 depth tensors; worlds shard across
ranks with no data-path collective.  Rank 0 prints ONE JSON line:
  value / scaling "weak"   --worlds per GPU (default 4096, the north-star shape)
  also_strong              BASELINE configs[3]: 16384 worlds in total, 16384 / N
                           per GPU, render-only and with the RCCL all-gather of
                           the output slabs (sharding.gather_slabs)
  also                     BASELINE configs[1] (1024 worlds), N = 1 only
  also_bvh                 1024 worlds of 482 triangles through the BVH path, N = 1 only
  also_loop                the headline batch with a pose update ahead of every render, N = 1 only

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
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
# A wave64 VALU instruction occupies its SIMD for 2 cycles when several waves share the SIMD
# (MI355X_MICROARCH.md:54 and the constants table, `v_fma_f32`: 2 cyc; one wave alone: 4).
# Rounds 2-3 priced the BVH kernel against 4 cycles (614 G wave-instr/s) from a micro-benchmark
# that assumed a 2.4 GHz clock; scripts/micro/valu_issue.hip (profiles/r04_valu_issue.txt)
# measures 601 G wave-instr/s chip-wide at one wave per SIMD, 872 at two, 944 at four and 973
# at eight for independent v_fma_f32 -- the guide's figure is the right peak: 1228.8 G.
VALU_CYCLES_PER_WAVE64_INSTR = 2.0
# untimed renders before the warm-up (clocks leave idle only under load); reported as `settle_s`
SETTLE_S = float(os.environ.get("MRX_BENCH_SETTLE_S", "0.25"))
# renders per host wait during the settle: long batches keep the card under continuous load (batches of 100
# leave a gap every 2.3 ms, and a K = 20 region measured right after them ran 0.5 - 1 us per step slower:
# profiles/r03_k20.txt)
SETTLE_BATCH = int(os.environ.get("MRX_BENCH_SETTLE_BATCH", "2000"))
STRONG_WORLDS = 16384    # BASELINE.json configs[3]


def parse(argv=None):
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
                    help="many-instance worlds: this many cubes + plane per world (scenes.cube_field)")
    ap.add_argument("--variant", type=int, default=0,
                    help="mrx_config.kernel_variant: 0 default dispatch, 2 BVH path, 3 raster kernels")
    ap.add_argument("--first-world", type=int, default=None,
                    help="global id of this rank's first world (default: rank * worlds)")
    ap.add_argument("--gather", action="store_true",
                    help="also time the headline workload with an RCCL all-gather of the output slabs")
    ap.add_argument("--spawn", action="store_true",
                    help="go through the rank launcher even for --gpus 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the secondary 1024-world measurement")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the BASELINE configs[2] / configs[4] entries (`also_configs`)")
    ap.add_argument("--no-multidev", action="store_true",
                    help="skip the one-Manager multi-shard host-cost entry (`also_multidev`)")
    ap.add_argument("--no-strong", action="store_true",
                    help="skip the configs[3] strong-scaling measurement")
    ap.add_argument("--strong-worlds", type=int, default=STRONG_WORLDS,
                    help="total worlds of the strong-scaling measurement (configs[3]: 16384)")
    ap.add_argument("--cpu-views", type=int, default=4096)
    ap.add_argument("--cpu-threads", type=int, default=16)
    return ap.parse_args(argv)


# --------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` with no RANK in the environment
def launch_ranks(n, argv):
    """Start N ranks of this script (LOCAL_RANK / RANK / WORLD_SIZE / MASTER_* set),
    wait for them, relay rank 0's stdout.  This process imports neither torch nor
    the renderer and makes no HIP call: the ranks are ordinary child processes
    started before anything here could have initialised a GPU.  A rank that fails
    ends the job: the others are stopped (by PID) and its code is returned."""
    # (the port is free when asked for and handed to the ranks after the probe socket is closed: a process that
    # grabs it in between makes rank 0's rendezvous fail, the job then ends with that rank's error -- the driver's
    # own launcher, torch.distributed.run with an explicit --master-port, has the same window)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   MRX_BENCH_LAUNCHER="self")
        # dmabuf IPC is the only kind the host driver supports (RCCL needs it)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # rank 0 owns the JSON line; whatever another rank prints goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if rank == 0 else sys.stderr,
                                      text=True if rank == 0 else None))
    rc = 0
    out0 = None
    pending = set(range(n))
    while pending and rc == 0:
        for rank in sorted(pending):
            p = procs[rank]
            if rank == 0 and out0 is None:
                # (drain rank 0's pipe while waiting so it can never block on a full pipe)
                try:
                    out0, _ = p.communicate(timeout=0.2)
                except subprocess.TimeoutExpired:
                    continue
            code = p.poll()
            if code is None:
                continue
            pending.discard(rank)
            if code != 0:
                rc = code if code > 0 else 1
                print("bench.py: rank %d exited with code %d" % (rank, code), file=sys.stderr)
                break
        if pending and rc == 0:
            time.sleep(0.05)
    for rank in pending:                               # a rank failed: stop the ones still running
        p = procs[rank]
        if p.poll() is None:
            p.terminate()
    for rank in pending:
        try:
            procs[rank].wait(timeout=30)
        except subprocess.TimeoutExpired:
            procs[rank].kill()
    # rank 0's JSON line goes to stdout; anything else a library printed there
    # (gloo announces its connections on stdout) goes to stderr
    for line in (out0 or "").splitlines():
        print(line, file=sys.stdout if line.startswith("{") else sys.stderr, flush=True)
    return rc


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


def pmc_table():
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            return json.load(f)
    except Exception:
        return {}


def bvh_kernel_hash():
    """sha256 (16 hex) over the sources the BVH kernel is compiled from: an "sq" entry of
    pmc_latest.json carries the hash of the build its counters were collected on
    (scripts/pmc_sq_summary.py writes it), so a figure that no longer describes the running
    kernel is recognised."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "madrona_renderer_amd", "csrc")
    for name in ("bvh.hip", "bvh.hpp", "raster_dev.hpp", "raster.hpp"):
        try:
            with open(os.path.join(csrc, name), "rb") as f:
                h.update(f.read())
        except OSError:
            return None
    return h.hexdigest()[:16]


def pmc_traffic(tag):
    """(HBM bytes per launch, where the figure comes from) from the committed
    rocprofv3 --pmc summaries of this same command (profiles/pmc_latest.json),
    or (None, None): the counters are not collected by this run."""
    ent = pmc_table().get(tag)
    if ent is None:
        return None, None
    if isinstance(ent, dict):
        return ent.get("bytes"), "static: profiles/pmc_latest.json <- %s" % ent.get("source", "?")
    return ent, "static: profiles/pmc_latest.json"


def make_scene(a, first_world, worlds=None, scenes=None, plain=False):
    worlds = a.worlds if worlds is None else worlds
    if a.cubes and not plain:
        return scenes.cube_field(worlds, a.cubes, width=a.width, height=a.height, mode=a.mode,
                                 textured=a.textured, first_world=first_world)
    if plain:       # the BASELINE 64x64 cube+plane shape whatever the flags (configs[1], configs[3])
        return scenes.synthetic_scene(worlds, first_world=first_world)
    return scenes.synthetic_scene(worlds, width=a.width, height=a.height, with_wall=a.wall,
                                  textured=a.textured, render_mode=a.mode, first_world=first_world)


def settle(r, seconds):
    """Untimed renders for `seconds` (the card leaves its idle power state only
    after tens of milliseconds of load; a new renderer's first launches carry the
    XCC report and cold instruction caches).  Returns the render count."""
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        r.time_renders(SETTLE_BATCH)
        n += SETTLE_BATCH
    return n


def bvh_roofline(tag, kern_us, views, tiles_per_view):
    """The BVH kernel is not HBM-bound: its own roofline is the VALU
    issue rate.  VALU instructions per wave come from the committed SQ-counter summary
    of exactly this workload (profiles/pmc_latest.json, "sq" of the workload's entry;
    SQ_INSTS_VALU / SQ_WAVES); the peak is what 1024 SIMDs (256 CUs x 4) issue when
    a wave64 vector instruction occupies its SIMD for two cycles at 2.4 GHz
    (VALU_CYCLES_PER_WAVE64_INSTR above): 1228.8 G wave-instructions/s."""
    ent = pmc_table().get(tag)
    sq = ent.get("sq") if isinstance(ent, dict) else None
    if not isinstance(sq, dict) or not sq.get("valu_per_wave"):
        return None
    waves = views * tiles_per_view * 8
    valu = float(sq["valu_per_wave"]) * waves
    peak = 256 * 4 * 2.4e9 / VALU_CYCLES_PER_WAVE64_INSTR
    achieved = valu / (kern_us * 1e-6)
    # the counters describe the build they were collected on: a different kernel source since
    # then makes the figure an estimate, and the HBM roofline stays the line's primary one
    stale = sq.get("kernel_hash") != bvh_kernel_hash()
    return {"bound": "valu-issue", "achieved": achieved / 1e9, "peak": peak / 1e9,
            "unit": "G wave-instr/s", "frac": achieved / peak,
            "valu_per_wave": sq["valu_per_wave"], "sq_source": sq.get("source"),
            "sq_kernel_hash": sq.get("kernel_hash"), "stale": bool(stale)}


def run_dry(a):
    """MRX_BENCH_DRY=1: the ranks' control flow without a renderer or a GPU (gloo on
    CPU tensors) -- process-group init from the launcher's environment, barriers, the
    MAX reduction of the timing, one line from rank 0.  tests/ use it to exercise the
    launcher here, where there is no GPU; the numbers mean nothing."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("MRX_BENCH_DRY_FAIL_RANK") == str(rank):
        print("rank %d: failing on request" % rank, file=sys.stderr)
        sys.exit(3)
    dist.init_process_group("gloo")
    dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    from madrona_renderer_amd import scenes
    lo, hi = scenes.shard_range(a.strong_worlds, rank, world)
    cnt = torch.tensor([hi - lo], dtype=torch.int64)
    dist.all_reduce(cnt)
    if rank == 0:
        print(json.dumps({"dry": True, "n_gpus": world, "ranks_seen": dist.get_world_size(),
                          "max_over_ranks": float(t.item()), "strong_worlds_total": int(cnt.item()),
                          "launcher": os.environ.get("MRX_BENCH_LAUNCHER", "external"),
                          "local_rank": int(os.environ.get("LOCAL_RANK", "-1"))}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def run_rank(a):
    import torch
    from madrona_renderer_amd import scenes, sharding

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # MRX_BENCH_REHEARSAL=1: every rank on cuda:0 with gloo for the barriers --
    # rehearses the N>1 control flow on a one-GPU box (the numbers mean nothing)
    rehearsal = os.environ.get("MRX_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    elif world > 1 and local >= torch.cuda.device_count():
        print("bench.py: rank %d has no GPU (LOCAL_RANK %d, %d devices)" % (rank, local, torch.cuda.device_count()),
              file=sys.stderr)
        sys.exit(2)
    # under a launcher (RANK is set) the process group is RCCL even for one
    # rank, so that init, device binding and the collectives run for real
    if world > 1 or "RANK" in os.environ:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    n_gpus = world if world > 1 else 1
    ranks_seen = dist.get_world_size() if dist is not None else 1
    if a.gpus != n_gpus and rank == 0:
        print(f"note: --gpus {a.gpus} but WORLD_SIZE={world}; using {n_gpus}",
              file=sys.stderr)
    torch.cuda.set_device(local)
    coll_dev = "cpu" if rehearsal else "cuda"

    def barrier():
        if dist is not None:
            dist.barrier()

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device=coll_dev)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather_leg(r, steps, views_total, counts=None):
        """K steps of render + all-gather of the rank's rgb and depth slabs into the
        global tensors through sharding.gather_slabs (RCCL all_gather_into_tensor over
        xGMI; gloo on host copies in a rehearsal).  ``counts`` = views per rank when the
        shards are ragged (a world count the ranks do not divide)."""
        rgb = r.rgb_tensor().to_torch()
        dep = r.depth_tensor().to_torch()
        if rehearsal:
            stage = lambda t: t.cpu()
        else:
            stage = lambda t: t
        g_rgb = g_dep = None
        s_rgb, s_dep = {}, {}
        for _ in range(3):
            g_rgb = sharding.gather_slabs(stage(rgb), counts=counts, out=g_rgb, scratch=s_rgb)
            g_dep = sharding.gather_slabs(stage(dep), counts=counts, out=g_dep, scratch=s_dep)
        barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            r.step()
            if rehearsal:
                r.sync()
            g_rgb = sharding.gather_slabs(stage(rgb), counts=counts, out=g_rgb, scratch=s_rgb)
            g_dep = sharding.gather_slabs(stage(dep), counts=counts, out=g_dep, scratch=s_dep)
        torch.cuda.synchronize()
        gw = max_over_ranks(time.perf_counter() - t0)
        barrier()
        # the slab of this rank inside the gathered tensor is what it rendered
        nv = rgb.shape[0]
        first = rank * nv if counts is None else sum(counts[:rank])
        same = bool(torch.equal(g_rgb[first:first + nv], stage(rgb))) and \
            bool(torch.equal(g_dep[first:first + nv], stage(dep))) and \
            int(g_rgb.shape[0]) == views_total
        ok = torch.tensor([1.0 if same else 0.0], dtype=torch.float64, device=coll_dev)
        if dist is not None:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        return {"value": views_total * steps / gw, "unit": "views/s", "steps": steps,
                "ms_per_step": gw * 1000.0 / steps,
                "collective": "sharding.gather_slabs: all_gather_into_tensor of rgb + depth" +
                              ("" if counts is None or len(set(counts)) == 1 else " (ragged shards, padded)"),
                "backend": dist.get_backend() if dist is not None else None,
                "gathered_views": int(g_rgb.shape[0]), "own_slab_intact": bool(ok.item() == 1.0)}

    if a.variant:
        os.environ["MADRONA_MI355_KERNEL"] = str(a.variant)
    first = rank * a.worlds if a.first_world is None else a.first_world
    desc = make_scene(a, first, scenes=scenes)
    r = scenes.make_renderer(desc, gpu_id=local)
    views = desc.num_views
    # The card leaves its idle power state only after some tens of milliseconds
    # of load (a 25 us step runs ~10 % slower until then), so the W warm-up
    # steps are preceded by a quarter second of untimed renders (`settle_s`).
    settle_renders = settle(r, SETTLE_S)
    for _ in range(a.warmup):
        r.step()
    barrier()                                   # (first use sets the communicator up)
    wall, dev_ms = timed_steps(r, a.steps, barrier)
    wall = max_over_ranks(wall)
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
        "ranks_seen": ranks_seen,
        "launcher": os.environ.get("MRX_BENCH_LAUNCHER", "external" if "RANK" in os.environ else "none"),
        "steps": a.steps,
        "warmup": a.warmup,
        "settle_s": SETTLE_S,
        "settle_renders": settle_renders,
        "settle_batch": SETTLE_BATCH,
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
            "parallelism": "worlds sharded x%d, no collective (weak: %d worlds on every GPU)" % (n_gpus, a.worlds),
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
        # which output allocation the renderer runs on: outputs of 256 MiB and more come in a
        # fast and a ~20 % slower mode per allocation (a platform property, profiles/r03_placement.txt);
        # mrx_create times two candidates and keeps the faster -- this says what it saw
        "placement": r.placement(),
    }
    if r.render_path() == "bvh":
        tiles = ((a.width + 63) // 64) * ((a.height + 63) // 64)
        own = bvh_roofline(workload_tag(a, n_gpus), kern_us, views, tiles)
        if own is not None and own["stale"]:
            out["roofline"]["valu_issue_stale"] = own
        elif own is not None:
            # the HBM figures stay, as frac_hbm; `frac` is against the bound that applies
            hb = out["roofline"]
            own.update({"frac_hbm": hb["frac"], "achieved_hbm_GBps": hb["achieved"], "traffic": hb["traffic"],
                        "traffic_source": hb["traffic_source"], "kernel": hb["kernel"], "kernel_us": kern_us,
                        "bytes_per_launch": bytes_per_launch, "frac_wall_hbm": hb["frac_wall"]})
            out["roofline"] = own

    if a.gather and dist is not None:
        out["with_gather"] = gather_leg(r, min(a.steps, 500), total_views)
    if n_gpus == 1 and not a.no_extra:
        # The same batch as a simulation loop would drive it: a pose update on the stream ahead of every render
        # (torch add_ on the instance positions, the tensor the reference's scripts/test.py writes), K iterations;
        # the update alone for comparison.  Back-to-back renders are the best case for clocks and caches: this is
        # the other one.
        import torch
        pos = r.instance_position_tensor().to_torch()
        delta = torch.zeros_like(pos)
        k5 = min(a.steps, 2000)
        for _ in range(20):
            pos.add_(delta)
            r.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k5):
            pos.add_(delta)
            r.step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(k5):
            pos.add_(delta)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        # ... and the same loop captured ONCE into a hipGraph and replayed (step() is one launch with nothing a capture
        # rejects; tests/test_parity_gpu.py::test_step_can_be_captured...): what a launch-bound caller does
        try:
            side = torch.cuda.Stream()
            r.set_stream(side.cuda_stream)
            graph = torch.cuda.CUDAGraph()
            per_graph = 20
            with torch.cuda.stream(side):
                r.step()
                side.synchronize()
                with torch.cuda.graph(graph, stream=side):
                    for _ in range(per_graph):
                        pos.add_(delta)
                        r.step()
                for _ in range(5):
                    graph.replay()
                side.synchronize()
                reps = max(1, k5 // per_graph)
                tg0 = time.perf_counter()
                for _ in range(reps):
                    graph.replay()
                side.synchronize()
                tg1 = time.perf_counter()
            graph_ms = (tg1 - tg0) * 1000.0 / (reps * per_graph)
            del graph
        except Exception as e:                       # (a capture the runtime refuses is reported, not fatal)
            graph_ms = None
            print("note: hipGraph capture of the loop failed: %r" % (e,), file=sys.stderr)
        finally:
            torch.cuda.synchronize()
            r.set_stream(0)
        out["also_loop"] = {"workload": "the batch above, a pose update (torch add_ on the instance positions) ahead of every render",
                            "ms_per_iteration_graph_replay": graph_ms,
                            "iterations": k5, "ms_per_iteration": (t1 - t0) * 1000.0 / k5,
                            "ms_update_alone": (t2 - t1) * 1000.0 / k5,
                            "ms_per_step_back_to_back": wall * 1000.0 / a.steps}
    del r

    if n_gpus == 1 and not a.no_extra and a.worlds != 1024:
        # BASELINE.json configs[1]: 1024 worlds, same scene, reported beside it
        d2 = make_scene(a, 0, worlds=1024, scenes=scenes)
        r2 = scenes.make_renderer(d2, gpu_id=local)
        settle2 = settle(r2, 0.1)
        for _ in range(a.warmup):
            r2.step()
        w2, ms2 = timed_steps(r2, a.steps, lambda: None)
        out["also"] = {"workload": "1024 worlds (BASELINE configs[1])",
                       "value": 1024 * a.steps / w2, "unit": "views/s",
                       "ms_per_step": w2 * 1000.0 / a.steps,
                       "kernel_us": ms2 * 1000.0 / a.steps,
                       "settle_s": 0.1, "settle_renders": settle2}
        del r2

    if n_gpus == 1 and not a.no_extra and not a.cubes and a.variant == 0:
        # the BVH ray-trace path on a scene that takes it by itself (more than 128 triangles per world):
        # 1024 worlds x 64x64, 40 cubes + plane = 482 triangles, against the kernel's own roofline
        d4 = scenes.cube_field(1024, 40)
        r4 = scenes.make_renderer(d4, gpu_id=local)
        settle4 = settle(r4, 0.1)
        for _ in range(a.warmup):
            r4.step()
        w4, ms4 = timed_steps(r4, a.steps, lambda: None)
        out["also_bvh"] = {"workload": "1024 worlds x 64x64, 40 cubes + plane (482 triangles/world): the BVH path",
                           "render_path": r4.render_path(),
                           "value": 1024 * a.steps / w4, "unit": "views/s",
                           "ms_per_step": w4 * 1000.0 / a.steps, "kernel_us": ms4 * 1000.0 / a.steps,
                           "roofline": bvh_roofline("1024x64x64+cubes40", ms4 * 1000.0 / a.steps, 1024, 1),
                           "settle_s": 0.1, "settle_renders": settle4}
        del r4

    if n_gpus == 1 and not a.no_extra and not a.no_configs and a.variant == 0 and not a.cubes:
        # BASELINE.json configs[2] and configs[4] at their full size, so that the driver's own run carries them:
        # 4096 x 128x128 cube+plane+wall (tiled raster), 4096 x 256x256 Raytracer textured cube+plane through the
        # default dispatch (the BVH ray-trace path the config names: bvhFlatKernel) and through the
        # tiled raster kernel (kernel_variant 3).  Outputs of 0.5 - 3 GiB: `placement` says which mode the allocation got.
        entries = []
        for label, kw, variant, k in (
                ("configs[2]: 4096 worlds x 128x128 cube+plane+wall, tiled raster",
                 dict(width=128, height=128, with_wall=True), 0, 300),
                ("configs[4]: 4096 worlds x 256x256 Raytracer, textured cube+plane, default dispatch (the BVH ray-trace path: "
                 "bvhFlatKernel)", dict(width=256, height=256, textured=True, render_mode="Raytracer"), 0, 60),
                ("configs[4] through the tiled raster kernel (kernel_variant 3)",
                 dict(width=256, height=256, textured=True, render_mode="Raytracer"), 3, 60)):
            if variant:
                os.environ["MADRONA_MI355_KERNEL"] = str(variant)
            try:
                rc = scenes.make_renderer(scenes.synthetic_scene(4096, **kw), gpu_id=local)
            finally:
                os.environ.pop("MADRONA_MI355_KERNEL", None)
            k = min(k, max(a.steps, 20))
            settle(rc, 0.1)
            wc, msc = timed_steps(rc, k, lambda: None)
            bc = int(rc.bytes_per_step())
            entries.append({"workload": label, "render_path": rc.render_path(), "steps": k,
                            "value": 4096 * k / wc, "unit": "views/s", "ms_per_step": wc * 1000.0 / k,
                            "kernel_us": msc * 1000.0 / k, "bytes_per_launch": bc,
                            "frac_kernel": bc / (msc * 1e-3 / k) / 1e9 / HBM_PEAK_GBPS,
                            "placement": rc.placement()})
            del rc
        out["also_configs"] = entries

    if n_gpus == 1 and not a.no_extra and not a.no_multidev and a.variant == 0 and not a.cubes:
        # BASELINE configs[3] in the one-Manager form every reference caller uses (one constructor, one step():
        # /root/reference/scripts/test.py:112-130): ONE renderer of eight shards -- all on this device, each on a stream
        # of its own (a node has eight devices) -- and what one step() costs the HOST: every launch from the calling
        # thread (the round-3 form), a host thread per shard (default on a node), and MRX_SHARD_ASYNC=1 (step() posts the
        # render to the device threads; every other call joins them).  Median over bursts of 40 back-to-back steps.
        import statistics
        dmd = scenes.synthetic_scene(a.strong_worlds)
        md = {"workload": "%d worlds x 64x64 cube+plane in ONE renderer of 8 shards on one device (rehearsal of a node)"
                          % a.strong_worlds, "unit": "host us per step()"}
        keep = {k: os.environ.get(k) for k in ("MRX_SHARD_THREADS", "MRX_SHARD_ASYNC")}
        try:
            for key, thr, asy, shards in (("one_shard", "0", "0", 1), ("eight_shards_calling_thread", "0", "0", 8),
                                          ("eight_shards_threads", "2", "0", 8), ("eight_shards_threads_async", "2", "1", 8)):
                os.environ["MRX_SHARD_THREADS"], os.environ["MRX_SHARD_ASYNC"] = thr, asy
                rm = scenes.make_renderer(dmd, gpu_id=local, device_ids=[local] * shards if shards > 1 else None)
                streams = [torch.cuda.Stream() for _ in range(shards)]
                for i, st in enumerate(streams):
                    if shards > 1:
                        rm.set_stream(st.cuda_stream, shard=i)
                    else:
                        rm.set_stream(st.cuda_stream)
                rm.time_steps_host(40)
                md[key] = statistics.median(rm.time_steps_host(40) for _ in range(15))
                if shards > 1 and thr == "2" and asy == "0":
                    md["device_us_per_step"] = rm.time_renders(200) * 1000.0 / 200
                del rm, streams
        finally:
            for k, v in keep.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        out["also_multidev"] = md

    if not a.no_strong:
        # BASELINE.json configs[3]: 16384 worlds x 64x64 cube+plane IN TOTAL, world-sharded:
        # rank r renders worlds [lo, hi) of the one job (strong scaling: the driver's
        # N = 1, 2, 4, 8 values of this entry divide into each other directly), then the
        # same with the RCCL gather of the output slabs into the global tensors.
        lo, hi = scenes.shard_range(a.strong_worlds, rank, n_gpus)
        d3 = make_scene(a, lo, worlds=hi - lo, scenes=scenes, plain=True)
        r3 = scenes.make_renderer(d3, gpu_id=local)
        settle3 = settle(r3, 0.1)
        k3 = min(a.steps, 4000)
        for _ in range(min(a.warmup, 400)):
            r3.step()
        w3, ms3 = timed_steps(r3, k3, barrier)
        w3 = max_over_ranks(w3)
        b3 = int(r3.bytes_per_step())
        strong = {"workload": "%d worlds in total x 64x64 cube+plane (BASELINE configs[3]), %d per GPU"
                              % (a.strong_worlds, hi - lo),
                  "scaling": "strong", "value": a.strong_worlds * k3 / w3, "unit": "views/s",
                  "steps": k3, "ms_per_step": w3 * 1000.0 / k3, "kernel_us": ms3 * 1000.0 / k3,
                  "frac_kernel": b3 / (ms3 * 1e-3 / k3) / 1e9 / HBM_PEAK_GBPS,
                  "settle_s": 0.1, "settle_renders": settle3, "placement": r3.placement()}
        if dist is not None and n_gpus > 1:
            # (one camera per world: views per rank = worlds per rank; ragged when N does not divide)
            per_rank = [scenes.shard_range(a.strong_worlds, q, n_gpus) for q in range(n_gpus)]
            strong["with_gather"] = gather_leg(r3, min(k3, 200), a.strong_worlds,
                                               counts=[h - l for l, h in per_rank])
        out["also_strong"] = strong
        del r3

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
            "sample": "unbinned scalar oracle (every triangle tested at every pixel; the checker, not a "
                      "tuned CPU renderer): %d views of the same scene per render, %d renders (%.1f s wall, "
                      "%.0f core-seconds), median, OpenMP over views"
                      % (nv, len(times), sum(times), sum(times) * int(res["threads"])),
        }

    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main():
    a = parse()
    if "RANK" not in os.environ and (a.gpus > 1 or a.spawn):
        # the launcher: nothing below this line runs in this process
        sys.exit(launch_ranks(max(1, a.gpus), sys.argv[1:]))
    if os.environ.get("MRX_BENCH_DRY") == "1":
        run_dry(a)
    else:
        run_rank(a)


if __name__ == "__main__":
    main()
