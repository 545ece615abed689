"""Host cost of one step() of a multi-device Manager.

BASELINE configs[3] (16384 worlds x 64x64) on ONE renderer of 1 / 2 / 4 / 8 shards, all on
device 0 of the one-GPU box (what differs on a real node is which device a launch lands on).
Measured: host wall time per mrx_step call over bursts of back-to-back calls with no
synchronisation in between (mrx_time_steps_host) -- what the calling thread pays to have one
step enqueued on every shard -- with a host thread per shard (MRX_SHARD_THREADS=2: what a node
of distinct devices gets by default, one thread per device) and without (MRX_SHARD_THREADS=0,
the round-3 form: every launch from the calling thread), on the shards' default null stream
(shared here, one per device on a node) and on a non-blocking stream per shard (separate
queues, as separate devices have).

    python scripts/multidev_host.py > profiles/r04_multidev_host.txt
"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from madrona_renderer_amd import scenes  # noqa: E402

WORLDS = int(os.environ.get("MRX_HOST_WORLDS", "16384"))
BURST = 40
REPS = 25


def measure(shards, threads, own_streams, async_steps=False):
    os.environ["MRX_SHARD_THREADS"] = "2" if threads else "0"
    os.environ["MRX_SHARD_ASYNC"] = "1" if async_steps else "0"
    desc = scenes.synthetic_scene(WORLDS)
    r = scenes.make_renderer(desc, device_ids=[0] * shards if shards > 1 else None)
    streams = []
    if own_streams:
        for i in range(shards):
            s = torch.cuda.Stream()
            streams.append(s)
            if shards > 1:
                r.set_stream(s.cuda_stream, shard=i)
            else:
                r.set_stream(s.cuda_stream)
    r.time_steps_host(BURST)
    us = [r.time_steps_host(BURST) for _ in range(REPS)]
    dev = r.time_renders(200) * 1000.0 / 200
    del r
    return statistics.median(us), min(us), dev


def main():
    print("# %d worlds x 64x64 cube+plane in ONE renderer, shards all on device 0 (%s)" %
          (WORLDS, torch.cuda.get_device_name(0)))
    print("# host us per step() = median (min) over %d bursts of %d back-to-back calls, no sync inside;"
          % (REPS, BURST))
    print("# device us per step = mrx_time_renders (the slowest shard's launches between its two events)")
    print("%-7s %-22s %-14s %18s %14s" % ("shards", "launch", "stream", "host us/step", "device us/step"))
    base = {}
    for own in (False, True):
        for threads in (False, True):
            for n in (1, 2, 4, 8):
                if n == 1 and threads:
                    continue
                med, lo, dev = measure(n, threads, own)
                if n == 1:
                    base[own] = med
                print("%-7d %-22s %-14s %9.2f (%6.2f) %14.2f   x%.2f of 1 shard" %
                      (n, "threads" if threads and n > 1 else "calling thread",
                       "per shard" if own else "null", med, lo, dev, med / base[own]))
                sys.stdout.flush()
    # opt-in MRX_SHARD_ASYNC=1: step() posts the render to the device threads and returns
    for n in (2, 4, 8):
        med, lo, dev = measure(n, True, True, async_steps=True)
        print("%-7d %-22s %-14s %9.2f (%6.2f) %14.2f   x%.2f of 1 shard" %
              (n, "threads, posted (async)", "per shard", med, lo, dev, med / base[True]))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
