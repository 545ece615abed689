"""Which kind of stream keeps the headline render at 22.5 us whatever other streams do?"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from madrona_renderer_amd import scenes
os.environ["MRX_PLACEMENT_TRIES"] = "1"
hip = ctypes.CDLL("libamdhip64.so")
hl = scenes.synthetic_scene(4096)
r = scenes.make_renderer(hl)


def us():
    t0 = time.time()
    while time.time() - t0 < 0.1:
        r.time_renders(50)
    return min(r.time_renders(300) for _ in range(3)) / 300 * 1000


lo, hi = ctypes.c_int(), ctypes.c_int()
hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi))
print("priority range (least, greatest):", lo.value, hi.value)


def mk(flags, prio=None):
    s = ctypes.c_void_p()
    if prio is None:
        assert hip.hipStreamCreateWithFlags(ctypes.byref(s), flags) == 0
    else:
        assert hip.hipStreamCreateWithPriority(ctypes.byref(s), flags, prio) == 0
    return s.value


others = [torch.cuda.Stream() for _ in range(6)]


def poke(i):
    with torch.cuda.stream(others[i]):
        x = torch.ones(1024, device="cuda") * 2
    torch.cuda.synchronize()


kinds = [("null stream", 0), ("blocking", mk(0)), ("non-blocking", mk(1)),
         ("non-blocking, highest priority", mk(1, hi.value)), ("blocking, highest priority", mk(0, hi.value)),
         ("non-blocking, lowest priority", mk(1, lo.value))]
for name, s in kinds:
    r.set_stream(s)
    row = [us()]
    for i in range(6):
        poke(i)
        row.append(us())
    print(f"{name:34s}", " ".join(f"{v:6.2f}" for v in row), flush=True)
