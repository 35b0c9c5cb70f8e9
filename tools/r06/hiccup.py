"""Where does the slow 20-step window of the default bench come from (one window at 7.5 ms per step among windows at 6.2)?  Per-step host
and device times of the bench's own Stepper over 120 steps, with the garbage collector's runs logged.  python tools/r06/hiccup.py [nogc]"""
import gc, os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from samplenerfro_amd import synthetic as syn, prng
from samplenerfro_amd.utils import Rays
dev = torch.device("cuda:0")
cfg = dict(syn.CONFIGS["ship_straight"])
model, variables, pf = bench.build_scene(cfg, dev, "f16x3", 0, "radiance", None)
o, d = syn.sphere_rays(4096, seed=syn.SEED)
rays = Rays(torch.from_numpy(o).to(dev), None, torch.from_numpy(d).to(dev), None)
key = prng.PRNGKey(syn.SEED)
args = types.SimpleNamespace(reserve_cus=32)
if "prime0" in sys.argv:
    bench.PRIME_STEPS = 0                  # every step of the process is on the record
st = bench.Stepper(args, cfg, model, variables, rays, key, 4096, 1, 0, 0, dev, "f16x3", "train", "radiance", True, False)
events = []
def cb(phase, info):
    if phase == "start": cb.t = time.perf_counter()
    else: events.append((len(host), info["generation"], 1e3 * (time.perf_counter() - cb.t), info.get("collected")))
gc.callbacks.append(cb)
if len(sys.argv) > 1 and sys.argv[1] == "nogc":
    gc.disable()
if len(sys.argv) > 1 and sys.argv[1] == "freeze":
    gc.collect(); gc.freeze()
host, evs = [], []
torch.cuda.synchronize()
if "events0" in sys.argv:                  # no per-step event: only host clocks, and a synchronize after every 20 steps like the bench's windows
    for w in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter(); hs = []
        for i in range(20):
            t = time.perf_counter(); st.step(); hs.append(1e3 * (time.perf_counter() - t))
        torch.cuda.synchronize()
        print(f"window {w}: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms per step; host ms per call:", " ".join(f"{x:.1f}" for x in hs))
    sys.exit(0)
for i in range(120):
    t = time.perf_counter()
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
    st.step()
    host.append(1e3 * (time.perf_counter() - t))
e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
torch.cuda.synchronize()
devms = [evs[i].elapsed_time(evs[i + 1]) for i in range(120)]
print("device ms per step (between events):", " ".join(f"{x:.1f}" for x in devms))
print("host ms per step call:", " ".join(f"{x:.1f}" for x in host))
print("gc runs (step, generation, ms, collected):", events)
for w in range(6):
    print(f"window {w}: {sum(devms[20 * w:20 * w + 20]) / 20:.3f} ms per step")
