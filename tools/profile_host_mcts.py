"""Wall time per host function of the parallel MCTS on the emulated device of tools/host_emulation.py (no GPU): which Python-side
calls the trees spend their time in when 64 of them run at once.  python tools/profile_host_mcts.py"""
import sys, time, threading, collections
sys.argv = ["x", "--trees", "64", "--expansions", "2", "--step-ms", "4.6", "--new-tokens", "512"]
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
acc = collections.defaultdict(float); cnt = collections.defaultdict(int); lock = threading.Lock()
def wrap(cls, name, label=None):
    f = getattr(cls, name); label = label or f"{cls.__name__}.{name}"
    def w(*a, **k):
        t = time.perf_counter()
        try: return f(*a, **k)
        finally:
            dt = time.perf_counter() - t
            with lock: acc[label] += dt; cnt[label] += 1
    setattr(cls, name, w)
from detikzify_amd.infer import generate as G, tikz as TZ
from detikzify_amd.evaluate import imagesim as IS
from detikzify_amd.model import processing as PR
wrap(G.DetikzifyGenerator, "decode"); wrap(G.DetikzifyGenerator, "score"); wrap(G.DetikzifyGenerator, "child_finder"); wrap(G.DetikzifyGenerator, "merge")
wrap(G.DetikzifyGenerator, "__init__", "Generator.__init__")
wrap(IS.ImageSim, "get_vision_features"); wrap(TZ.SyntheticTikzDocument, "rasterize")
wrap(PR.DetikzifyImageProcessor, "preprocess")
exec(open(ROOT / "tools" / "host_emulation.py").read())
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"{k:40s} {cnt[k]:5d} calls  {v:8.2f} s total  {1e3 * v / max(1, cnt[k]):8.2f} ms each")
