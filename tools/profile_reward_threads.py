"""How the host part of one SelfSim reward (render -> load -> trim + LANCZOS pad -> image processor) scales over threads on this
box: ms per reward (wall / rewards) at 1..64 threads.  python tools/profile_reward_threads.py"""
import sys, threading, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from detikzify_amd.infer import SyntheticTikzDocument
from detikzify_amd.util import expand, load
from tests.helpers import fake_processor
import tests.test_generate_loop as T

ip = fake_processor(T.VOCAB, T.NIMG, 384).image_processor
code = "\\begin{tikzpicture}\n" + "\n".join(f"\\draw ({i},{i%7}) -- ({i+3},{(i*5)%11});" for i in range(40)) + "\n\\end{tikzpicture}"


def reward():
    img = SyntheticTikzDocument(code).rasterize()
    im = load(img)
    ex = expand(im, max(im.size), do_trim=True)
    return ip(images=ex, return_tensors="pt")


for _ in range(3):
    reward()
for nt in (1, 2, 4, 8, 16, 32, 64):
    th = [threading.Thread(target=lambda: [reward() for _ in range(4)]) for _ in range(nt)]
    t0 = time.perf_counter(); [x.start() for x in th]; [x.join() for x in th]
    dt = time.perf_counter() - t0
    print(f"{nt:3d} threads: {1e3 * dt / (4 * nt):6.2f} ms per reward, wave of {nt} rewards {1e3 * dt / 4:7.1f} ms", flush=True)
