"""Row a·X / f3 end to end with REAL child processes: `latexmk` is an executable on PATH (a script that behaves like the
TeX driver as far as TikzDocument can tell: reads the .tex, writes <tex>.pdf and a -file-line-error log, exit status,
optionally hangs with a grandchild), started by fork/exec through util.check_output with its timeout and process-group
kill — not an injected toolchain object.  The three Python-side tools that are absent offline (pymupdf, pdfCropMargins,
pdf2image) are tiny stand-in MODULES on PYTHONPATH, so the spawned workers of the compile pool import them like the real
ones.  Reference: detikzify/infer/tikz.py:89-156 (compile / rasterize), examples/refine.py:151-185 (process pool + imap)."""
import os
import stat
import sys
import textwrap
import time
from pathlib import Path

import pytest

FAKE_LATEXMK = r'''#!/usr/bin/env python3
import os, subprocess, sys, time
tex = sys.argv[-1]
engine = [a[1:] for a in sys.argv[1:] if a in ("-pdflatex", "-lualatex", "-xelatex")][0]
assert "-file-line-error" in sys.argv and "-interaction=nonstopmode" in sys.argv and os.environ.get("max_print_line") == "1000"
code = open(tex).read()
lines = code.split("\n")
print(f"Latexmk: fake run of {engine}\n({tex}")
once = os.path.join(os.environ["FAKE_TEX_STATE"], "hung.once")
if "HANG" in code and not ("HANG_ONCE" in code and os.path.exists(once)):   # a run that never ends and has a child of its own (TeX spawns helpers)
    open(once, "w").write("x")
    child = subprocess.Popen(["sleep", "600"])
    open(os.path.join(os.environ["FAKE_TEX_STATE"], "grandchild.pid"), "w").write(str(child.pid))
    time.sleep(600)
if "SLOW" in code:
    time.sleep(0.4)
bad = [i + 1 for i, l in enumerate(lines) if "\\undefinedmacro" in l]
only = [l for l in lines if l.startswith("%ENGINE ")]
if only and engine not in only[0]:      # this document only compiles with another engine
    print(f"{tex}:1: Engine {engine} cannot do this.")
    sys.exit(12)
open(tex + ".pdf", "w").write("PDF " + engine + " " + str(sum(map(ord, code)) % 9973) + " pages=2")
if bad:
    print(f"{tex}:{bad[0]}: Undefined control sequence.")
    if "FATAL" in code:
        os.remove(tex + ".pdf")
    sys.exit(12)
'''

FAKE_MODULES = {
    "pymupdf.py": '''
        import builtins
        class _Doc:
            def __init__(self, path):
                self.text = builtins.open(path).read()
            def __len__(self):
                return int(self.text.rsplit("pages=", 1)[1]) if "pages=" in self.text else 1
            def select(self, pages):
                self.text = self.text.split(" pages=")[0] + f" lastpage={pages[0]}"
            def save(self, path):
                builtins.open(path, "w").write(self.text)
            def tobytes(self):
                return self.text.encode()
            def __bool__(self):
                return True
        def open(path):
            return _Doc(path)
    ''',
    "pdfCropMargins.py": '''
        import shutil
        def crop(args, quiet=False):
            shutil.copy(args[-1], args[args.index("-o") + 1])
    ''',
    "pdf2image/__init__.py": "",
    "pdf2image/pdf2image.py": '''
        from PIL import Image, ImageDraw
        def convert_from_bytes(data, size=420, single_file=True):
            seed = sum(data) % 251
            img = Image.new("RGB", (size, size * 2 // 3), "white")
            d = ImageDraw.Draw(img)
            d.line([seed, 5, size - 5, (seed * 7) % (size * 2 // 3)], fill="black", width=4)
            d.text((10, 10), data.decode()[:40], fill="black")
            return [img]
    ''',
}


@pytest.fixture()
def tex_box(tmp_path, monkeypatch):
    """a directory with the fake TeX driver first on PATH and the stand-in modules first on PYTHONPATH"""
    bindir, moddir, state = tmp_path / "bin", tmp_path / "mods", tmp_path / "state"
    for d in (bindir, moddir / "pdf2image", state):
        d.mkdir(parents=True)
    exe = bindir / "latexmk"
    exe.write_text(FAKE_LATEXMK)
    exe.chmod(exe.stat().st_mode | stat.S_IXUSR | stat.S_IXGRP | stat.S_IXOTH)
    for name, body in FAKE_MODULES.items():
        (moddir / name).write_text(textwrap.dedent(body))
    monkeypatch.setenv("PATH", f"{bindir}{os.pathsep}{os.environ['PATH']}")
    monkeypatch.setenv("PYTHONPATH", f"{moddir}{os.pathsep}{os.environ.get('PYTHONPATH', '')}")
    monkeypatch.setenv("FAKE_TEX_STATE", str(state))
    monkeypatch.syspath_prepend(str(moddir))
    for m in ("pymupdf", "pdfCropMargins", "pdf2image", "pdf2image.pdf2image"):
        monkeypatch.delitem(sys.modules, m, raising=False)
    return state


GOOD = "\\documentclass{standalone}\n\\begin{document}\n\\draw (0,0) -- (1,1);\n\\end{document}\n"
BROKEN = "\\documentclass{standalone}\n\\begin{document}\n\\undefinedmacro\n\\draw (0,0);\n\\end{document}\n"
FATAL = BROKEN + "% FATAL\n"
LUA_ONLY = "%ENGINE lualatex\n" + GOOD


def _pid_alive(pid: int) -> bool:
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    try:        # a zombie still answers signal 0: read its state
        return Path(f"/proc/{pid}/stat").read_text().split(") ")[1][0] != "Z"
    except OSError:
        return False


def test_tikz_document_compiles_through_a_real_latexmk_process(tex_box):
    from detikzify_amd.infer import TikzDocument
    good = TikzDocument(GOOD)
    assert good.status == 0 and not good.compiled_with_errors and good.errors == {}
    assert good.is_rasterizable and good.has_content and good.rasterize().size == (420, 420)
    assert good.pdf.tobytes().decode().startswith("PDF pdflatex") and "lastpage=1" in good.pdf.tobytes().decode()   # last of 2 pages kept
    broken = TikzDocument(BROKEN)                  # recoverable error: a PDF is still produced (latexmk -f)
    assert broken.status == 12 and broken.compiled_with_errors and broken.is_rasterizable
    assert list(broken.errors) == [4] and "Undefined control sequence" in broken.errors[4]    # line 3 of the code + the inserted pagestyle line
    fatal = TikzDocument(FATAL)
    assert fatal.status == 12 and not fatal.is_rasterizable and list(fatal.errors) == [4]
    lua = TikzDocument(LUA_ONLY)                   # engine fallback: pdflatex fails, lualatex succeeds (tikz.py:28,111-131)
    assert lua.status == 0 and lua.pdf.tobytes().decode().startswith("PDF lualatex")


def test_a_hanging_tex_run_is_killed_with_its_children(tex_box):
    from detikzify_amd.infer import TikzDocument
    TikzDocument.set_engines("pdflatex")
    try:
        t0 = time.perf_counter()
        doc = TikzDocument(GOOD + "% HANG\n", timeout=1)
        assert doc.status == -1 and not doc.is_rasterizable          # TimeoutExpired: no return code, nothing produced
        assert time.perf_counter() - t0 < 20
        pid = int((tex_box / "grandchild.pid").read_text())
        for _ in range(50):
            if not _pid_alive(pid):
                break
            time.sleep(0.1)
        assert not _pid_alive(pid), "the TeX run's own child survived the timeout"
    finally:
        TikzDocument.set_engines(["pdflatex", "lualatex", "xelatex"])


def test_compile_pool_runs_latex_in_worker_processes(tex_box):
    from detikzify_amd.infer import TikzDocument
    from detikzify_amd.infer.compile_pool import CompilePool, pooled_document_class
    codes = [GOOD, BROKEN, FATAL, LUA_ONLY] + [GOOD + f"% SLOW {i}\n" for i in range(6)]
    direct = [TikzDocument(c) for c in codes[:4]]
    with CompilePool(workers=3) as pool:
        assert 1 <= pool.warm() <= 3                                  # workers up (spawn + imports) before timing
        t0 = time.perf_counter()
        figs = list(pool.imap(codes))
        dt = time.perf_counter() - t0
        assert [f.status for f in figs] == [0, 12, 12, 0] + [0] * 6
        assert [f.png is not None for f in figs] == [True, True, False, True] + [True] * 6
        assert [f.log for f in figs[:4]] != [d.log for d in direct]   # temp-file names differ ...
        assert dt < 6 * 0.4 + 1.0, f"6 slow documents took {dt:.2f} s on 3 workers: no overlap"
        # the document class that compiles in the pool: same verdicts as the in-process document
        Pooled = pooled_document_class(pool)
        docs = [Pooled(c).prefetch() for c in codes[:4]]              # all four in flight before the first result is read
        for d, ref in zip(docs, direct):
            assert (d.status, d.is_rasterizable, list(d.errors), d.compiled_with_errors) == \
                   (ref.status, ref.is_rasterizable, list(ref.errors), ref.compiled_with_errors)
            if ref.is_rasterizable:
                assert d.rasterize().size == ref.rasterize().size == (420, 420)
                assert d.rasterize().tobytes() == ref.rasterize().tobytes()
        assert docs[0].pdf.tobytes() == direct[0].pdf.tobytes()


def test_parallel_trees_compile_in_the_pool_while_the_others_decode(tex_box):
    """the MCTS with the real reward path: three trees on the scripted device, every rollout's document goes through the
    fake TeX driver in a pool worker; scores are compiler diagnostics (metric "fast"), so they are in {-1, 0, 1}"""
    from detikzify_amd.infer import DetikzifyPipeline
    from detikzify_amd.infer.compile_pool import CompilePool, pooled_document_class

    from .helpers import fake_processor, sketch_image
    from .test_generate_loop import NIMG, VOCAB, ScriptedDevice
    with CompilePool(workers=2) as pool:
        pipe = DetikzifyPipeline(ScriptedDevice(slots=4), fake_processor(VOCAB, NIMG), metric="fast", max_length=NIMG + 30,
                                 document_class=pooled_document_class(pool), compile_timeout=20)
        out = list(pipe.simulate(sketch_image(9, 96), expansions=2, trees=3))
        assert len(out) == 6 and all(s in (-1, 0, 1) for s, _ in out)
        assert all(d.status in (0, 12) for _, d in out)
        seq = list(pipe.simulate(sketch_image(9, 96), expansions=2))           # one tree: the same class works sequentially
        assert len(seq) == 2


def _wait_for(path, what):
    for _ in range(100):
        if path.exists():
            return
        time.sleep(0.1)
    raise AssertionError(what)


def test_a_dying_worker_fails_one_compile_not_the_search(tex_box):
    """a worker killed under a TeX run (OOM killer) breaks concurrent.futures' executor for good: the pool replaces it and the
    lost job is run ONCE more (the executor fails every pending job, innocent siblings included, and a cached failed
    compile is a reward of -1 for good) — in an executor of its own (on the shared pool a poison document broke the new
    executor again and took the siblings' second attempt with it); a job that keeps killing its worker reads as a failed compile
    (status -1, nothing to rasterise), costs the shared pool ONE restart, and the jobs after it compile normally"""
    import signal
    from detikzify_amd.infer.compile_pool import CompilePool, pooled_document_class
    pidfile = tex_box / "grandchild.pid"

    def kill_worker_and_tex(pool, isolated=False):
        _wait_for(pidfile, "the fake TeX run never started")
        ex = pool._isolated if isolated else pool._pool      # the retry of a lost job runs in an executor of its own
        worker = next(iter(ex._processes))                   # the one worker process
        os.kill(worker, signal.SIGKILL)
        os.kill(int(pidfile.read_text()), signal.SIGKILL)
        pidfile.unlink()

    with CompilePool(workers=1) as pool:
        assert pool.warm() == 1
        Pooled = pooled_document_class(pool)
        victim = Pooled(GOOD + "% HANG_ONCE\n", timeout=30).prefetch()       # hangs the first time only: an OOM kill, not a bad document
        sibling = Pooled(GOOD + "% sibling\n").prefetch()                    # queued behind it on the same executor
        kill_worker_and_tex(pool)
        assert victim.status == 0 and victim.is_rasterizable                 # the isolated retry compiled it
        assert sibling.status == 0 and sibling.is_rasterizable               # ... and the innocent sibling is not a failed compile
        assert pool.restarts == 1
        deadly = Pooled(GOOD + "% HANG\n", timeout=30).prefetch()            # this one takes its worker down every time
        kill_worker_and_tex(pool)
        import threading
        second = threading.Thread(target=kill_worker_and_tex, args=(pool, True))  # the retry hangs again: kill ITS worker too
        second.start()
        assert deadly.status == -1 and not deadly.is_rasterizable
        second.join(timeout=30)
        assert pool.restarts == 2                                            # the isolated retry's death did not touch the shared pool
        after = Pooled(GOOD)
        assert after.status == 0 and after.is_rasterizable
        figs = list(pool.imap([GOOD, BROKEN]))
        assert [f.status for f in figs] == [0, 12]


def test_lost_siblings_retry_side_by_side_and_a_hung_retry_is_cut_off(monkeypatch):
    """when a worker dies, concurrent.futures fails every pending job of the executor — a 64-tree reward wave loses up to
    64 innocent siblings — and they used to retry strictly one at a time behind one lock (a process spawn, the package import and
    a compile each).  Now up to `retry_slots` isolated executors run side by side, and the wait for one is bounded: a retry whose
    worker hangs is terminated after the job's timeout + a start-up allowance and reads as lost."""
    import threading
    from detikzify_amd.infer.compile_pool import CompilePool
    from detikzify_amd.infer.tikz import SleepingSyntheticTikzDocument
    monkeypatch.setenv("DTK_SYNTH_COMPILE_SECONDS", "1.5")
    with CompilePool(workers=4, document_class=SleepingSyntheticTikzDocument) as pool:
        assert pool.retry_slots == 4
        t0 = time.perf_counter()
        one = pool.retry_isolated("\\draw (0,0) -- (1,1);\nwarm\n", timeout=30)       # what ONE retry costs here: spawn + import + 1.5 s
        t_one = time.perf_counter() - t0
        assert one is not None and one.status == 0
        figs = [None] * 4

        def retry(i):
            figs[i] = pool.retry_isolated(f"\\draw (0,0) -- ({i},1);\nline {i}\n", timeout=30)
        t0 = time.perf_counter()
        ths = [threading.Thread(target=retry, args=(i,)) for i in range(4)]
        [t.start() for t in ths]
        [t.join(timeout=120) for t in ths]
        t_four = time.perf_counter() - t0
        assert all(f is not None and f.status in (0, 12) for f in figs) and pool.retries == 5 and pool.retries_lost == 0
        assert t_four < 2.5 * t_one + 2.0, f"four retries took {t_four:.1f} s, one takes {t_one:.1f} s: serialised"
        # a retry that hangs (compile 1.5 s, allowance 0.0 + timeout 0.2): cut off, counted as lost, its worker gone
        pool.retry_slack_s = 0.0
        assert pool.retry_isolated("\\draw (0,0) -- (2,2);\nhang\n", timeout=0.2) is None and pool.retries_lost == 1
        assert pool._isolated is None


def test_pool_inherits_the_parents_engines_and_rasterises_other_sizes_from_the_pdf(tex_box):
    from detikzify_amd.infer import TikzDocument
    from detikzify_amd.infer.compile_pool import CompilePool, pooled_document_class
    TikzDocument.set_engines("pdflatex")
    try:
        with CompilePool(workers=1) as pool:
            assert pool.engines == ["pdflatex"]
            Pooled = pooled_document_class(pool)
            assert Pooled(LUA_ONLY).status == 12                 # the worker did not fall back to lualatex
            direct, pooled = TikzDocument(GOOD), Pooled(GOOD)
            for size in (420, 224):                              # 224: re-rasterised from the PDF like the base class, no resample
                assert pooled.rasterize(size=size).tobytes() == direct.rasterize(size=size).tobytes()
    finally:
        TikzDocument.set_engines(["pdflatex", "lualatex", "xelatex"])


def test_multi_gpu_example_runs_as_a_script_with_a_compile_pool(tex_box, tmp_path):
    """examples/mcts_multi_gpu.py started the way a user starts it (`python examples/...`), LaTeX in a 1-worker spawn pool: the
    spawned worker re-imports the main script as __mp_main__, which must not parse arguments, join a process group or load a
    model again (round 2's script did all three at module level and died with BrokenProcessPool).  The device is the scripted
    one of tests/test_generate_loop.py (DTK_EXAMPLE_LOADER), everything else is the shipped path."""
    import subprocess
    from .helpers import sketch_image
    root = Path(__file__).resolve().parents[1]
    image = tmp_path / "sketch.png"
    sketch_image(3, 96).save(image)
    env = dict(os.environ, DTK_EXAMPLE_LOADER="tests.test_generate_loop:example_loader",
               PYTHONPATH=f"{root}{os.pathsep}{os.environ.get('PYTHONPATH', '')}")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, str(root / "examples" / "mcts_multi_gpu.py"), "--model", "scripted", "--image", str(image),
                          "--trees", "2", "--expansions", "2", "--tex-workers", "1", "--metric", "fast", "--keep", "2"],
                         capture_output=True, text=True, timeout=240, env=env, cwd=tmp_path)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    assert res.stdout.count("% score") == 2, res.stdout[-2000:]


def test_documents_of_parallel_trees_compile_concurrently(monkeypatch):
    """the first read of `is_rasterizable` is what runs the compile; as a functools.cached_property (Python < 3.12: one lock per
    descriptor, shared by ALL instances) it serialised the LaTeX runs of every tree of a parallel search.  Eight documents whose
    compile takes 0.4 s each, read from eight threads, must take about one compile, not eight."""
    import threading
    from detikzify_amd.infer.tikz import SleepingSyntheticTikzDocument
    monkeypatch.setenv("DTK_SYNTH_COMPILE_SECONDS", "0.4")
    docs = [SleepingSyntheticTikzDocument(f"\\draw (0,0) -- ({i},1);\nline {i}\n") for i in range(8)]
    seen = [None] * 8

    def read(i):
        seen[i] = (docs[i].is_rasterizable, docs[i].has_content)
    t0 = time.perf_counter()
    ths = [threading.Thread(target=read, args=(i,)) for i in range(8)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    dt = time.perf_counter() - t0
    assert dt < 1.6, f"8 concurrent 0.4 s compiles took {dt:.2f} s: serialised"
    t0 = time.perf_counter()
    assert [(d.is_rasterizable, d.has_content) for d in docs] == seen       # memoised: no second compile
    assert time.perf_counter() - t0 < 0.2
