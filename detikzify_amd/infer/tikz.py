"""
TikzDocument — the CPU reward back-end (row a·X, unchanged by design): compile with latexmk, keep the last page,
crop, rasterise, read `file:line:error` diagnostics.  Behaviour of reference detikzify/infer/tikz.py:21-168: engine
fallback in order, the engine whose first error comes LATEST decides status and log (:111-131), diagnostics keyed by
line with 0 for "elsewhere" (:54-73), 420 px raster (:149-156).

The external programs (latexmk of TeX Live, pymupdf, pdfCropMargins, pdf2image / poppler) sit behind `LatexToolchain`,
so the compile policy above is testable with a fake toolchain (tests/test_host_logic.py) — none of them exists in this
environment.  A missing toolchain degrades like the reference (:141-142): an error is logged, status -1, nothing to
rasterise.

SyntheticTikzDocument is a TeX-free stand-in with the same interface (deterministic pseudo compile result derived from
a hash of the code) used by tests and by bench.py's stub reward — never mixed with real-reward numbers.
"""
from __future__ import annotations

import hashlib
import logging
from collections import namedtuple
from io import BytesIO
from os import environ
from os.path import isfile, join
from re import MULTILINE, escape, findall, search
from subprocess import DEVNULL, CalledProcessError, TimeoutExpired
from tempfile import NamedTemporaryFile, TemporaryDirectory
from typing import Dict, List, Optional, Union

from PIL import Image, ImageDraw

from ..util import check_output, expand

logger = logging.getLogger("detikzify_amd")

Output = namedtuple("Output", ["pdf", "status", "log"], defaults=[None, -1, ""])

NO_PAGE_NUMBERS = r"{cmd}\AtBeginDocument{{{cmd}}}".format(cmd=r"\thispagestyle{empty}\pagestyle{empty}")


class LatexToolchain:
    """The programs a compile needs.  Constructing it imports the Python-side ones (ImportError = not installed);
    latexmk itself is found (or not: FileNotFoundError) when it is first run."""

    def __init__(self):
        import pymupdf
        from pdfCropMargins import crop
        self._pymupdf, self._crop = pymupdf, crop

    def latexmk(self, engine: str, texfile: str, cwd: str, timeout: Optional[int]):
        """one latexmk run; raises CalledProcessError / TimeoutExpired (with .output = the log) when it fails"""
        check_output(cwd=cwd, timeout=timeout, stderr=DEVNULL,
                     env=environ | dict(max_print_line="1000"),     # long log lines: file:line:error stays on one line
                     args=["latexmk", "-f", "-nobibtex", "-norc", "-file-line-error", "-interaction=nonstopmode",
                           f"-{engine}", texfile])

    def keep_last_page(self, src_pdf: str, dst_pdf: str):
        doc = self._pymupdf.open(src_pdf)
        doc.select([len(doc) - 1])
        doc.save(dst_pdf)

    def crop(self, src_pdf: str, dst_pdf: str):
        self._crop(["-gsf", "-c", "gb", "-p", "0", "-a", "-1", "-o", dst_pdf, src_pdf], quiet=True)

    def open_pdf(self, path: str):
        return self._pymupdf.open(path)

    @staticmethod
    def to_image(pdf, size: int) -> Image.Image:
        from pdf2image.pdf2image import convert_from_bytes
        return convert_from_bytes(pdf.tobytes(), size=size, single_file=True)[0]


def first_error_line(log: str, texfile: str) -> int:
    """line of the first `texfile:LINE: message` diagnostic in a latexmk log, 0 if there is none"""
    hit = search(rf"^{escape(texfile)}:(\d+):.+$", log, MULTILINE)
    return int(hit.group(1)) if hit else 0


def located_errors(log: str) -> Dict[int, str]:
    """{line: message} of a -file-line-error log; errors outside the root file (the first `(path` the log opens) go
    under line 0; a log without any located error still reports the fatal one"""
    opened = search(r"^\((.+)$", log, MULTILINE)
    root = opened.group(1) if opened else None
    out: Dict[int, str] = {}
    for file, line, message in findall(r"^(.+):(\d+):(.+)$", log, MULTILINE):
        out[int(line) if file == root else 0] = message.strip()
    return out or {0: "Fatal error occurred, no output PDF file produced!"}


class TikzDocument:
    engines: List[str] = ["pdflatex", "lualatex", "xelatex"]
    toolchain = LatexToolchain          # a callable returning the toolchain; tests substitute a fake
    Output = Output

    def __init__(self, code: str, timeout: Optional[int] = 60):
        self.code = code
        self.timeout = timeout
        self._compiled: Optional[Output] = None

    # ---- compile results (one compile per document, reference :31) --------------------------------------------
    def compile(self) -> Output:
        if self._compiled is None:
            self._compiled = self._compile()
        return self._compiled

    @property
    def status(self) -> int:
        return self.compile().status

    @property
    def pdf(self):
        return self.compile().pdf

    @property
    def log(self) -> str:
        return self.compile().log

    @property
    def compiled_with_errors(self) -> bool:
        return self.status != 0

    @property
    def errors(self) -> Dict[int, str]:
        return located_errors(self.log) if self.compiled_with_errors else {}

    # Memoised per document WITHOUT functools.cached_property: until Python 3.12 that descriptor computes under ONE lock shared by
    # every instance of the class, and the first read of `is_rasterizable` is what runs latexmk (1-60 s) — the LaTeX runs of all
    # the trees of a parallel search were serialised behind it (bench.py --reward-latency: 16 trees, 1 s per compile -> 1.0
    # rollouts/s, pool or no pool).  The reference declares both as cached_property (infer/tikz.py); same values, no lock.
    @property
    def is_rasterizable(self) -> bool:
        if "is_rasterizable" not in self.__dict__:
            self.__dict__["is_rasterizable"] = self.rasterize() is not None
        return self.__dict__["is_rasterizable"]

    @property
    def has_content(self) -> bool:
        if "has_content" not in self.__dict__:
            img = self.rasterize()
            self.__dict__["has_content"] = img is not None and img.getcolors(1) is None
        return self.__dict__["has_content"]

    @classmethod
    def set_engines(cls, engines: Union[str, list]):
        cls.engines = [engines] if isinstance(engines, str) else engines

    # ---- the compile policy -----------------------------------------------------------------------------------------
    def _source(self) -> str:
        """the document with page numbers switched off right after its first line (they would defeat cropping)"""
        lines = self.code.split("\n")
        lines.insert(1, NO_PAGE_NUMBERS)
        return "\n".join(lines)

    def _compile(self) -> Output:
        try:
            tools = self.toolchain()
        except ImportError as e:        # same degradation as a missing TeX Live
            logger.error("Missing dependencies: %s (TeX Live, ghostscript, poppler needed)", e)
            return Output()
        status, log, pdf = -1, "", None
        with TemporaryDirectory() as workdir, NamedTemporaryFile(dir=workdir, buffering=0) as tex:
            tex.write(self._source().encode())
            produced, last_page, cropped = f"{tex.name}.pdf", join(workdir, "tikz.pdf"), f"{tex.name}.crop"
            open(f"{tex.name}.bbl", "a").close()        # some classes insist on a bibliography file
            try:
                furthest = -1
                for engine in self.engines:
                    try:
                        tools.latexmk(engine, tex.name, workdir, self.timeout)
                    except (CalledProcessError, TimeoutExpired) as failed:
                        engine_log = (getattr(failed, "output", b"") or b"").decode(errors="ignore")
                        line = first_error_line(engine_log, tex.name)
                        if line <= furthest:
                            continue            # an earlier engine got at least as far: its verdict stands
                        furthest, status, log = line, getattr(failed, "returncode", -1), engine_log
                    else:
                        status, log = 0, ""
                    try:                        # whatever this engine produced: its last page is the figure
                        tools.keep_last_page(produced, last_page)
                    except Exception:  # noqa: BLE001
                        pass
                    if status == 0:
                        break
                tools.crop(last_page, cropped)
                if isfile(cropped):
                    pdf = tools.open_pdf(cropped)
            except FileNotFoundError:
                logger.error("Missing dependencies: Did you install TeX Live?")
            except RuntimeError:            # pdf trouble while cropping
                pass
        if status == 0 and not pdf:
            logger.warning("Could compile document but something seems to have gone wrong during cropping!")
        return Output(pdf=pdf, status=status, log=log)

    def rasterize(self, size: int = 420, expand_to_square: bool = True, **_) -> Optional[Image.Image]:
        pdf = self.pdf
        if not pdf:
            return None
        image = self.toolchain.to_image(pdf, size)
        return expand(image, size) if expand_to_square else image

    def save(self, filename: str, *args, **kwargs):
        kind = filename.rsplit(".", 1)[-1]
        if kind == "tex":
            data = self.code.encode()
        elif kind == "pdf" and self.pdf:
            data = self.pdf.tobytes()
        else:
            image = self.rasterize(*args, **kwargs)
            if image is None:
                raise ValueError(f"Couldn't save with format '{kind}'!")
            buffer = BytesIO()
            image.save(buffer, format=kind)
            data = buffer.getvalue()
        with open(filename, "wb") as f:
            f.write(data)


class SyntheticTikzDocument(TikzDocument):
    """TeX-free pseudo compiler: outcome is a pure function of the code.
    hash % 8 == 0 -> fatal (no output), error line = 1 + hash2 % (#lines); hash % 8 == 1 ->
    recoverable error but rasterisable; otherwise clean.  The raster is a deterministic drawing
    seeded by the hash, so SelfSim-style rewards are reproducible."""

    def _digest(self) -> int:
        return int.from_bytes(hashlib.blake2b(self.code.encode(), digest_size=8).digest(), "little")

    def _compile(self) -> Output:
        h = self._digest()
        n_lines = max(1, self.code.count("\n"))
        kind = h % 8
        if kind in (0, 1):
            line = 1 + (h >> 8) % n_lines
            log = f"(/tmp/synthetic.tex\n/tmp/synthetic.tex:{line}: Undefined control sequence.\n"
            return Output(pdf=("synthetic", h) if kind == 1 else None, status=12, log=log)
        return Output(pdf=("synthetic", h), status=0, log="")

    def rasterize(self, size: int = 420, expand_to_square: bool = True, **_) -> Optional[Image.Image]:
        pdf = self.pdf
        if not pdf:
            return None
        h = pdf[1]
        img = Image.new("RGB", (size, size), "white")
        d = ImageDraw.Draw(img)
        for k in range(6):
            v = (h >> (k * 9)) & 0x1FF
            x0, y0 = (v * 7) % size, (v * 13) % size
            x1, y1 = (v * 29 + 40) % size, (v * 31 + 90) % size
            d.line([x0, y0, x1, y1], fill="black", width=3)
        return img


class SleepingSyntheticTikzDocument(SyntheticTikzDocument):
    """SyntheticTikzDocument whose compile takes DTK_SYNTH_COMPILE_SECONDS of wall time (a sleep: like the wait for a latexmk
    child process it holds neither the GIL nor a core) — bench.py --reward-latency measures what a reward that costs what
    LaTeX costs (1-60 s, reference infer/tikz.py:89-147) does to rollouts/s.  Module level and configured through the
    environment so that spawned compile-pool workers build the same class."""

    def _compile(self) -> Output:
        import os
        import time
        time.sleep(float(os.environ.get("DTK_SYNTH_COMPILE_SECONDS", "0") or 0))
        return super()._compile()
