"""
TikzDocument — the unchanged CPU reward back-end (row a·X): compile with latexmk, keep the last
page, crop, rasterise, parse `file:line:error` diagnostics.  Same behaviour as reference
detikzify/infer/tikz.py:21-168 (engines fallback :28,111-131; error map :54-73; 420 px raster
:149-156).  TeX Live / ghostscript / poppler / pymupdf / pdf2image / pdfCropMargins are external
and absent in this environment: they are imported lazily and a missing toolchain degrades exactly
like the reference does (:141-142: log an error, status -1, nothing rasterisable).

SyntheticTikzDocument is a TeX-free stand-in with the same interface (deterministic pseudo
compile result derived from a hash of the code) used by tests and by bench.py's stub reward —
never mixed with real-reward numbers.
"""
from __future__ import annotations

import hashlib
import logging
from collections import namedtuple
from functools import cached_property
from io import BytesIO
from os import environ
from os.path import isfile, join
from re import MULTILINE, escape, findall, search
from subprocess import DEVNULL, CalledProcessError, TimeoutExpired
from tempfile import NamedTemporaryFile, TemporaryDirectory
from typing import Dict, Optional, Union

from PIL import Image, ImageDraw

from ..util import check_output, expand

logger = logging.getLogger("detikzify_amd")

Output = namedtuple("Output", ["pdf", "status", "log"], defaults=[None, -1, ""])


class TikzDocument:
    engines = ["pdflatex", "lualatex", "xelatex"]
    Output = Output

    def __init__(self, code: str, timeout: Optional[int] = 60):
        self.code = code
        self.timeout = timeout
        self._compiled: Optional[Output] = None

    # ---- compile results (memoised per document, reference :31) --------------------------------
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
        """{line: message}; line 0 collects errors that cannot be located in the root file."""
        if not self.compiled_with_errors:
            return {}
        root = None
        if m := search(r"^\((.+)$", self.log, MULTILINE):
            root = m.group(1)
        found: Dict[int, str] = {}
        for file, line, msg in findall(r"^(.+):(\d+):(.+)$", self.log, MULTILINE):
            found[int(line) if file == root else 0] = msg.strip()
        return found or {0: "Fatal error occurred, no output PDF file produced!"}

    @cached_property
    def is_rasterizable(self) -> bool:
        return self.rasterize() is not None

    @cached_property
    def has_content(self) -> bool:
        img = self.rasterize()
        return img is not None and img.getcolors(1) is None

    @classmethod
    def set_engines(cls, engines: Union[str, list]):
        cls.engines = [engines] if isinstance(engines, str) else engines

    def _compile(self) -> Output:
        result: dict = {}
        try:
            import pymupdf
            from pdfCropMargins import crop
        except ImportError as e:  # same degradation as a missing TeX Live
            logger.error("Missing dependencies: %s (TeX Live, ghostscript, poppler needed)", e)
            return Output(**result)
        with TemporaryDirectory() as tmpdir:
            with NamedTemporaryFile(dir=tmpdir, buffering=0) as tmp:
                lines = self.code.split("\n")
                # no page numbers in the compiled pdf (they would defeat cropping)
                lines.insert(1, r"{cmd}\AtBeginDocument{{{cmd}}}".format(cmd=r"\thispagestyle{empty}\pagestyle{empty}"))
                tmp.write("\n".join(lines).encode())
                try:
                    best_line, tmppdf, outpdf = -1, f"{tmp.name}.pdf", join(tmpdir, "tikz.pdf")
                    open(f"{tmp.name}.bbl", "a").close()

                    def keep_last_page():
                        try:
                            doc = pymupdf.open(tmppdf)
                            doc.select([len(doc) - 1])
                            doc.save(outpdf)
                        except Exception:
                            pass

                    for engine in self.engines:
                        try:
                            check_output(
                                cwd=tmpdir, timeout=self.timeout, stderr=DEVNULL,
                                env=environ | dict(max_print_line="1000"),
                                args=["latexmk", "-f", "-nobibtex", "-norc", "-file-line-error",
                                      "-interaction=nonstopmode", f"-{engine}", tmp.name])
                        except (CalledProcessError, TimeoutExpired) as proc:
                            log = (getattr(proc, "output", b"") or b"").decode(errors="ignore")
                            err = search(rf"^{escape(tmp.name)}:(\d+):.+$", log, MULTILINE)
                            line = int(err.group(1)) if err else 0
                            if line > best_line:  # keep the engine that got furthest
                                best_line = line
                                result.update(status=getattr(proc, "returncode", -1), log=log)
                                keep_last_page()
                        else:
                            result.update(status=0, log="")
                            keep_last_page()
                            break
                    cropped = f"{tmp.name}.crop"
                    crop(["-gsf", "-c", "gb", "-p", "0", "-a", "-1", "-o", cropped, outpdf], quiet=True)
                    if isfile(cropped):
                        result["pdf"] = pymupdf.open(cropped)
                except FileNotFoundError:
                    logger.error("Missing dependencies: Did you install TeX Live?")
                except RuntimeError:
                    pass
        if result.get("status") == 0 and not result.get("pdf"):
            logger.warning("Could compile document but something seems to have gone wrong during cropping!")
        return Output(**result)

    def rasterize(self, size: int = 420, expand_to_square: bool = True, **_) -> Optional[Image.Image]:
        pdf = self.pdf
        if not pdf:
            return None
        from pdf2image.pdf2image import convert_from_bytes
        image = convert_from_bytes(pdf.tobytes(), size=size, single_file=True)[0]
        return expand(image, size) if expand_to_square else image

    def save(self, filename: str, *args, **kwargs):
        ext = filename.rsplit(".", 1)[-1]
        if ext == "tex":
            content = self.code.encode()
        elif ext == "pdf" and self.pdf:
            content = self.pdf.tobytes()
        elif (img := self.rasterize(*args, **kwargs)) is not None:
            buf = BytesIO()
            img.save(buf, format=ext)
            content = buf.getvalue()
        else:
            raise ValueError(f"Couldn't save with format '{ext}'!")
        with open(filename, "wb") as f:
            f.write(content)


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
