"""check_output that kills the whole process group on timeout (latexmk spawns children;
reference detikzify/util/subprocess.py:8-48, pinned by tests/golden/subprocess.json)."""
from __future__ import annotations

import os
import signal
import subprocess
from typing import Optional


def _kill_group(proc: subprocess.Popen):
    try:
        os.killpg(os.getpgid(proc.pid), signal.SIGKILL)
    except ProcessLookupError:      # it has just exited by itself (bpo-40550)
        pass


def check_output(args, timeout: Optional[float] = None, **popen_kwargs) -> bytes:
    """stdout of `args`; CalledProcessError (with .output) on a non-zero exit; on a timeout the process AND its
    children are killed and the TimeoutExpired of communicate() is passed on as it is (.output = what had been read)"""
    popen_kwargs.setdefault("stdout", subprocess.PIPE)
    with subprocess.Popen(args, start_new_session=True, **popen_kwargs) as proc:
        try:
            out, _ = proc.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            _kill_group(proc)
            proc.wait()
            raise
        except BaseException:
            _kill_group(proc)
            raise
        if proc.returncode:
            raise subprocess.CalledProcessError(proc.returncode, proc.args, output=out)
    return out
