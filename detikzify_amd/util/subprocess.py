"""check_output that kills the whole process group on timeout (latexmk spawns children;
reference detikzify/util/subprocess.py:8-48)."""
from __future__ import annotations

import os
import signal
import subprocess
from typing import Optional


def check_output(args, timeout: Optional[float] = None, **popen_kwargs) -> bytes:
    popen_kwargs.setdefault("stdout", subprocess.PIPE)
    proc = subprocess.Popen(args, start_new_session=True, **popen_kwargs)
    try:
        out, _ = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(os.getpgid(proc.pid), signal.SIGKILL)
        except ProcessLookupError:
            pass
        out, _ = proc.communicate()
        raise subprocess.TimeoutExpired(args, timeout, output=out)
    except BaseException:
        try:
            os.killpg(os.getpgid(proc.pid), signal.SIGKILL)
        except ProcessLookupError:
            pass
        proc.wait()
        raise
    if proc.returncode:
        raise subprocess.CalledProcessError(proc.returncode, args, output=out)
    return out
