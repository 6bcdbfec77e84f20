#!/usr/bin/env python3
"""sha256 over the PRODUCT's sources (everyvoice_amd/**/*.py, csrc, include/*.h: path + bytes, sorted): the same on every checkout
of the same code, unchanged by commits that only touch docs, profiles, tests or tools.  bench.py prints it as ``code`` and the
counter summaries record it (``*.meta.json: code``), so a reader can see that `traffic` was collected on the code the line timed
even though committing the summaries makes a new HEAD (traffic_commit != commit on every leg)."""
import hashlib
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def code_fingerprint(root: Path = ROOT) -> str:
    files = [p for pat in ("everyvoice_amd/**/*.py", "everyvoice_amd/csrc/*", "include/*.h") for p in root.glob(pat) if p.is_file() and "__pycache__" not in p.parts]
    h = hashlib.sha256()
    for p in sorted(files, key=lambda p: p.relative_to(root).as_posix()):
        h.update(p.relative_to(root).as_posix().encode())
        h.update(b"\0")
        h.update(p.read_bytes())
        h.update(b"\0")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(code_fingerprint(Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT))
