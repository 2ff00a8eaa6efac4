"""Identity of the native sources a measurement belongs to.

bench.py quotes counters (HBM bytes, instruction counts per launch) that come from separate
rocprofv3 passes; this id ties the two together: profiles/summarize.py stores it in the PMC
record, bench.py compares it with the tree it runs from and refuses to derive a roofline
fraction from a record of other sources (`"pmc_stale": true`).  A hash of the sources, not of
the .so: a rebuild of the same sources must not look like another build.
"""
import glob
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def source_files():
    root = os.path.dirname(_HERE)
    pats = [os.path.join(_HERE, "csrc", p) for p in ("*.hip", "*.cpp", "*.h", "Makefile")]
    pats.append(os.path.join(root, "include", "*.h"))
    return sorted(f for p in pats for f in glob.glob(p))


def build_id():
    """16 hex digits over the names and contents of csrc/*.{hip,cpp,h}, its Makefile and include/*.h."""
    h = hashlib.sha256()
    for f in source_files():
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(build_id())
