"""sha1 over the kernel sources of this directory (*.hip, *.hpp: file name + content, in sorted order), first 16 hex digits.
The Makefile bakes it into libpcad.so (pcad_build_hash); plantcaduceus_amd.engine / bench.py compare it with the sources."""
import hashlib
import os
import sys


def source_hash(d: str) -> str:
    h = hashlib.sha1()
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".hpp")):
            h.update(fn.encode())
            with open(os.path.join(d, fn), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash(sys.argv[1] if len(sys.argv) > 1 else os.path.dirname(os.path.abspath(__file__))))
