"""Stand-alone PNG writer process of run_aug (started with subprocess by saspa_aug_amd.run_aug._PngWriters; imports
nothing but numpy / Pillow).  Protocol on stdin, repeated until EOF: one header line "H W C <utf-8 path>\\n", then
H*W*C raw bytes (u8, C-contiguous).  Exit code 0 when every image was written."""
import sys

import numpy as np
from PIL import Image


def main():
    inp = sys.stdin.buffer
    while True:
        header = inp.readline()
        if not header:
            return 0
        h, w, c, path = header.decode("utf-8").rstrip("\n").split(" ", 3)
        h, w, c = int(h), int(w), int(c)
        n = h * w * c
        buf = inp.read(n)
        if len(buf) != n:
            return 2
        arr = np.frombuffer(buf, np.uint8).reshape((h, w, c) if c > 1 else (h, w))
        Image.fromarray(arr).save(path)


if __name__ == "__main__":
    sys.exit(main())
