"""Writes libsamplerate's two in-tree SINC coefficient tables as DATA files (little-endian: int32 increment, int32 count,
count x float32), read from the reference where it lies.  Run in the build container only:

    python tests/golden/make_sinc_tables.py

Sources: thirdparty/libsamplerate/src/fastest_coeffs.h (SRC_SINC_FASTEST, 2 464 coefficients, increment 128) and
mid_qual_coeffs.h (SRC_SINC_MEDIUM_QUALITY, 22 438, increment 491).  high_qual_coeffs.h (SRC_SINC_BEST_QUALITY) is a
missing blob of the reference checkout (.MISSING_LARGE_BLOBS:3) and has no file here.  The numbers are parsed as the C
compiler parses them (decimal -> double -> float).
"""
import pathlib
import re
import struct

import numpy as np

REF = pathlib.Path("/root/reference/thirdparty/libsamplerate/src")
OUT = pathlib.Path(__file__).resolve().parents[2] / "godot-whisper_amd" / "csrc" / "data"

NUM = re.compile(r"^\s*([-+]?\d+\.\d+(?:[eE][-+]?\d+)?|[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?)\s*,?\s*(?:/\*.*\*/)?\s*$")


def table(header: str, count: int):
    txt = (REF / header).read_text().splitlines()
    start = next(i for i, l in enumerate(txt) if l.strip().startswith("{") and l.strip()[1:].strip().rstrip(",").isdigit())
    increment = int(txt[start].strip()[1:].strip().rstrip(","))
    vals = []
    for l in txt[start + 1:]:
        s = l.strip()
        if s.startswith("{"):
            s = s[1:]
        if s.startswith("}"):
            break
        m = NUM.match(s)
        if m:
            vals.append(float(m.group(1)))
    assert len(vals) == count, (header, len(vals), count)
    return increment, np.asarray(vals, np.float64).astype(np.float32)


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    for header, count, name in (("fastest_coeffs.h", 2464, "sinc_fastest.bin"), ("mid_qual_coeffs.h", 22438, "sinc_medium.bin")):
        inc, c = table(header, count)
        assert c[-1] == 0.0 and 0.8 < c[0] < 1.0 and c[1] < c[0]
        (OUT / name).write_bytes(struct.pack("<ii", inc, count) + c.astype("<f4").tobytes())
        print(name, inc, count, float(c[0]), float(c[1]))


if __name__ == "__main__":
    main()
