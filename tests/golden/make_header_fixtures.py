#!/usr/bin/env python3
"""Extract the DATA held by the reference's own test models (W/models/for-tests-ggml-*.bin,
used by W/tests/CMakeLists.txt:15-76) into small fixtures:

  mel_filters_80.f32   80x201 float32 mel filterbank  (file section: W/whisper.cpp:1194-1203)
  vocab_en.bin.gz      vocab section of the *.en models   (i32 n, then n x {u32 len, bytes})
  vocab_multi.bin.gz   vocab section of the multilingual models (W/whisper.cpp:1206-1235)

Run in the build container only (needs /root/reference).  The fixtures are data (filter
coefficients and token byte strings), not source text.
"""
import gzip, struct, sys, pathlib
import numpy as np

REF = pathlib.Path("/root/reference/thirdparty/whisper.cpp/models")
OUT = pathlib.Path(__file__).resolve().parent


def split(path):
    b = path.read_bytes()
    assert struct.unpack_from("<I", b, 0)[0] == 0x67676D6C
    off = 4 + 11 * 4
    n_mel, n_fft = struct.unpack_from("<2i", b, off)
    off += 8
    filt = np.frombuffer(b, dtype="<f4", count=n_mel * n_fft, offset=off).copy()
    off += 4 * n_mel * n_fft
    v0 = off
    (nv,) = struct.unpack_from("<i", b, off)
    off += 4
    for _ in range(nv):
        (ln,) = struct.unpack_from("<I", b, off)
        off += 4 + ln
    assert off == len(b), "for-tests models carry no tensors"
    return filt.reshape(n_mel, n_fft), b[v0:off]


def main():
    f_en, v_en = split(REF / "for-tests-ggml-base.en.bin")
    f_ml, v_ml = split(REF / "for-tests-ggml-base.bin")
    assert np.array_equal(f_en, f_ml)
    f_en.astype("<f4").tofile(OUT / "mel_filters_80.f32")
    for name, blob in (("vocab_en.bin.gz", v_en), ("vocab_multi.bin.gz", v_ml)):
        with gzip.GzipFile(OUT / name, "wb", compresslevel=9, mtime=0) as g:
            g.write(blob)
    print("wrote", [p.name for p in OUT.iterdir()])


if __name__ == "__main__":
    sys.exit(main())
