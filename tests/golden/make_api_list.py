#!/usr/bin/env python3
"""List the functions the reference's public header declares (W/whisper.h, v1.5.4) -> whisper_h_api.txt, one name per line.

The list is the coverage contract of the drop-in boundary: tests/test_abi.py requires every name to be declared in
include/whisper_mi355.h and exported by libwhisper_mi355.so.  Run in the build container only (needs /root/reference
and gcc); the fixture is data (symbol names), not source text.
"""
import pathlib, re, subprocess, sys

HDR = pathlib.Path("/root/reference/thirdparty/whisper.cpp/whisper.h")
OUT = pathlib.Path(__file__).resolve().parent / "whisper_h_api.txt"


def main():
    text = subprocess.run(["gcc", "-E", "-I", str(HDR.parent), str(HDR)], check=True, capture_output=True, text=True).stdout
    text = " ".join(l for l in text.splitlines() if not l.startswith("#"))
    names = sorted(set(re.findall(r"\b(whisper_[a-z0-9_]+)\s*\(", text)))
    OUT.write_text("\n".join(names) + "\n")
    print(len(names), "functions ->", OUT.name)


if __name__ == "__main__":
    sys.exit(main())
