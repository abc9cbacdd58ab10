"""TEST INFRASTRUCTURE — checkers for the hot path.  Never imported by the product.

  reflib.RefWhisper   the reference's own CPU path compiled from /root/reference into
                      oracle/_ref/libwhisper_ref.so (`make -C oracle ref`), driven via ctypes.
  port.PortWhisper    this repository's CPU restatement (oracle/whisper_port.cpp).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use these.
"""
