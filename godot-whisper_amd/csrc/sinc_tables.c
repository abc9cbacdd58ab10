/* libsamplerate's SINC coefficient tables as binary data (data/sinc_*.bin: int32 increment, int32 count, count x float32,
 * written from the reference's fastest_coeffs.h / mid_qual_coeffs.h by tests/golden/make_sinc_tables.py), placed in .rodata. */
__asm__(".section .rodata\n"
        ".balign 16\n"
        ".global wmi_sinc_fastest_bin\n"
        "wmi_sinc_fastest_bin:\n"
        ".incbin \"data/sinc_fastest.bin\"\n"
        ".balign 16\n"
        ".global wmi_sinc_medium_bin\n"
        "wmi_sinc_medium_bin:\n"
        ".incbin \"data/sinc_medium.bin\"\n"
        ".previous\n");
