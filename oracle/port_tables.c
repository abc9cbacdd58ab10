/* TEST INFRASTRUCTURE — lookup tables of the CPU restatement.
 *
 * The reference evaluates GELU and exp through 65536-entry f16 tables that it fills at start-up
 * (W/ggml.c:2222-2235) from C code compiled with -O3 -mfma.  Whether a*b+c contracts to an FMA decides
 * the last bit of a handful of entries, so the tables are built here in a C translation unit compiled
 * with the same compiler, language standard and flags as the reference's ggml.c (oracle/Makefile);
 * tests/test_oracle_port.py checks them entry by entry against the compiled reference.
 */
#include <immintrin.h>
#include <math.h>
#include <stdint.h>

static const float GELU_A = 0.044715f;
static const float SQRT_2_PI = 0.79788456080286535587989211986876f;

static inline float gelu_f32(float x) { return 0.5f*x*(1.0f + tanhf(SQRT_2_PI*x*(1.0f + GELU_A*x*x))); }

void port_fill_tables(uint16_t * gelu, uint16_t * expt) {
    for (int i = 0; i < (1 << 16); ++i) {
        const float f = _cvtsh_ss((uint16_t) i);
        gelu[i] = _cvtss_sh(gelu_f32(f), 0);
        expt[i] = _cvtss_sh(expf(f), 0);
    }
}
