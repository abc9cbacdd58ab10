// TEST INFRASTRUCTURE — C accessors to the reference's own W/examples/common.cpp (compiled in place by `make ref`).
// Used only to pin oracle/host_dsp.c's high-pass filter and VAD (tests/test_oracle_host_dsp.py).
#include "common.h"
#include <vector>

extern "C" {

void ref_high_pass_filter(float * data, int n, float cutoff, float sample_rate) {
    std::vector<float> v(data, data + n);
    high_pass_filter(v, cutoff, sample_rate);
    for (int i = 0; i < n; i++) data[i] = v[i];
}

// returns vad_simple's answer; `data` receives the filtered samples
int ref_vad_simple(float * data, int n, int sample_rate, int last_ms, float vad_thold, float freq_thold) {
    std::vector<float> v(data, data + n);
    const bool r = vad_simple(v, sample_rate, last_ms, vad_thold, freq_thold, false);
    for (int i = 0; i < n; i++) data[i] = v[i];
    return r ? 1 : 0;
}

}
