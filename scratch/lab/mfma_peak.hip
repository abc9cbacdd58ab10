#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float    f4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k_peak(float * out, int iters) {
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16) (threadIdx.x * 0.001f + e); b[e] = (_Float16) (e * 0.5f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    float t = 0;
    for (int i = 0; i < NACC; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (t == 12345.678f) out[threadIdx.x] = t;
}
// LDS read + MFMA, no barriers: each wave reads its own fragments
template <int WPS>
__global__ __launch_bounds__(256) void k_lds_mfma(float * out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768 / 4; i += 256) ((float *) smem)[i] = 0.001f * i;
    __syncthreads();
    f4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f4{0, 0, 0, 0};
    const int frow = lane & 15, fq = lane >> 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            h8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = *(const h8 *) (smem + ((wave >> 1) * 64 + i * 16 + frow) * 128 + (((kk * 4 + fq) ^ (frow & 7)) << 4));
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = *(const h8 *) (smem + 16384 + ((wave & 1) * 64 + j * 16 + frow) * 128 + (((kk * 4 + fq) ^ (frow & 7)) << 4));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (WPS) __syncthreads();
    }
    float t = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 12345.678f) out[threadIdx.x] = t;
}
int main() {
    float * d; hipMalloc(&d, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto T = [&](const char * name, auto && launch, double mfma_per_wave, int waves) {
        launch(); launch();
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        const double flop = mfma_per_wave * waves * 16384.0;
        printf("%-34s %8.1f us  %8.1f TF/s\n", name, ms * 1000, flop / (ms * 1e-3) / 1e12);
    };
    const int iters = 2000;
    for (int bpc = 1; bpc <= 4; bpc *= 2) {
        char nm[64];
        snprintf(nm, 64, "mfma only, 16 acc, %d blk/CU", bpc);
        T(nm, [&]() { hipLaunchKernelGGL((k_peak<16>), dim3(256 * bpc), dim3(256), 0, 0, d, iters); }, 16.0 * iters, 256 * bpc * 4);
        snprintf(nm, 64, "mfma only, 4 acc, %d blk/CU", bpc);
        T(nm, [&]() { hipLaunchKernelGGL((k_peak<4>), dim3(256 * bpc), dim3(256), 0, 0, d, iters); }, 4.0 * iters, 256 * bpc * 4);
    }
    for (int bpc = 1; bpc <= 4; ++bpc) {
        char nm[64];
        snprintf(nm, 64, "lds+mfma no barrier, %d blk/CU", bpc);
        T(nm, [&]() { hipLaunchKernelGGL((k_lds_mfma<0>), dim3(256 * bpc), dim3(256), 32768, 0, d, iters); }, 32.0 * iters, 256 * bpc * 4);
        snprintf(nm, 64, "lds+mfma + barrier, %d blk/CU", bpc);
        T(nm, [&]() { hipLaunchKernelGGL((k_lds_mfma<1>), dim3(256 * bpc), dim3(256), 32768, 0, d, iters); }, 32.0 * iters, 256 * bpc * 4);
    }
    return 0;
}
