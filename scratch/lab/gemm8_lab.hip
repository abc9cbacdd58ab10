// GEMM lab, round 4: the eight-wavefront ping-pong kernel (csrc/k_gemm8.hip) against the shipping k_gemm on the lock-step encoder's
// shapes — bit-identity of every output element and microseconds per launch.
// build: hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -I../../godot-whisper_amd/csrc gemm8_lab.hip -o gemm8_lab
#include "../../godot-whisper_amd/csrc/k_gemm.hip"
#include "../../godot-whisper_amd/csrc/k_gemm8.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#include <algorithm>

using namespace wmi::k;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char ** argv) {
    struct Shape { int M, N, K, epi; const char * what; };
    const Shape shapes[] = {
        {12000, 2048,  512, EPI_F16_BIAS_GELU,  "mlp.0 x8 (GELU)"},
        {12000,  512, 2048, EPI_F32_BIAS_RESID, "mlp.2 x8 (resid)"},
        {12000,  512,  512, EPI_F32_BIAS_RESID, "out   x8 (resid)"},
        {12000, 6144,  512, EPI_CROSS_KV,       "cross x8"},
        {12000, 1536,  512, EPI_F16_BIAS,       "qkv-shaped x8 (f16+bias)"},
        { 1500, 2048,  512, EPI_F16_BIAS_GELU,  "mlp.0 x1 (GELU)"},
        {24000, 2048,  512, EPI_F16_BIAS_GELU,  "mlp.0 x16 (GELU)"},
    };
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    hipStream_t st; CK(hipStreamCreate(&st));
    int si = -1;
    for (const Shape & s : shapes) {
        ++si; if (only >= 0 && si != only) continue;
        const size_t nA = (size_t) s.M * s.K, nW = (size_t) s.N * s.K, nC = (size_t) s.M * s.N;
        std::vector<__half> hA(nA), hW(nW); std::vector<float> hb(s.N), hR(nC);
        uint32_t seed = 12345u + si;
        auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xffff) / 65536.0f - 0.5f; };
        for (auto & v : hA) v = __float2half(rnd() * 2.0f);
        for (auto & v : hW) v = __float2half(rnd() * 0.2f);
        for (auto & v : hb) v = rnd();
        for (auto & v : hR) v = rnd();
        const size_t csz = nC * 4;                      // f32 outputs are the largest
        __half * dA, * dW; float * db; unsigned char * dC0, * dC1, * dX0, * dX1; float * dR;
        CK(hipMalloc(&dA, nA * 2 + 4096)); CK(hipMalloc(&dW, nW * 2 + 4096)); CK(hipMalloc(&db, s.N * 4));
        CK(hipMalloc(&dC0, csz)); CK(hipMalloc(&dC1, csz)); CK(hipMalloc(&dX0, csz)); CK(hipMalloc(&dX1, csz)); CK(hipMalloc(&dR, csz));
        CK(hipMemcpy(dA, hA.data(), nA * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), nW * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, hb.data(), s.N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dR, hR.data(), csz, hipMemcpyHostToDevice));
        auto args = [&](unsigned char * C, unsigned char * X) {
            GemmArgs a{}; a.A = dA; a.lda = s.K; a.W = dW; a.ldw = s.K; a.M = s.M; a.N = s.N; a.K = s.K; a.bias = db; a.C = C; a.ldc = s.N;
            if (s.epi == EPI_F32_BIAS_RESID) { a.resid = dR; a.ldr = s.N; }
            if (s.epi == EPI_CROSS_KV) {           // columns [layer][K: S | V: S], outputs [layer][M][S] each
                a.S = 512; a.scale = 0.35355339f; a.aux = X; a.ldc = 512; a.ldaux = 512; a.layer_stride = (size_t) s.M * 512;
            }
            return a;
        };
        auto time_it = [&](auto && fn, int iters) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int i = 0; i < 3; ++i) fn();
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) fn();
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipGetLastError());
            return ms * 1000.0 / iters;
        };
        const double flop = 2.0 * s.M * s.N * s.K;
        const int iters = getenv("LAB_ITERS") ? atoi(getenv("LAB_ITERS")) : 40;
        CK(hipMemset(dC0, 0, csz)); CK(hipMemset(dX0, 0, csz));
        const GemmArgs a0 = args(dC0, dX0);
        const double t0 = time_it([&]() { gemm(s.epi, a0, st); }, iters);
        printf("%-26s M=%5d N=%5d K=%5d | shipping k_gemm %8.2f us %7.1f TF/s\n", s.what, s.M, s.N, s.K, t0, flop / t0 / 1e6);
        std::vector<unsigned char> c0(csz), x0(csz), c1(csz), x1(csz);
        CK(hipMemcpy(c0.data(), dC0, csz, hipMemcpyDeviceToHost)); CK(hipMemcpy(x0.data(), dX0, csz, hipMemcpyDeviceToHost));
        for (int bm : {96, 128, 192, 256}) for (int sw = 1; sw >= 0; --sw) {
            if (getenv("LAB_BM") && atoi(getenv("LAB_BM")) != bm) continue;
            if (getenv("LAB_SW") && atoi(getenv("LAB_SW")) != sw) continue;
            CK(hipMemset(dC1, 0, csz)); CK(hipMemset(dX1, 0, csz));
            const GemmArgs a1 = args(dC1, dX1);
            if (!gemm8(s.epi, bm, sw != 0, a1, st)) { printf("    gemm8 bm=%3d sw=%d: not served\n", bm, sw); continue; }
            CK(hipStreamSynchronize(st)); CK(hipGetLastError());
            CK(hipMemcpy(c1.data(), dC1, csz, hipMemcpyDeviceToHost)); CK(hipMemcpy(x1.data(), dX1, csz, hipMemcpyDeviceToHost));
            const bool same = memcmp(c0.data(), c1.data(), csz) == 0 && memcmp(x0.data(), x1.data(), csz) == 0;
            size_t nbad = 0; if (!same) for (size_t i = 0; i < csz; ++i) nbad += c0[i] != c1[i] || x0[i] != x1[i];
            const double t1 = time_it([&]() { gemm8(s.epi, bm, sw != 0, a1, st); }, iters);
            const int tiles = ((s.M + bm - 1) / bm) * (s.N / 256);
            printf("    gemm8 bm=%3d sw=%d %5d tiles (%.2f rounds) %8.2f us %7.1f TF/s  %s", bm, sw, tiles, tiles / 256.0, t1, flop / t1 / 1e6,
                   same ? "bit-identical\n" : "DIFFERS\n");
            if (!same) printf("        %zu differing bytes\n", nbad);
        }
        hipFree(dA); hipFree(dW); hipFree(db); hipFree(dC0); hipFree(dC1); hipFree(dX0); hipFree(dX1); hipFree(dR);
    }
    return 0;
}
