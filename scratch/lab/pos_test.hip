// does k_gemm give position-independent results?  A = the same [1500][128] block repeated R times; compare the copies.
#include "../../godot-whisper_amd/csrc/k_gemm.hip"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace wmi::k;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main() {
    const int T = 1500, S = 128, L = 3, N = L * 2 * S, K = S;
    for (int R : {1, 3, 6}) {
        const int M = R * T;
        std::vector<__half> hA((size_t) M * K), hW((size_t) N * K); std::vector<float> hb(N);
        uint32_t seed = 777u;
        auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xffff) / 65536.0f - 0.5f; };
        for (int t = 0; t < T; ++t) for (int k = 0; k < K; ++k) { const __half v = __float2half(rnd() * 4.0f); for (int r = 0; r < R; ++r) hA[((size_t) r * T + t) * K + k] = v; }
        for (auto & v : hW) v = __float2half(rnd() * 0.2f);
        for (auto & v : hb) v = rnd();
        __half * dA, * dW, * dK, * dV; float * db;
        CK(hipMalloc(&dA, hA.size() * 2 + 65536)); CK(hipMalloc(&dW, hW.size() * 2 + 65536)); CK(hipMalloc(&db, N * 4));
        CK(hipMalloc(&dK, (size_t) L * M * S * 2)); CK(hipMalloc(&dV, (size_t) L * M * S * 2));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
        GemmArgs a{}; a.A = dA; a.lda = K; a.W = dW; a.ldw = K; a.M = M; a.N = N; a.K = K; a.bias = db;
        a.C = dK; a.ldc = S; a.aux = dV; a.ldaux = S; a.S = S; a.layer_stride = (int64_t) M * S; a.scale = powf(64.0f, -0.25f);
        gemm(EPI_CROSS_KV, a, 0);
        CK(hipDeviceSynchronize());
        std::vector<__half> k((size_t) L * M * S), v((size_t) L * M * S);
        CK(hipMemcpy(k.data(), dK, k.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(v.data(), dV, v.size() * 2, hipMemcpyDeviceToHost));
        static std::vector<__half> k0, v0;
        if (R == 1) { k0 = k; v0 = v; }
        size_t dk = 0, dv = 0;
        for (int l = 0; l < L; ++l) for (int r = 0; r < R; ++r) for (int t = 0; t < T; ++t) for (int c = 0; c < S; ++c) {
            const size_t i = ((size_t) l * M + (size_t) r * T + t) * S + c, i0 = ((size_t) l * T + t) * S + c;
            if (__half2float(k[i]) != __half2float(k0[i0])) ++dk;
            if (__half2float(v[i]) != __half2float(v0[i0])) ++dv;
        }
        printf("R=%d M=%d: elements differing from the R=1 result: k %zu, v %zu\n", R, M, dk, dv);
    }
    return 0;
}
