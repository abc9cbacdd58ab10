// GEMM lab, round 4: the eight-wavefront ping-pong kernel (csrc/k_gemm8.hip) against the shipping k_gemm on the lock-step encoder's
// shapes — bit-identity of every output element and microseconds per launch.
// build: hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -I../../godot-whisper_amd/csrc gemm8_lab.hip -o gemm8_lab
#define WMI_G8_LAB 1
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
        {12000, 1536,  512, EPI_F16_BIAS_GELU,  "qkv-shaped x8 (GELU epilogue)"},
        { 1500, 2048,  512, EPI_F16_BIAS_GELU,  "mlp.0 x1 (GELU)"},
        {24000, 2048,  512, EPI_F16_BIAS_GELU,  "mlp.0 x16 (GELU)"},
        {12000, 2048,  512, EPI_F16_BIAS,       "mlp.0-shaped x8, f16 + bias only"},
        {12000, 1536,  512, EPI_QKV_ENC,        "q|k|v^T x8 (the encoder's own epilogue)"},
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
            if (s.epi == EPI_QKV_ENC) {            // q -> C, k -> X, V^T [chunk][S][Tpad] -> X + 16 MiB (all f16)
                a.S = 512; a.ldc = 512; a.aux = X; a.ldaux = 512; a.aux2 = X + (16u << 20); a.ldaux2 = 1504; a.rows_per_chunk = 1500;
                a.chunk_stride_aux2 = (int64_t) 512 * 1504;
            }
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
        for (int bm : {96, 128, 160, 192, 256, 288}) for (int ks : {64, 32}) for (int sw = 1; sw >= 1; --sw) {   // 2xx: deferred stores
            if (getenv("LAB_SET")) {                      // "bm:ks,bm:ks,..."
                char key[32]; snprintf(key, sizeof(key), "%d:%d", bm, ks);
                const char * set = getenv("LAB_SET"); const char * f = strstr(set, key);
                bool hit = false;
                while (f) { const char e = f[strlen(key)]; if ((f == set || f[-1] == ',') && (e == 0 || e == ',')) { hit = true; break; } f = strstr(f + 1, key); }
                if (!hit) continue;
            }
            if (getenv("LAB_BM") && atoi(getenv("LAB_BM")) != bm) continue;
            if (getenv("LAB_KS") && atoi(getenv("LAB_KS")) != ks) continue;
            CK(hipMemset(dC1, 0, csz)); CK(hipMemset(dX1, 0, csz));
            GemmArgs a1 = args(dC1, dX1); if (getenv("LAB_FLAGS")) a1.no_glds = atoi(getenv("LAB_FLAGS"));
            if (!gemm8(s.epi, bm, sw != 0, a1, st, ks)) { printf("    gemm8 bm=%3d ks=%d: not served\n", bm, ks); continue; }
            CK(hipStreamSynchronize(st)); CK(hipGetLastError());
            CK(hipMemcpy(c1.data(), dC1, csz, hipMemcpyDeviceToHost)); CK(hipMemcpy(x1.data(), dX1, csz, hipMemcpyDeviceToHost));
            const bool same = memcmp(c0.data(), c1.data(), csz) == 0 && memcmp(x0.data(), x1.data(), csz) == 0;
            size_t nbad = 0; if (!same) for (size_t i = 0; i < csz; ++i) nbad += c0[i] != c1[i] || x0[i] != x1[i];
            const double t1 = time_it([&]() { gemm8(s.epi, bm, sw != 0, a1, st, ks); }, iters);
            const int tiles = ((s.M + bm - 1) / bm) * (s.N / 256);
            printf("    gemm8 bm=%3d ks=%d %5d tiles (%.2f rounds) %8.2f us %7.1f TF/s  %s", bm, ks, tiles, tiles / 256.0, t1, flop / t1 / 1e6,
                   same ? "bit-identical\n" : "DIFFERS\n");
            if (!same) printf("        %zu differing bytes\n", nbad);
            if (getenv("LAB_PROBE")) {
                // per-workgroup stamps of one launch (wave 0): entry, first tile landed, K loop done, stores left; and the ablations
                const int cap = tiles < 256 ? ((tiles + 7) & ~7) : 256;
                unsigned long long * dp; CK(hipMalloc(&dp, (size_t) cap * 5 * 8)); CK(hipMemset(dp, 0, (size_t) cap * 5 * 8));
                GemmArgs ap = a1; ap.probe = dp;
                gemm8(s.epi, bm, sw != 0, ap, st, ks); CK(hipStreamSynchronize(st));
                std::vector<unsigned long long> h((size_t) cap * 5); CK(hipMemcpy(h.data(), dp, h.size() * 8, hipMemcpyDeviceToHost)); hipFree(dp);
                unsigned long long tmin = ~0ull, tmax = 0; double f = 0, l = 0, e = 0; std::vector<double> starts;
                for (int i = 0; i < cap; ++i) { const unsigned long long * q = &h[(size_t) i * 5]; tmin = std::min(tmin, q[0]); tmax = std::max(tmax, q[3]); }
                double tot = 0;
                for (int i = 0; i < cap; ++i) { const unsigned long long * q = &h[(size_t) i * 5]; f += q[1] - q[0]; l += q[2] - q[1]; e += q[4] - q[2]; tot += q[3] - q[0]; starts.push_back((q[0] - tmin) * 0.01); }
                std::sort(starts.begin(), starts.end());
                printf("        probe (wave 0): span %.2f us; per workgroup: entry -> first tile %.2f, first tile's K loop %.2f, its epilogue %.2f, whole life %.2f us; starts p50 %.2f max %.2f\n",
                       (tmax - tmin) * 0.01, f / cap * 0.01, l / cap * 0.01, e / cap * 0.01, tot / cap * 0.01, starts[cap / 2], starts.back());
                {   // slot accounting: cycles per wave in LOAD work / barrier behind LOAD / MFMA work / vmcnt wait / barrier behind MFMA
                    unsigned long long * dq; const size_t nq = (size_t) cap * 5 + (size_t) cap * 8 * 5;
                    CK(hipMalloc(&dq, nq * 8)); CK(hipMemset(dq, 0, nq * 8));
                    GemmArgs aq = a1; aq.probe = dq; aq.no_glds = 2048;
                    gemm8(s.epi, bm, sw != 0, aq, st, ks); CK(hipStreamSynchronize(st));
                    std::vector<unsigned long long> hq(nq); CK(hipMemcpy(hq.data(), dq, nq * 8, hipMemcpyDeviceToHost)); hipFree(dq);
                    double sum[2][5] = {{0}}; 
                    for (int w = 0; w < cap; ++w) for (int v = 0; v < 8; ++v) for (int i = 0; i < 5; ++i) sum[v >> 2][i] += (double) hq[(size_t) cap * 5 + ((size_t) w * 8 + v) * 5 + i];
                    const double nsteps = (double) s.K / 64 * tiles / cap;      // K steps per workgroup (average)
                    for (int g = 0; g < 2; ++g)
                        printf("        slots, group %d (cycles per K step and wave): LOAD work %.0f, barrier behind LOAD %.0f, MFMA work %.0f, vmcnt wait %.0f, barrier behind MFMA %.0f\n", g,
                               sum[g][0] / (cap * 4) / nsteps, sum[g][1] / (cap * 4) / nsteps, sum[g][2] / (cap * 4) / nsteps, sum[g][3] / (cap * 4) / nsteps, sum[g][4] / (cap * 4) / nsteps);
                }
                {   // a quarter of the CUs (64 workgroups walking the whole tile list): per-tile epilogue time when few CUs store at once
                    unsigned long long * dq; CK(hipMalloc(&dq, (size_t) 64 * 5 * 8)); CK(hipMemset(dq, 0, (size_t) 64 * 5 * 8));
                    GemmArgs aq = a1; aq.probe = dq; aq.no_glds = 4096;
                    gemm8(s.epi, bm, sw != 0, aq, st, ks); CK(hipStreamSynchronize(st));
                    std::vector<unsigned long long> hq((size_t) 64 * 5); CK(hipMemcpy(hq.data(), dq, hq.size() * 8, hipMemcpyDeviceToHost)); (void) hipFree(dq);
                    double l = 0, e = 0; for (int i = 0; i < 64; ++i) { l += hq[i * 5 + 2] - hq[i * 5 + 1]; e += hq[i * 5 + 4] - hq[i * 5 + 2]; }
                    printf("        64 workgroups only: first tile's K loop %.2f us, its epilogue %.2f us\n", l / 64 * 0.01, e / 64 * 0.01);
                }
                for (int flags : {4, 4 | 8, 4 | 16, 4 | 8 | 16, 4 | 8 | 32, 4 | 8 | 16 | 32}) {
                    GemmArgs af = a1; af.no_glds = flags;
                    const double tf = time_it([&]() { gemm8(s.epi, bm, sw != 0, af, st, ks); }, iters);
                    printf("        ablation%s%s%s%s%s%s: %8.2f us\n", flags & 4 ? " -epilogue" : "", flags & 8 ? " -dma" : "", flags & 16 ? " -mfma" : "", flags & 32 ? " -ldsreads" : "", flags & 64 ? " (no setprio on MFMA)" : "", "", tf);
                }
            }
        }
        hipFree(dA); hipFree(dW); hipFree(db); hipFree(dC0); hipFree(dC1); hipFree(dX0); hipFree(dX1); hipFree(dR);
    }
    return 0;
}
