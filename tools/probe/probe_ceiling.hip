// Sustained-rate ablation ladder of the PRODUCT per-sample GEMM (text-to-sound-synthesis_amd/csrc/gemm_f16x2_ps.hip is included
// verbatim, once per PS_ABLATE value, each copy in its own namespace), with the board power and the shader clock sampled
// through rocm_smi while every variant runs back to back for a fixed time.  Answers two questions the round-2 review left
// open: (1) what the MFMA stream of THIS 8-wave program sustains when nothing else is in the loop (the ceiling a schedule
// change can approach), and (2) whether the launch time is set by the schedule or by the power budget (clock x busy).
//   bash tools/probe/build_probe_ceiling.sh      (applies tools/probe/ps_probe.patch -- the PS_ABLATE switches -- to a scratch copy
//                                                 of csrc/ and compiles this file against it; -> tools/probe/probe_ceiling)
// Not part of the product library.
#include <hip/hip_runtime.h>
#include <rocm_smi/rocm_smi.h>
#include <stdarg.h>
#include <stdlib.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

void ds_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    fputc('\n', stderr);
    va_end(ap);
}

#undef PS_ABLATE
#define PS_ABLATE 0
namespace v0 {
#include "../../text-to-sound-synthesis_amd/csrc/gemm_f16x2_ps.hip"
}
#undef PS_ABLATE
#define PS_ABLATE 8
namespace v8 {
#include "../../text-to-sound-synthesis_amd/csrc/gemm_f16x2_ps.hip"
}
#undef PS_ABLATE
#define PS_ABLATE 9
namespace v9 {
#include "../../text-to-sound-synthesis_amd/csrc/gemm_f16x2_ps.hip"
}
#undef PS_ABLATE
#define PS_ABLATE 11
namespace v11 {
#include "../../text-to-sound-synthesis_amd/csrc/gemm_f16x2_ps.hip"
}
#undef PS_ABLATE
#define PS_ABLATE 27
namespace v27 {
#include "../../text-to-sound-synthesis_amd/csrc/gemm_f16x2_ps.hip"
}
#undef PS_ABLATE
#define PS_ABLATE 12
namespace v12 {
#include "../../text-to-sound-synthesis_amd/csrc/gemm_f16x2_ps.hip"
}
#undef PS_ABLATE
#define PS_ABLATE 13
namespace v13 {
#include "../../text-to-sound-synthesis_amd/csrc/gemm_f16x2_ps.hip"
}
#undef PS_ABLATE
#define PS_ABLATE 14
namespace v14 {
#include "../../text-to-sound-synthesis_amd/csrc/gemm_f16x2_ps.hip"
}
#undef PS_ABLATE
#define PS_ABLATE 128
namespace v128 {
#include "../../text-to-sound-synthesis_amd/csrc/gemm_f16x2_ps.hip"
}
#undef PS_ABLATE
#define PS_ABLATE 136
namespace v136 {
#include "../../text-to-sound-synthesis_amd/csrc/gemm_f16x2_ps.hip"
}
#undef PS_ABLATE
#define PS_ABLATE 32
namespace v32 {
#include "../../text-to-sound-synthesis_amd/csrc/gemm_f16x2_ps.hip"
}
#undef PS_ABLATE
#define PS_ABLATE 64
namespace v64 {
#include "../../text-to-sound-synthesis_amd/csrc/gemm_f16x2_ps.hip"
}

#undef PS_ABLATE
#define PS_ABLATE 256
namespace v256 {
#include "../../text-to-sound-synthesis_amd/csrc/gemm_f16x2_ps.hip"
}
// tools/probe/prev/ (git-ignored): `git show <rev>:<path>` copies of the kernel + its epilogue, for a same-run A/B against
// an earlier revision
#if __has_include("prev/gemm_f16x2_ps.hip")
#undef PS_ABLATE
#define PS_ABLATE 0
namespace vprev {
#include "prev/gemm_f16x2_ps.hip"
}
#define PROBE_HAVE_PREV 1
#endif

// ---- telemetry ------------------------------------------------------------------------------------------------------
struct Telemetry {
    std::atomic<bool> run{false}, stop{false};
    double pw_sum[16] = {0}, clk_sum[16] = {0};
    long n[16] = {0};
    uint32_t ndev = 0;
    bool ok = false;
    std::thread th;
    void start() {
        if (rsmi_init(0) != RSMI_STATUS_SUCCESS) { printf("rocm_smi: init failed -- no power / clock telemetry\n"); return; }
        rsmi_num_monitor_devices(&ndev);
        if (ndev > 16) ndev = 16;
        ok = ndev > 0;
        for (uint32_t d = 0; d < ndev; ++d) {
            uint64_t cap = 0;
            rsmi_dev_power_cap_get(d, 0, &cap);
            printf("rocm_smi: device %u power cap %.0f W\n", d, cap * 1e-6);
        }
        th = std::thread([this] {
            while (!stop.load()) {
                if (run.load())
                    for (uint32_t d = 0; d < ndev; ++d) {
                        uint64_t pw = 0;
                        RSMI_POWER_TYPE ty;
                        if (rsmi_dev_power_get(d, &pw, &ty) != RSMI_STATUS_SUCCESS) rsmi_dev_current_socket_power_get(d, &pw);
                        rsmi_frequencies_t f;
                        memset(&f, 0, sizeof(f));
                        double mhz = 0;
                        if (rsmi_dev_gpu_clk_freq_get(d, RSMI_CLK_TYPE_SYS, &f) == RSMI_STATUS_SUCCESS && f.current < RSMI_MAX_NUM_FREQUENCIES)
                            mhz = f.frequency[f.current] * 1e-6;
                        pw_sum[d] += pw * 1e-6;
                        clk_sum[d] += mhz;
                        ++n[d];
                    }
                std::this_thread::sleep_for(std::chrono::milliseconds(5));
            }
        });
    }
    void begin() { for (int d = 0; d < 16; ++d) { pw_sum[d] = clk_sum[d] = 0; n[d] = 0; } run = true; }
    void end(double* watts, double* mhz) {
        run = false;
        std::this_thread::sleep_for(std::chrono::milliseconds(12));
        *watts = *mhz = 0;
        for (uint32_t d = 0; d < ndev; ++d)       // the busy device = the one drawing the most
            if (n[d] && pw_sum[d] / n[d] > *watts) { *watts = pw_sum[d] / n[d]; *mhz = clk_sum[d] / n[d]; }
    }
    void finish() { stop = true; if (th.joinable()) th.join(); if (ok) rsmi_shut_down(); }
};

static size_t packed_off(int row, int col, int ktiles) {
    return ((size_t)(row >> 4) * ktiles + (col >> 5)) * 512 + (row & 15) * 32 + ((((col >> 3) & 3) ^ ((row >> 2) & 3)) << 3) + (col & 7);
}

typedef int (*launch_fn)(const GemmParams&, hipStream_t);

int main(int argc, char** argv) {
    const double dur = argc > 1 ? atof(argv[1]) : 0.5;           // seconds per (variant, shape)
    const int B = argc > 2 ? atoi(argv[2]) : 64;                  // samples (64 = the benchmarked batch: whole rounds of the 256 CUs)
    const int L = 272, M = B * L, H = 16;
    const int Nmax = 4096, Kmax = 4096;
    const int zero_lo = getenv("PROBE_ZERO_LO") ? atoi(getenv("PROBE_ZERO_LO")) : 0;   // 1: lo planes zero, 2: everything zero
    // round 4: the lowest n mantissa bits of every lo-plane value cleared (the power experiment of DESIGN.md section 3:
    // how much of the clock the chip loses to full-mantissa lo operands comes back when a few of their bits stop toggling)
    const int mask_lo = getenv("PROBE_MASK_LO") ? atoi(getenv("PROBE_MASK_LO")) : 0;
    auto lo_of = [&](float a, _Float16 hi) {
        _Float16 lo = (_Float16)(a - (float)hi);
        if (mask_lo > 0) {
            unsigned short u;
            memcpy(&u, &lo, 2);
            u &= (unsigned short)(0xffffu << mask_lo);
            memcpy(&lo, &u, 2);
        }
        return lo;
    };
    // operands with the statistics of the denoiser: activations ~ N(0, 1), weights ~ N(0, 0.02) pre-scaled by 2^s so that
    // max |w| ~ 2^13 (what _lib.split_f16x2 does), both split hi + lo and stored as packed planes
    std::vector<_Float16> ha((size_t)2 * M * Kmax), hw((size_t)2 * Nmax * Kmax);
    unsigned long long st = 88172645463325252ull;
    auto gauss = [&]() {
        float a = 0.f;
        for (int q = 0; q < 4; ++q) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; a += (float)(st >> 40) / 16777216.f - 0.5f; }
        return a * 1.7320508f;
    };
    _Float16 *A, *W;
    hipMalloc(&A, ha.size() * 2); hipMalloc(&W, hw.size() * 2);
    float *C, *R, *bias;
    hipMalloc(&C, (size_t)M * Nmax * 4); hipMalloc(&R, (size_t)M * 1024 * 4); hipMalloc(&bias, Nmax * 4);
    hipMemset(R, 0, (size_t)M * 1024 * 4); hipMemset(bias, 0, Nmax * 4);
    _Float16 *img;
    hipMalloc(&img, (size_t)B * H * 4 * 288 * 64 * 2); hipMemset(img, 0, (size_t)B * H * 4 * 288 * 64 * 2);
    Telemetry tel;
    tel.start();
    struct Shape { const char* name; int N, K, epi; };
    const Shape shapes[] = {{"fc1  N=4096 K=1024 (GELU2 + planes)", 4096, 1024, 1}, {"proj N=1024 K=1024 (row + residual)", 1024, 1024, 0},
                            {"qkv  N=3072 K=1024 (Q / K / V^T)", 3072, 1024, 2}, {"fc2  N=1024 K=4096 (row + residual)", 1024, 4096, 0}};
    struct Var { const char* name; launch_fn fn; };
    const Var vars[] = {{"product kernel", v0::ds_launch_gemm_f16x2_ps},
                        {"epilogue: LDS staging only", v32::ds_launch_gemm_f16x2_ps},
                        {"epilogue: global ld/st only", v64::ds_launch_gemm_f16x2_ps},
                        {"no epilogue", v8::ds_launch_gemm_f16x2_ps},
                        {"no epilogue, no DMA", v9::ds_launch_gemm_f16x2_ps},
                        {"MFMA + barriers only", v11::ds_launch_gemm_f16x2_ps},
                        {"MFMA only", v27::ds_launch_gemm_f16x2_ps},
                        {"DMA + reads + barriers (no MFMA)", v12::ds_launch_gemm_f16x2_ps},
                        {"reads + barriers only", v13::ds_launch_gemm_f16x2_ps},
                        {"DMA + barriers only", v14::ds_launch_gemm_f16x2_ps},
                        {"product, builtin DMA form", v128::ds_launch_gemm_f16x2_ps},
                        {"no epilogue, builtin DMA form", v136::ds_launch_gemm_f16x2_ps},
                        {"product, residual requested in its own step", v256::ds_launch_gemm_f16x2_ps},
#ifdef PROBE_HAVE_PREV
                        {"previous revision (tools/probe/prev)", vprev::ds_launch_gemm_f16x2_ps},
#endif
    };
    int K_packed = 0;
    for (const Shape& sh : shapes) {
        const int N = sh.N, K = sh.K;
        if (K != K_packed) {     // the packed layout depends on K
            for (int r = 0; r < M; ++r)
                for (int k = 0; k < K; ++k) {
                    const float a = zero_lo == 2 ? 0.f : gauss();
                    const _Float16 hi = (_Float16)a;
                    const size_t o = packed_off(r, k, K / 32);
                    ha[o] = hi;
                    ha[(size_t)M * K + o] = zero_lo ? (_Float16)0.f : lo_of(a, hi);
                }
            for (int r = 0; r < Nmax; ++r)
                for (int k = 0; k < K; ++k) {
                    const float a = zero_lo == 2 ? 0.f : gauss() * 0.02f * 65536.f;
                    const _Float16 hi = (_Float16)a;
                    const size_t o = packed_off(r, k, K / 32);
                    hw[o] = hi;
                    hw[(size_t)Nmax * K + o] = zero_lo ? (_Float16)0.f : lo_of(a, hi);
                }
            hipMemcpy(A, ha.data(), (size_t)2 * M * K * 2, hipMemcpyHostToDevice);
            hipMemcpy(W, hw.data(), (size_t)2 * Nmax * K * 2, hipMemcpyHostToDevice);
            K_packed = K;
        }
        GemmParams p;
        memset(&p, 0, sizeof(p));
        p.A = (const float*)A; p.W = (const float*)W; p.bias = bias; p.C = C;
        p.M = M; p.N = N; p.K = K; p.lda = K; p.ldw = K; p.ldc = N; p.ldr = N; p.groups = 1;
        p.out_scale = 1.f / 65536.f; p.a_split = 1; p.a_plane = (long long)M * K; p.w3_plane = (long long)Nmax * K;
        p.rows_per_sample = L;
        if (sh.epi == 0) { p.store = DS_STORE_ROW; p.R = R; }
        if (sh.epi == 1) { p.store = DS_STORE_ROW; p.c_split = 1; p.c_plane = (long long)M * N; p.act = DS_ACT_GELU2; }
        if (sh.epi == 2) { p.store = DS_STORE_ATTN; p.attn_kv = img; p.attn_heads = H; p.attn_nkey = 288; p.attn_qplane = (long long)B * H * L * 64; }
        printf("B = %d: %s   %d tiles = %.2f per CU%s\n", B, sh.name, B * N / 256, B * N / 256 / 256.0,
               zero_lo == 2 ? "   [all operands zero]" : zero_lo ? "   [lo planes zero]" : mask_lo ? "   [lo planes: low mantissa bits cleared, PROBE_MASK_LO]" : "");
        int vi = -1;
        for (const Var& v : vars) {
            ++vi;
            if (const char* sel = getenv("PROBE_VARIANTS")) {
                char key[8];
                snprintf(key, sizeof(key), ",%d,", vi);
                std::string hay = std::string(",") + sel + ",";
                if (hay.find(key) == std::string::npos) continue;
            }
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            // settle for half the duration, then measure the second half
            const auto t0 = std::chrono::steady_clock::now();
            long launches = 0, warm = 0;
            bool measuring = false;
            int rc = 0;
            while (true) {
                const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (!measuring && el >= dur * 0.5) { hipDeviceSynchronize(); measuring = true; warm = launches; hipEventRecord(e0, 0); tel.begin(); }
                if (el >= dur) break;
                for (int i = 0; i < 8 && rc == 0; ++i) { rc = v.fn(p, 0); ++launches; }
                if (rc) break;
                if ((launches & 63) == 0) hipStreamSynchronize(0);       // bound the queue depth
            }
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            double watts, mhz;
            tel.end(&watts, &mhz);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const hipError_t err = hipGetLastError();
            if (rc || err != hipSuccess) { printf("    %-34s launch error rc=%d %s\n", v.name, rc, hipGetErrorString(err)); continue; }
            const double us = ms * 1e3 / (double)(launches - warm);
            const double alg = 2.0 * B * 265 * (double)N * K;           // algorithmic: the 265 real rows of a sample
            printf("    %-34s %8.1f us  %6.1f TF-eq (alg.)  MFMA stream %6.0f TF   board %5.0f W  sclk %4.0f MHz   (%ld launches)\n", v.name, us,
                   alg / us / 1e6, 3.0 * 2.0 * M * (double)N * K / us / 1e6, watts, mhz, launches - warm);
            fflush(stdout);
        }
    }
    tel.finish();
    return 0;
}
