// Standalone limiter probe for the f16x2 GEMM main loop (LDS-DMA staging, 128xBN tile, pre-split operands).
// Variants remove one ingredient at a time so the launch time shows what bounds the loop; the shader clock is
// measured in-kernel (s_memtime cycles / s_memrealtime 100 MHz ticks).  Not part of the product library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/probe_gemm_f16x2.hip -o tools/probe/probe_gemm_f16x2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) const void* gptr;
typedef __attribute__((address_space(3))) void* lptr;
#define HLD 32
#define HBK 32
// PROBE bits: 1 = no DMA in the loop, 2 = no MFMA, 4 = no fragment reads in the loop, 8 = no barrier in the loop
template <int BM, int BN, int PROBE, int OCC, int GM>
__global__ __launch_bounds__(256, OCC) void probe_kernel(const _Float16* A, long long a_plane, const _Float16* W,
                                                       long long w_plane, float* C, int M, int N, int K,
                                                       unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int APL = BM * HLD, BPL = BN * HLD, STAGE = 2 * (APL + BPL);
    _Float16* smem = (_Float16*)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (N + BN - 1) / BN, nblk = gridDim.x;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tm_ = bid / tiles_n, tn_ = bid % tiles_n;
    if (GM > 0) {   // grouped raster: GM row tiles x all column tiles per group, column-major inside the group
        const int tiles_m = (M + BM - 1) / BM;
        const int per = GM * tiles_n, grp = bid / per, first = grp * GM;
        const int gsz = tiles_m - first < GM ? tiles_m - first : GM;
        const int in = bid - grp * per;
        tm_ = first + in % gsz;
        tn_ = in / gsz;
    }
    const int m0 = tm_ * BM, n0 = tn_ * BN;
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    constexpr int G = (BM + BN) / 32;
    const _Float16* src[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        int r = 16 * (wave + 4 * i) + (lane >> 2);
        const _Float16* base;
        int org, lim, ld;
        if (r < 2 * BM) { base = A; if (r >= BM) { r -= BM; base += a_plane; } org = m0; lim = M; ld = K; }
        else { r -= 2 * BM; base = W; if (r >= BN) { r -= BN; base += w_plane; } org = n0; lim = N; ld = K; }
        int gr = org + r;
        if (gr >= lim) gr = lim - 1;
        if (PROBE & 16)   // tile-packed planes: [row group of 16][k tile][16 rows x 64 B, pre-swizzled] -> 1 KB linear per instruction
            src[i] = base + ((size_t)(gr >> 4) * (K / HBK)) * 512 + lane * 8;
        else
            src[i] = base + (size_t)gr * ld + ((lane & 3) ^ ((r >> 2) & 3)) * 8;
    }
    const int swz[2] = {((0 + hh) ^ ((l31 >> 2) & 3)) * 8, ((2 + hh) ^ ((l31 >> 2) & 3)) * 8};
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#define DMA(stage_, k0_)                                                                              \
    do {                                                                                              \
        unsigned char* d_ = smem_raw + (stage_) * (STAGE * 2) + wave * 1024;                          \
        _Pragma("unroll") for (int i = 0; i < G; ++i)                                                 \
            __builtin_amdgcn_global_load_lds((gptr)(src[i] + ((PROBE & 16) ? (k0_) * 16 : (k0_))), (lptr)(d_ + i * 4096), 16, 0, 0); \
    } while (0)
#define READS(cur_)                                                                                   \
    do {                                                                                              \
        const _Float16* Ac = smem + (cur_) * STAGE + (wm * TM * 32 + l31) * HLD;                      \
        const _Float16* Bc = smem + (cur_) * STAGE + 2 * APL + (wn * TN * 32 + l31) * HLD;            \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                            \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                          \
                fa0[ks][i] = *(const h8*)(Ac + i * 32 * HLD + swz[ks]);                               \
                fa1[ks][i] = *(const h8*)(Ac + APL + i * 32 * HLD + swz[ks]);                         \
            }                                                                                         \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                          \
                fb0[ks][j] = *(const h8*)(Bc + j * 32 * HLD + swz[ks]);                               \
                fb1[ks][j] = *(const h8*)(Bc + BPL + j * 32 * HLD + swz[ks]);                         \
            }                                                                                         \
        }                                                                                             \
    } while (0)
    const int nk = K / HBK;
    h8 fa0[2][TM], fa1[2][TM], fb0[2][TN], fb1[2][TN];
    DMA(0, 0);
    __syncthreads();
    READS(0);
    for (int kt = 0; kt < nk; ++kt) {
        if (!(PROBE & 8)) __syncthreads();
        if (!(PROBE & 1) && kt + 1 < nk) DMA((kt + 1) & 1, (kt + 1) * HBK);
        if (!(PROBE & 4)) READS(kt & 1);
        if (!(PROBE & 2)) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        f32x16 c = acc[i][j];
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa1[ks][i], fb0[ks][j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0[ks][i], fb1[ks][j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0[ks][i], fb0[ks][j], c, 0, 0, 0);
                        acc[i][j] = c;
                    }
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j][0] += (float)(fa0[ks][i][0] + fa1[ks][i][1]) + (float)(fb0[ks][j][2] + fb1[ks][j][3]);
        }
        if (!(PROBE & 4)) {
            __builtin_amdgcn_sched_group_barrier(0x100, 4 * (TM + TN), 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 6 * TM * TN, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (row < M && col < N) C[(size_t)row * N + col] = acc[i][j][r];
            }
        }
    if (tid == 0 && clk) {
        clk[2 * blockIdx.x] = __builtin_readcyclecounter() - t0;
        clk[2 * blockIdx.x + 1] = wall_clock64() - w0;
    }
}

template <int BM, int BN, int PROBE, int OCC, int GM>
static void run(const char* name, const _Float16* A, const _Float16* W, float* C, int M, int N, int K,
                unsigned long long* clk) {
    const size_t lds = (size_t)2 * 2 * (BM + BN) * HLD * 2;
    hipFuncSetAttribute((const void*)probe_kernel<BM, BN, PROBE, OCC, GM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i)
        hipLaunchKernelGGL((probe_kernel<BM, BN, PROBE, OCC, GM>), dim3(tiles), dim3(256), lds, 0, A, (long long)M * K, W,
                           (long long)N * K, C, M, N, K, clk);
    const int reps = 30;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL((probe_kernel<BM, BN, PROBE, OCC, GM>), dim3(tiles), dim3(256), lds, 0, A, (long long)M * K, W,
                           (long long)N * K, C, M, N, K, clk);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * tiles);
    hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int i = 0; i < tiles; ++i) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
    const double us = ms * 1e3 / reps;
    printf("%-26s GM%-2d %dx%d occ%d M=%d N=%d K=%d: %8.1f us  %6.1f TF-eq  clock %.2f GHz  block life %.0f cyc\n", name, GM, BM, BN, OCC,
           M, N, K, us, 2.0 * M * N * K / us / 1e6, cyc / wall * 0.1, cyc / tiles);
    fflush(stdout);
}

int main() {
    const int M = 64 * 265, Nmax = 4096, Kmax = 4096;
    _Float16 *A, *W; float* C; unsigned long long* clk;
    hipMalloc(&A, (size_t)2 * M * Kmax * 2); hipMalloc(&W, (size_t)2 * Nmax * Kmax * 2);
    hipMalloc(&C, (size_t)M * Nmax * 4); hipMalloc(&clk, 1 << 20);
    // realistic operand statistics (the MFMA power draw, hence the clock, depends on the data): ~N(0,1) values split
    // into hi / lo fp16 planes; weights scaled to max |w| ~ 2^13 like split_f16x2 does
    std::vector<_Float16> h((size_t)2 * M * Kmax), hw((size_t)2 * Nmax * Kmax);
    unsigned long long st = 88172645463325252ull;
    auto gauss = [&]() {
        float a = 0.f;
        for (int q = 0; q < 4; ++q) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; a += (float)(st >> 40) / 16777216.f - 0.5f; }
        return a * 1.7320508f;
    };
    const char* flat = getenv("PROBE_FLAT");
    for (size_t i = 0; i < (size_t)M * Kmax; ++i) {
        const float a = flat ? (float)((i * 2654435761u) >> 20 & 1023) / 1024.f - 0.5f : gauss();
        const _Float16 hi = (_Float16)a;
        h[i] = hi; h[(size_t)M * Kmax + i] = flat ? hi : (_Float16)(a - (float)hi);
    }
    for (size_t i = 0; i < (size_t)Nmax * Kmax; ++i) {
        const float a = (flat ? (float)((i * 2654435761u) >> 20 & 1023) / 1024.f - 0.5f : gauss()) * 4096.f;
        const _Float16 hi = (_Float16)a;
        hw[i] = hi; hw[(size_t)Nmax * Kmax + i] = flat ? hi : (_Float16)(a - (float)hi);
    }
    hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
#define ALL(BM, BN, OCC, N, K, GM)                                                   \
    run<BM, BN, 16, OCC, GM>("packed full", A, W, C, M, N, K, clk);                  \
    run<BM, BN, 16 + 13, OCC, GM>("MFMA only", A, W, C, M, N, K, clk);
#define RAST(BM, BN, OCC, N, K) ALL(BM, BN, OCC, N, K, 8)
    RAST(128, 128, 2, 3072, 1024)
    RAST(128, 128, 2, 1024, 1024)
    RAST(128, 128, 2, 1024, 4096)
    RAST(128, 128, 2, 4096, 1024)
    return 0;
}
