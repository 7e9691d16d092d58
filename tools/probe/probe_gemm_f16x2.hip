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
        if (!(PROBE & 8)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }   // (the product writes this wait out too)
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


// ---- software-pipelined variant: ONE workgroup per CU (4 waves, 128x128), ring of NS stages of 32 KB, LDS-DMA
// issued NS-1 tiles ahead with counted vmcnt, one barrier per k-tile placed between the two k-halves, and the
// fragments of tile k+1 read during the second half of tile k's MFMAs (fragment double buffering in registers).
template <int NS, int GM>
__global__ __launch_bounds__(256, 1) void sp_kernel(const _Float16* A, long long a_plane, const _Float16* W,
                                                    long long w_plane, float* C, int M, int N, int K,
                                                    unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int BM = 128, BN = 128, TM = 2, TN = 2;
    constexpr int APL = BM * HLD, BPL = BN * HLD, STAGE = 2 * (APL + BPL);   // halves
    _Float16* smem = (_Float16*)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (N + BN - 1) / BN, nblk = gridDim.x;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tm_, tn_;
    {
        const int tiles_m = (M + BM - 1) / BM;
        const int per = GM * tiles_n, grp = bid / per, first = grp * GM;
        const int gsz = tiles_m - first < GM ? tiles_m - first : GM;
        const int in = bid - grp * per;
        tm_ = first + in % gsz;
        tn_ = in / gsz;
    }
    const int m0 = tm_ * BM, n0 = tn_ * BN;
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    const int nk = K / HBK;
    constexpr int G = (BM + BN) / 32;   // 8 DMA instructions per wave per k-tile
    unsigned long long src[G];          // wave-uniform tile base address (k-tile 0); lanes add 16 B each
#pragma unroll
    for (int i = 0; i < G; ++i) {
        int r = 16 * (wave + 4 * i);
        const _Float16* base;
        int rg, rgs;
        if (r < 2 * BM) { base = A; if (r >= BM) { r -= BM; base += a_plane; } rg = (m0 + r) >> 4; rgs = (M + 15) >> 4; }
        else { r -= 2 * BM; base = W; if (r >= BN) { r -= BN; base += w_plane; } rg = (n0 + r) >> 4; rgs = (N + 15) >> 4; }
        if (rg >= rgs) rg = rgs - 1;
        const unsigned long long a_ = (unsigned long long)(base + (size_t)rg * nk * 512);
        src[i] = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a_ >> 32)) << 32) |
                 (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a_);
    }
    const unsigned lane16 = lane * 16;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;
#define SP_DMA1(tile_, i)                                                                               \
    do {                                                                                                \
        const unsigned long long sb_ = src[i] + (unsigned long long)(tile_) * 1024;                     \
        const unsigned dd_ = __builtin_amdgcn_readfirstlane(lds0 + ((tile_) % NS) * (STAGE * 2) + wave * 1024 + (i) * 4096); \
        unsigned keep_;                                                                                 \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"                            \
                     "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"                               \
                     : "=&s"(keep_) : "v"(lane16), "s"(sb_), "s"(dd_) : "memory");                      \
    } while (0)
#define SP_DMA(tile_)                                                                                   \
    do {                                                                                                \
        _Pragma("unroll") for (int i = 0; i < G; ++i) SP_DMA1(tile_, i);                                \
    } while (0)
// quarter q of the 16 fragment reads of a tile: q = 0,1 -> k-step 0 (A then B), q = 2,3 -> k-step 1
#define SP_READQ(F, tile_, q)                                                                           \
    do {                                                                                                \
        const _Float16* Ac = smem + ((tile_) % NS) * STAGE + (wm * TM * 32 + l31) * HLD;                \
        const _Float16* Bc = smem + ((tile_) % NS) * STAGE + 2 * APL + (wn * TN * 32 + l31) * HLD;      \
        constexpr int ks = (q) >> 1;                                                                    \
        if (((q) & 1) == 0) {                                                                           \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                            \
                F.a0[ks][i] = *(const h8*)(Ac + i * 32 * HLD + swz[ks]);                                \
                F.a1[ks][i] = *(const h8*)(Ac + APL + i * 32 * HLD + swz[ks]);                          \
            }                                                                                           \
        } else {                                                                                        \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                            \
                F.b0[ks][j] = *(const h8*)(Bc + j * 32 * HLD + swz[ks]);                                \
                F.b1[ks][j] = *(const h8*)(Bc + BPL + j * 32 * HLD + swz[ks]);                          \
            }                                                                                           \
        }                                                                                               \
    } while (0)
#define SP_MFMA1(F, ks, i, j)                                                                           \
    do {                                                                                                \
        f32x16 c = acc[i][j];                                                                           \
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a1[ks][i], F.b0[ks][j], c, 0, 0, 0);               \
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a0[ks][i], F.b1[ks][j], c, 0, 0, 0);               \
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a0[ks][i], F.b0[ks][j], c, 0, 0, 0);               \
        acc[i][j] = c;                                                                                  \
    } while (0)
#define SP_READS(F, tile_)                                                                              \
    do {                                                                                                \
        const _Float16* Ac = smem + ((tile_) % NS) * STAGE + (wm * TM * 32 + l31) * HLD;                \
        const _Float16* Bc = smem + ((tile_) % NS) * STAGE + 2 * APL + (wn * TN * 32 + l31) * HLD;      \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                              \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                            \
                F.a0[ks][i] = *(const h8*)(Ac + i * 32 * HLD + swz[ks]);                                \
                F.a1[ks][i] = *(const h8*)(Ac + APL + i * 32 * HLD + swz[ks]);                          \
            }                                                                                           \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                            \
                F.b0[ks][j] = *(const h8*)(Bc + j * 32 * HLD + swz[ks]);                                \
                F.b1[ks][j] = *(const h8*)(Bc + BPL + j * 32 * HLD + swz[ks]);                          \
            }                                                                                           \
        }                                                                                               \
    } while (0)
#define SP_MFMA(F, ks)                                                                                  \
    do {                                                                                                \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                  \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                            \
                f32x16 c = acc[i][j];                                                                   \
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a1[ks][i], F.b0[ks][j], c, 0, 0, 0);       \
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a0[ks][i], F.b1[ks][j], c, 0, 0, 0);       \
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a0[ks][i], F.b0[ks][j], c, 0, 0, 0);       \
                acc[i][j] = c;                                                                          \
            }                                                                                           \
    } while (0)
    struct Frag { h8 a0[2][TM], a1[2][TM], b0[2][TN], b1[2][TN]; };
    const int swz[2] = {((0 + hh) ^ ((l31 >> 2) & 3)) * 8, ((2 + hh) ^ ((l31 >> 2) & 3)) * 8};
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    Frag FX, FY;
    // prologue: tiles 0 .. NS-2 in flight, tile 0 landed, its fragments in FX
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < nk) SP_DMA(t);
    if (nk >= NS - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G * (NS - 2)));
    else asm volatile("s_waitcnt vmcnt(0)");
    __builtin_amdgcn_s_barrier();
    SP_READS(FX, 0);
    // one k-tile: F holds tile kt, FN receives tile kt+1
#define SP_GROUP(F, FN, q, i, j)                                                                        \
    do {                                                                                                \
        if (issue) { SP_DMA1(kt + NS - 1, 2 * (q)); SP_DMA1(kt + NS - 1, 2 * (q) + 1); }                \
        if (more) SP_READQ(FN, kt + 1, q);                                                              \
        SP_MFMA1(F, 1, i, j);                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                              \
    } while (0)
#define SP_BODY(F, FN)                                                                                  \
    do {                                                                                                \
        const bool issue = kt + NS - 1 < nk, more = kt + 1 < nk;                                        \
        SP_MFMA(F, 0);                                                                                  \
        if (kt + NS - 1 <= nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(G * (NS - 3)));      \
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");                                             \
        __builtin_amdgcn_s_barrier();                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        SP_GROUP(F, FN, 0, 0, 0);                                                                       \
        SP_GROUP(F, FN, 1, 0, 1);                                                                       \
        SP_GROUP(F, FN, 2, 1, 0);                                                                       \
        SP_GROUP(F, FN, 3, 1, 1);                                                                       \
        ++kt;                                                                                           \
    } while (0)
    int kt = 0;
    while (kt + 1 < nk) {
        SP_BODY(FX, FY);
        SP_BODY(FY, FX);
    }
    if (kt < nk) SP_BODY(FX, FY);
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


// ---- big-tile variant (round-2 candidate): ONE workgroup per CU of WGM x WGN waves, BM x BN tile, ring of NS stages,
// LDS-DMA issued NS-1 tiles ahead with a counted vmcnt.  Motivation (profiles/README.md, limiter analysis): at 128x128
// the loop moves 341 B L2->LDS and 683 B LDS->VGPR per MFMA; MFMA-only and DMA-only loops each take ~2/3 of the full
// loop's time and overlap poorly.  256x256 (8 waves of 128x64) halves the DMA bytes and cuts the fragment reads by a
// quarter per MFMA, and a k-tile's MFMAs (1536 cycles per wave) outlast the DMA latency; 256x128 with three stages keeps
// two tiles in flight.  Same packed operands, same MFMA order per accumulator -> results are bit-identical to probe_kernel.
template <int BM, int BN, int WGM, int WGN, int NS, int GM, int SCHED = 0>
__global__ __launch_bounds__(WGM * WGN * 64, 1) void big_kernel(const _Float16* A, long long a_plane, const _Float16* W,
                                                                 long long w_plane, float* C, int M, int N, int K,
                                                                 unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NW = WGM * WGN;
    constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
    constexpr int APL = BM * HLD, BPL = BN * HLD, STAGE = 2 * (APL + BPL);   // halves
    constexpr int G = 2 * (BM + BN) / 16 / NW;                                // DMA instructions per wave per k-tile
    static_assert(G * NW * 16 == 2 * (BM + BN), "16-row groups must divide over the waves");
    static_assert(BM % (WGM * 32) == 0 && BN % (WGN * 32) == 0, "wave tiles of 32x32 blocks");
    _Float16* smem = (_Float16*)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int wm = wave / WGN, wn = wave % WGN;
    const int tiles_n = (N + BN - 1) / BN, nblk = gridDim.x;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tm_, tn_;
    {
        const int tiles_m = (M + BM - 1) / BM;
        const int per = GM * tiles_n, grp = bid / per, first = grp * GM;
        const int gsz = tiles_m - first < GM ? tiles_m - first : GM;
        const int in = bid - grp * per;
        tm_ = first + in % gsz;
        tn_ = in / gsz;
    }
    const int m0 = tm_ * BM, n0 = tn_ * BN;
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    const int nk = K / HBK;
    const _Float16* src[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        int r = 16 * (wave + NW * i);
        const _Float16* base;
        int rg, rgs;
        if (r < 2 * BM) { base = A; if (r >= BM) { r -= BM; base += a_plane; } rg = (m0 + r) >> 4; rgs = (M + 15) >> 4; }
        else { r -= 2 * BM; base = W; if (r >= BN) { r -= BN; base += w_plane; } rg = (n0 + r) >> 4; rgs = (N + 15) >> 4; }
        if (rg >= rgs) rg = rgs - 1;
        src[i] = base + (size_t)rg * nk * 512 + lane * 8;
    }
#define BG_DMA(stage_, tile_)                                                                         \
    do {                                                                                              \
        unsigned char* d_ = smem_raw + (stage_) * (STAGE * 2) + wave * 1024;                          \
        _Pragma("unroll") for (int i = 0; i < G; ++i)                                                 \
            __builtin_amdgcn_global_load_lds((gptr)(src[i] + (size_t)(tile_) * 512), (lptr)(d_ + i * NW * 1024), 16, 0, 0); \
    } while (0)
    const int swz[2] = {((0 + hh) ^ ((l31 >> 2) & 3)) * 8, ((2 + hh) ^ ((l31 >> 2) & 3)) * 8};
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // prologue: tiles 0 .. NS-2 in flight
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < nk) BG_DMA(t, t);
    int cur = 0, nxt = NS - 1;       // stage of tile kt, stage that receives tile kt + NS - 1
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt landed: at most the G (NS-2) younger instructions of this wave may still be in flight
        if (kt + NS - 1 <= nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G * (NS - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();             // ... for every wave, and stage `nxt` (tile kt-1) is free
        if (kt + NS - 1 < nk) BG_DMA(nxt, kt + NS - 1);
        const _Float16* Ac = smem + cur * STAGE + (wm * TM * 32 + l31) * HLD;
        const _Float16* Bc = smem + cur * STAGE + 2 * APL + (wn * TN * 32 + l31) * HLD;
        h8 fa0[2][TM], fa1[2][TM], fb0[2][TN], fb1[2][TN];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                fa0[ks][i] = *(const h8*)(Ac + i * 32 * HLD + swz[ks]);
                fa1[ks][i] = *(const h8*)(Ac + APL + i * 32 * HLD + swz[ks]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                fb0[ks][j] = *(const h8*)(Bc + j * 32 * HLD + swz[ks]);
                fb1[ks][j] = *(const h8*)(Bc + BPL + j * 32 * HLD + swz[ks]);
            }
        }
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
        if (SCHED == 0) {            // k-step 0 fragments, its MFMAs, k-step 1 fragments (same registers), its MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TM + TN), 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * TM * TN, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TM + TN), 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * TM * TN, 0);
        } else if (SCHED == 1) {     // all fragments first (two register sets), like the product kernel
            __builtin_amdgcn_sched_group_barrier(0x100, 4 * (TM + TN), 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 6 * TM * TN, 0);
        } else {                     // k-step 1 fragments issued one third into k-step 0's MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TM + TN), 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TM + TN), 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 5 * TM * TN, 0);
        }
        cur = cur + 1 == NS ? 0 : cur + 1;
        nxt = nxt + 1 == NS ? 0 : nxt + 1;
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

template <int BM, int BN, int WGM, int WGN, int NS, int GM, int SCHED = 0>
static void run_big(const char* name, const _Float16* A, const _Float16* W, float* C, int M, int N, int K,
                    unsigned long long* clk) {
    const size_t lds = (size_t)NS * 2 * (BM + BN) * HLD * 2;
    hipFuncSetAttribute((const void*)big_kernel<BM, BN, WGM, WGN, NS, GM, SCHED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN), nt = WGM * WGN * 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i)
        hipLaunchKernelGGL((big_kernel<BM, BN, WGM, WGN, NS, GM, SCHED>), dim3(tiles), dim3(nt), lds, 0, A, (long long)M * K, W,
                           (long long)N * K, C, M, N, K, clk);
    const int reps = 30;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL((big_kernel<BM, BN, WGM, WGN, NS, GM, SCHED>), dim3(tiles), dim3(nt), lds, 0, A, (long long)M * K, W,
                           (long long)N * K, C, M, N, K, clk);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) printf("  launch error: %s\n", hipGetErrorString(err));
    std::vector<unsigned long long> h(2 * tiles);
    hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int i = 0; i < tiles; ++i) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
    const double us = ms * 1e3 / reps;
    printf("%-26s NS%d GM%-2d %dx%d %dw  M=%d N=%d K=%d: %8.1f us  %6.1f TF-eq  clock %.2f GHz  block life %.0f cyc  (%d tiles = %.2f rounds)\n",
           name, NS, GM, BM, BN, WGM * WGN, M, N, K, us, 2.0 * M * N * K / us / 1e6, cyc / wall * 0.1, cyc / tiles, tiles,
           tiles / 256.0);
    fflush(stdout);
}


// ---- register-staged variant on PACKED planes (round-2 candidate): the 128x128 4-wave program at two workgroups per
// CU, but the tile bytes travel global -> VGPR -> ds_write_b128 instead of LDS-DMA, with two register sets so that the
// loads of tile kt+2 are issued before tile kt is computed: twice the latency tolerance of the two-stage DMA ring at
// the same LDS footprint (the third stage lives in 64 VGPRs).  Same lane -> byte mapping as the DMA (1 KB per wave
// instruction, linear LDS writes).  Never measured with packed planes: the register-staged kernel of the early round
// read row-major planes (half-cacheline requests).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int BM, int BN, int OCC, int GM>
__global__ __launch_bounds__(256, OCC) void reg_kernel(const _Float16* A, long long a_plane, const _Float16* W,
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
    int tm_, tn_;
    {
        const int tiles_m = (M + BM - 1) / BM;
        const int per = GM * tiles_n, grp = bid / per, first = grp * GM;
        const int gsz = tiles_m - first < GM ? tiles_m - first : GM;
        const int in = bid - grp * per;
        tm_ = first + in % gsz;
        tn_ = in / gsz;
    }
    const int m0 = tm_ * BM, n0 = tn_ * BN;
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    const int nk = K / HBK;
    constexpr int G = (BM + BN) / 32;
    const _Float16* src[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        int r = 16 * (wave + 4 * i);
        const _Float16* base;
        int rg, rgs;
        if (r < 2 * BM) { base = A; if (r >= BM) { r -= BM; base += a_plane; } rg = (m0 + r) >> 4; rgs = (M + 15) >> 4; }
        else { r -= 2 * BM; base = W; if (r >= BN) { r -= BN; base += w_plane; } rg = (n0 + r) >> 4; rgs = (N + 15) >> 4; }
        if (rg >= rgs) rg = rgs - 1;
        src[i] = base + (size_t)rg * nk * 512 + lane * 8;
    }
    const int swz[2] = {((0 + hh) ^ ((l31 >> 2) & 3)) * 8, ((2 + hh) ^ ((l31 >> 2) & 3)) * 8};
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    u32x4 rX[G], rY[G];
#define RG_LOAD(R, tile_)                                                                             \
    do {                                                                                              \
        _Pragma("unroll") for (int i = 0; i < G; ++i) R[i] = *(const u32x4*)(src[i] + (size_t)(tile_) * 512); \
    } while (0)
#define RG_WRITE(R, stage_)                                                                           \
    do {                                                                                              \
        unsigned char* d_ = smem_raw + (stage_) * (STAGE * 2) + wave * 1024 + lane * 16;              \
        _Pragma("unroll") for (int i = 0; i < G; ++i) *(u32x4*)(d_ + i * 4096) = R[i];                \
    } while (0)
#define RG_COMPUTE(cur_)                                                                              \
    do {                                                                                              \
        const _Float16* Ac = smem + (cur_) * STAGE + (wm * TM * 32 + l31) * HLD;                      \
        const _Float16* Bc = smem + (cur_) * STAGE + 2 * APL + (wn * TN * 32 + l31) * HLD;            \
        h8 fa0[2][TM], fa1[2][TM], fb0[2][TN], fb1[2][TN];                                            \
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
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                              \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                            \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                      \
                    f32x16 c = acc[i][j];                                                             \
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa1[ks][i], fb0[ks][j], c, 0, 0, 0);   \
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0[ks][i], fb1[ks][j], c, 0, 0, 0);   \
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0[ks][i], fb0[ks][j], c, 0, 0, 0);   \
                    acc[i][j] = c;                                                                    \
                }                                                                                     \
    } while (0)
    // X holds tile kt+1 (landed or landing), Y receives tile kt+2; tile kt is in stage cur
#define RG_BODY(RX, RY)                                                                               \
    do {                                                                                              \
        RG_LOAD(RY, kt + 2 < nk ? kt + 2 : nk - 1);                                                   \
        RG_COMPUTE(cur);                                                                              \
        RG_WRITE(RX, cur ^ 1);                                                                        \
        __builtin_amdgcn_sched_group_barrier(0x020, G, 0);           /* the VMEM loads first */        \
        __builtin_amdgcn_sched_group_barrier(0x100, 4 * (TM + TN), 0); /* fragment reads */           \
        __builtin_amdgcn_sched_group_barrier(0x008, 6 * TM * TN, 0);   /* MFMAs */                    \
        __builtin_amdgcn_sched_group_barrier(0x200, G, 0);           /* then the staging writes */    \
        __syncthreads();                                                                              \
        cur ^= 1;                                                                                     \
        ++kt;                                                                                         \
    } while (0)
    RG_LOAD(rX, 0);
    RG_WRITE(rX, 0);
    RG_LOAD(rX, nk > 1 ? 1 : 0);
    __syncthreads();
    int cur = 0, kt = 0;
    while (kt + 1 < nk) {
        RG_BODY(rX, rY);
        RG_BODY(rY, rX);
    }
    if (kt < nk) RG_BODY(rX, rY);
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

template <int BM, int BN, int OCC, int GM>
static void run_reg(const char* name, const _Float16* A, const _Float16* W, float* C, int M, int N, int K,
                    unsigned long long* clk) {
    const size_t lds = (size_t)2 * 2 * (BM + BN) * HLD * 2;
    hipFuncSetAttribute((const void*)reg_kernel<BM, BN, OCC, GM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i)
        hipLaunchKernelGGL((reg_kernel<BM, BN, OCC, GM>), dim3(tiles), dim3(256), lds, 0, A, (long long)M * K, W,
                           (long long)N * K, C, M, N, K, clk);
    const int reps = 30;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL((reg_kernel<BM, BN, OCC, GM>), dim3(tiles), dim3(256), lds, 0, A, (long long)M * K, W,
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


// ---- ping-pong variant (round-2 candidate; the structure cdna_hip_programming.md calls "256^2 8-phase"): one 8-wave
// workgroup per CU, 256x256 tile, waves 2 (M) x 4 (N) of 128x64, two 64 KB LDS buffers (k-tiles of 32 x two planes).
// A k-tile is consumed in FOUR phases, one 64x32 quadrant pair of the wave tile each (12 MFMAs):
//     phase 0: read A-sub0 (8 x ds_read_b128) + B-sub0 (4)  -> quadrant (A0, B0)
//     phase 1: read B-sub1 (4)                              -> (A0, B1)
//     phase 2: read A-sub1 (8)                              -> (A1, B1)
//     phase 3: (B-sub0 is still in registers)               -> (A1, B0)
// and staged in four QUARTERS of 16 KB (A-sub0 rows of both wave rows, B-sub0, B-sub1, A-sub1), one quarter issued per
// phase (2 DMA instructions per wave), LEAD quarters ahead of the phase that computes: the wait of a phase is a
// COUNTED vmcnt(2 (LEAD - 2)) -- never 0 in the steady state -- so LEAD-2 quarters stay in flight across the barriers.
// Each phase = { ds_read sub-tile; issue a quarter; vmcnt(n); barrier; MFMAs under setprio(1); barrier }, and the two
// wave rows run one barrier apart (the second row passes one extra barrier up front), so on every SIMD one wave
// issues MFMAs while the other reads LDS and issues DMA.
// Hazards (phase numbers g = 4 tile + p; quarter q = 4 tile + type is needed at phase 4 tile + {0, 0, 1, 2}[type]):
//   RAW: the wait of phase g retires the quarters <= g + 2 of this wave, the barrier behind it makes that true of every
//        wave of its row, the other row is at most one barrier away -> quarter q is read in phase >= (its wait) + 1.
//   WAR: quarter g + LEAD lands on the region whose previous occupant was last read >= 2 phases earlier (LEAD <= 6).
// Same MFMA order per accumulator as probe_kernel -> bit-identical results.
// FLAGS bit 0: the two wave rows run one barrier apart (ping-pong); bit 1: s_setprio(1) around the MFMA clusters;
// bit 2: quarter order {B-sub0, A-sub0, B-sub1, A-sub1} with the NEXT tile's B-sub0 read in phase 3 into a second
//        register set -> 8 / 4 / 8 / 4 ds_reads per phase instead of 12 / 4 / 8 / 0, every region is re-staged >= 3
//        phases after its last read at LEAD 6, so LEAD 7 (five quarters = 80 KB in flight) becomes legal
template <int LEAD, int GM, int FLAGS = 3>
__global__ __launch_bounds__(512, 1) void pp_kernel(const _Float16* A, long long a_plane, const _Float16* W,
                                                    long long w_plane, float* C, int M, int N, int K,
                                                    unsigned long long* clk) {
    constexpr bool ORD = (FLAGS & 4) != 0;
    static_assert(LEAD == 5 || LEAD == 6 || (ORD && LEAD == 7), "quarters in flight: see the WAR analysis");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int BM = 256, BN = 256;
    constexpr int APL = BM * HLD, BPL = BN * HLD, STAGE = 2 * (APL + BPL);   // halves; STAGE * 2 = 64 KB
    _Float16* smem = (_Float16*)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int wr = wave >> 2, wc = wave & 3;
    const int tiles_n = (N + BN - 1) / BN, nblk = gridDim.x;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tm_, tn_;
    {
        const int tiles_m = (M + BM - 1) / BM;
        const int per = GM * tiles_n, grp = bid / per, first = grp * GM;
        const int gsz = tiles_m - first < GM ? tiles_m - first : GM;
        const int in = bid - grp * per;
        tm_ = first + in % gsz;
        tn_ = in / gsz;
    }
    const int m0 = tm_ * BM, n0 = tn_ * BN;
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    const int nk = K / HBK;                       // even (K % 64 == 0)
    // this wave's two 16-row groups of each quarter type: 0 = A-sub0, 1 = B-sub0, 2 = B-sub1, 3 = A-sub1
    // (ORD: 0 = B-sub0, 1 = A-sub0, 2 = B-sub1, 3 = A-sub1)
    // wave-uniform 64-bit bases (SGPRs) + one per-lane byte offset: keeps the eight tile pointers out of the VGPR file
    unsigned long long src[4][2];
    int ldsoff[4][2];
    const unsigned lane16 = lane * 16;
#pragma unroll
    for (int ty = 0; ty < 4; ++ty)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = 2 * wave + k, plane = idx >> 3, r = idx & 7;
            const bool isA = ORD ? (ty & 1) : (ty == 0 || ty == 3);
            const int s = isA ? (ty == 3) : (ty == 2);
            const int gip = isA ? (r >> 2) * 8 + s * 4 + (r & 3)       // 16-row group inside the 256-row plane
                                : (r >> 1) * 4 + s * 2 + (r & 1);
            int rg = ((isA ? m0 : n0) >> 4) + gip;
            const int rgs = ((isA ? M : N) + 15) >> 4;
            if (rg >= rgs) rg = rgs - 1;
            const unsigned long long a_ = (unsigned long long)((isA ? A + plane * a_plane : W + plane * w_plane) + (size_t)rg * nk * 512);
            src[ty][k] = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a_ >> 32)) << 32) |
                         (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a_);   // (unsigned): the builtin returns int -- without it the low word is SIGN-extended into the high one
            ldsoff[ty][k] = __builtin_amdgcn_readfirstlane(((isA ? 0 : 32) + plane * 16 + gip) * 1024);
        }
#define PP_ISSUE(tile_, ty_, buf_)                                                                   \
    do {                                                                                             \
        _Pragma("unroll") for (int k = 0; k < 2; ++k)                                                \
            __builtin_amdgcn_global_load_lds((gptr)((const unsigned char*)(src[ty_][k] + (unsigned long long)(tile_) * 1024) + lane16), \
                                             (lptr)(smem_raw + (buf_) * (STAGE * 2) + ldsoff[ty_][k]), 16, 0, 0); \
    } while (0)
#define PP_FENCE()                                                                                   \
    do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PP_BAR()                                                                                     \
    do { PP_FENCE(); __builtin_amdgcn_s_barrier(); PP_FENCE(); } while (0)
    const int swz[2] = {((0 + hh) ^ ((l31 >> 2) & 3)) * 8, ((2 + hh) ^ ((l31 >> 2) & 3)) * 8};
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    h8 a0[2][2], a1[2][2];          // [ks][row block of the current A-sub]: hi, lo planes
    h8 b0[3][2], b1[3][2];          // [slot][ks]: hi, lo planes; slot 0 = B-sub0 (ORD: of even k-tiles), 1 = B-sub1,
                                    // 2 = B-sub0 of odd k-tiles (ORD only)
#define PP_READ_A(buf_, s_)                                                                          \
    do {                                                                                             \
        const _Float16* Ac = smem + (buf_) * STAGE + (wr * 128 + (s_) * 64 + l31) * HLD;             \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                             \
            _Pragma("unroll") for (int ib = 0; ib < 2; ++ib) {                                       \
                a0[ks][ib] = *(const h8*)(Ac + ib * 32 * HLD + swz[ks]);                             \
                a1[ks][ib] = *(const h8*)(Ac + APL + ib * 32 * HLD + swz[ks]);                       \
            }                                                                                        \
    } while (0)
#define PP_READ_B(buf_, s_, slot_)                                                                   \
    do {                                                                                             \
        const _Float16* Bc = smem + (buf_) * STAGE + 2 * APL + (wc * 64 + (s_) * 32 + l31) * HLD;    \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                           \
            b0[slot_][ks] = *(const h8*)(Bc + swz[ks]);                                              \
            b1[slot_][ks] = *(const h8*)(Bc + BPL + swz[ks]);                                        \
        }                                                                                            \
    } while (0)
    // quadrant (A-sub sa, B-sub sb): per accumulator the order is ks 0 {a1 b0, a0 b1, a0 b0}, ks 1 {...}
#define PP_QUAD(sa_, sb_, sl_)                                                                       \
    do {                                                                                             \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                           \
            _Pragma("unroll") for (int ib = 0; ib < 2; ++ib)                                         \
                acc[2 * (sa_) + ib][sb_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[ks][ib], b0[sl_][ks], acc[2 * (sa_) + ib][sb_], 0, 0, 0); \
            _Pragma("unroll") for (int ib = 0; ib < 2; ++ib)                                         \
                acc[2 * (sa_) + ib][sb_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[ks][ib], b1[sl_][ks], acc[2 * (sa_) + ib][sb_], 0, 0, 0); \
            _Pragma("unroll") for (int ib = 0; ib < 2; ++ib)                                         \
                acc[2 * (sa_) + ib][sb_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[ks][ib], b0[sl_][ks], acc[2 * (sa_) + ib][sb_], 0, 0, 0); \
        }                                                                                            \
    } while (0)
    // one phase: P = phase within the k-tile (compile time), BUF = parity of the k-tile t (compile time)
#define PP_PHASE(P, BUF)                                                                             \
    do {                                                                                             \
        if (P == 0) { PP_READ_A(BUF, 0); if (!ORD) PP_READ_B(BUF, 0, 0); }                           \
        if (P == 1) PP_READ_B(BUF, 1, 1);                                                            \
        if (P == 2) PP_READ_A(BUF, 1);                                                               \
        if (P == 3 && ORD && t + 1 < nk) PP_READ_B((BUF) ^ 1, 0, (BUF) ? 0 : 2);   /* next tile's B-sub0 */ \
        PP_FENCE();                                                                                  \
        {                                                                                            \
            constexpr int dq = (P) + LEAD;                     /* quarter 4 t + dq */                \
            const int tq = t + (dq >> 2);                                                            \
            if (tq < nk) {                                                                           \
                PP_ISSUE(tq, dq & 3, ((BUF) + (dq >> 2)) & 1);                                       \
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (LEAD - 2)) : "memory");                \
            } else {                                                                                 \
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   /* tail: nothing younger to count */ \
            }                                                                                        \
        }                                                                                            \
        PP_BAR();                                                                                    \
        if (FLAGS & 2) __builtin_amdgcn_s_setprio(1);                                                \
        if (P == 0) PP_QUAD(0, 0, (ORD && (BUF)) ? 2 : 0);                                           \
        if (P == 1) PP_QUAD(0, 1, 1);                                                                \
        if (P == 2) PP_QUAD(1, 1, 1);                                                                \
        if (P == 3) PP_QUAD(1, 0, (ORD && (BUF)) ? 2 : 0);                                           \
        if (FLAGS & 2) __builtin_amdgcn_s_setprio(0);                                                \
        PP_BAR();                                                                                    \
    } while (0)
    // prologue: quarters 0 .. LEAD-1 (k-tile 0 and the first LEAD-4 quarters of k-tile 1), quarters 0 and 1 landed
#pragma unroll
    for (int q = 0; q < LEAD; ++q)
        if ((q >> 2) < nk) PP_ISSUE(q >> 2, q & 3, (q >> 2) & 1);
    if (4 * nk >= LEAD) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (LEAD - 2)) : "memory");   // quarters 0, 1 landed
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // fewer than LEAD quarters exist
    PP_BAR();
    if ((FLAGS & 1) && wr == 1) PP_BAR();   // the second wave row runs one barrier behind the first
    if (ORD) { PP_READ_B(0, 0, 0); PP_FENCE(); }   // k-tile 0's B-sub0 (quarter 0, retired by the prologue wait)
    for (int t = 0; t < nk; t += 2) {
        PP_PHASE(0, 0); PP_PHASE(1, 0); PP_PHASE(2, 0); PP_PHASE(3, 0);
        ++t;
        PP_PHASE(0, 1); PP_PHASE(1, 1); PP_PHASE(2, 1); PP_PHASE(3, 1);
        --t;
    }
    if ((FLAGS & 1) && wr == 0) PP_BAR();   // ... and the first row waits for it at the end
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + (wc * 2 + j) * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wr * 4 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (row < M && col < N) C[(size_t)row * N + col] = acc[i][j][r];
            }
        }
    if (tid == 0 && clk) {
        clk[2 * blockIdx.x] = __builtin_readcyclecounter() - t0;
        clk[2 * blockIdx.x + 1] = wall_clock64() - w0;
    }
}

template <int LEAD, int GM, int FLAGS = 3>
static void run_pp(const char* name, const _Float16* A, const _Float16* W, float* C, int M, int N, int K,
                   unsigned long long* clk) {
    const size_t lds = (size_t)2 * 2 * (256 + 256) * HLD * 2;
    hipFuncSetAttribute((const void*)pp_kernel<LEAD, GM, FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i)
        hipLaunchKernelGGL((pp_kernel<LEAD, GM, FLAGS>), dim3(tiles), dim3(512), lds, 0, A, (long long)M * K, W, (long long)N * K, C,
                           M, N, K, clk);
    const int reps = 30;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL((pp_kernel<LEAD, GM, FLAGS>), dim3(tiles), dim3(512), lds, 0, A, (long long)M * K, W, (long long)N * K, C,
                           M, N, K, clk);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) printf("  launch error: %s\n", hipGetErrorString(err));
    std::vector<unsigned long long> h(2 * tiles);
    hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int i = 0; i < tiles; ++i) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
    const double us = ms * 1e3 / reps;
    printf("%-26s LEAD%d GM%-2d 256x256 8w  M=%d N=%d K=%d: %8.1f us  %6.1f TF-eq  clock %.2f GHz  block life %.0f cyc  (%d tiles = %.2f rounds)\n",
           name, LEAD, GM, M, N, K, us, 2.0 * M * N * K / us / 1e6, cyc / wall * 0.1, cyc / tiles, tiles, tiles / 256.0);
    fflush(stdout);
}

// ---- per-sample ping-pong variant ("ps"): the pp_kernel main loop on a 288 x 256 tile that covers exactly ONE sample
// (RPS = 265 rows) of the denoiser's activation matrix.  M = B * 265 rows never divide into 256-row tiles (B = 64:
// 66.25 row tiles -> 1.05 / 3.14 / 4.19 rounds of 256 CUs), but B * N / 256 per-sample tiles are whole rounds at B = 64
// for every N of the network, with no tail program.  Tile rows = [m0, m0 + 288), m0 = 16 floor(265 b / 16) (packed
// planes come in 16-row groups), which contains the sample's rows [265 b, 265 b + 265); only those are stored.
// Blocks 0..7 (rows 0..255 of the tile) are pp_kernel's: wave (wr, wc) owns rows wr 128 + [0, 128), columns wc 64 +
// [0, 64).  The ninth block row (rows 256..287, 32 x 256) is split by COLUMNS over all eight waves: wave (wr, wc) adds
// the 32 x 32 block at columns (2 wc + wr) 32 -- B-sub `wr` of its own column range, whose fragments it already holds --
// so every wave runs 54 instead of 48 MFMAs per k-tile (265 / 288 = 92 % useful), 16 more accumulator registers, 4 more
// fragment reads.  The block-8 rows travel with the A-sub1 quarter (3 instead of 2 DMA instructions per wave; waves
// 4..7 repeat the loads of waves 0..3 -- same bytes to the same LDS address -- so that every wave counts the same):
// four consecutive quarters are always one of each type = 9 instructions, so the steady-state wait is vmcnt(9).
// Block 8 is read and multiplied in phase 3 (which reads nothing in pp_kernel): its quarter (type 3 of tile t) is retired
// by the wait of phase 4t + 1, and the region is re-staged by quarter 4t + 11, issued in phase 4t + 5 -- two phases
// after the last read (pp_kernel's WAR rule).
// EX: MFMA cost model of the ninth block row (timing only -- EX < 2 computes wrong values for rows 256..): 2 = the 32-row
// block of the product kernel (6 MFMAs 32x32x16 per k-tile and wave), 1 = half of it (= what a 16-row block on
// v_mfma_f32_16x16x32_f16 would issue if samples were aligned to 16-row groups: 272-row tiles), 0 = none (256 rows)
template <int GM, int EX = 2>
__global__ __launch_bounds__(512, 1) void ps_kernel(const _Float16* A, long long a_plane, const _Float16* W,
                                                    long long w_plane, float* C, int M, int N, int K,
                                                    unsigned long long* clk) {
    constexpr int LEAD = 6, RPS = 265;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int BM = 288, BN = 256;
    constexpr int APL = BM * HLD, BPL = BN * HLD, STAGE = 2 * (APL + BPL);   // halves; STAGE * 2 = 68 KB
    _Float16* smem = (_Float16*)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int wr = wave >> 2, wc = wave & 3;
    const int tiles_n = (N + BN - 1) / BN, nblk = gridDim.x;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tm_, tn_;
    {
        const int tiles_m = M / RPS;
        const int per = GM * tiles_n, grp = bid / per, first = grp * GM;
        const int gsz = tiles_m - first < GM ? tiles_m - first : GM;
        const int in = bid - grp * per;
        tm_ = first + in % gsz;
        tn_ = in / gsz;
    }
    const int row_lo = tm_ * RPS, row_hi = row_lo + RPS;      // the rows this tile stores
    const int m0 = (row_lo >> 4) << 4, n0 = tn_ * BN;
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    const int nk = K / HBK;                       // even (K % 64 == 0)
    unsigned long long src[4][2], src8;           // quarter types: 0 = A-sub0, 1 = B-sub0, 2 = B-sub1, 3 = A-sub1 (+ block 8)
    int ldsoff[4][2], ldsoff8;
    const unsigned lane16 = lane * 16;
    const int rgsA = (M + 15) >> 4, rgsB = (N + 15) >> 4;
#define PS_BASE(dst_, ptr_)                                                                          \
    do {                                                                                             \
        const unsigned long long a_ = (unsigned long long)(ptr_);                                    \
        dst_ = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a_ >> 32)) << 32) | \
               (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a_);           \
    } while (0)
#pragma unroll
    for (int ty = 0; ty < 4; ++ty)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = 2 * wave + k, plane = idx >> 3, r = idx & 7;
            const bool isA = (ty == 0 || ty == 3);
            const int s = isA ? (ty == 3) : (ty == 2);
            const int gip = isA ? (r >> 2) * 8 + s * 4 + (r & 3)       // 16-row group inside rows 0..255 of the A plane
                                : (r >> 1) * 4 + s * 2 + (r & 1);
            int rg = ((isA ? m0 : n0) >> 4) + gip;
            const int rgs = isA ? rgsA : rgsB;
            if (rg >= rgs) rg = rgs - 1;
            PS_BASE(src[ty][k], (isA ? A + plane * a_plane : W + plane * w_plane) + (size_t)rg * nk * 512);
            ldsoff[ty][k] = __builtin_amdgcn_readfirstlane((isA ? plane * 18 + gip : 36 + plane * 16 + gip) * 1024);
        }
    {   // block 8: groups 16, 17 of both planes = 4 KB; wave w (and w + 4) loads piece e = w & 3
        const int e = wave & 3, plane = e >> 1, gip = 16 + (e & 1);
        int rg = (m0 >> 4) + gip;
        if (rg >= rgsA) rg = rgsA - 1;
        PS_BASE(src8, A + plane * a_plane + (size_t)rg * nk * 512);
        ldsoff8 = __builtin_amdgcn_readfirstlane((plane * 18 + gip) * 1024);
    }
#define PS_ISSUE(tile_, ty_, buf_)                                                                   \
    do {                                                                                             \
        _Pragma("unroll") for (int k = 0; k < 2; ++k)                                                \
            __builtin_amdgcn_global_load_lds((gptr)((const unsigned char*)(src[ty_][k] + (unsigned long long)(tile_) * 1024) + lane16), \
                                             (lptr)(smem_raw + (buf_) * (STAGE * 2) + ldsoff[ty_][k]), 16, 0, 0); \
        if ((ty_) == 3)                                                                              \
            __builtin_amdgcn_global_load_lds((gptr)((const unsigned char*)(src8 + (unsigned long long)(tile_) * 1024) + lane16), \
                                             (lptr)(smem_raw + (buf_) * (STAGE * 2) + ldsoff8), 16, 0, 0); \
    } while (0)
    const int swz[2] = {((0 + hh) ^ ((l31 >> 2) & 3)) * 8, ((2 + hh) ^ ((l31 >> 2) & 3)) * 8};
    f32x16 acc[4][2], acc8;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc8[r] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    h8 a0[2][2], a1[2][2];          // [ks][row block of the current A-sub]: hi, lo planes
    h8 b0[2][2], b1[2][2];          // [B-sub][ks]: hi, lo planes
    h8 e0[2], e1[2];                // block 8 [ks]: hi, lo planes
#define PS_READ_A(buf_, s_)                                                                          \
    do {                                                                                             \
        const _Float16* Ac = smem + (buf_) * STAGE + (wr * 128 + (s_) * 64 + l31) * HLD;             \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                             \
            _Pragma("unroll") for (int ib = 0; ib < 2; ++ib) {                                       \
                a0[ks][ib] = *(const h8*)(Ac + ib * 32 * HLD + swz[ks]);                             \
                a1[ks][ib] = *(const h8*)(Ac + APL + ib * 32 * HLD + swz[ks]);                       \
            }                                                                                        \
    } while (0)
#define PS_READ_E(buf_)                                                                              \
    do {                                                                                             \
        const _Float16* Ec = smem + (buf_) * STAGE + (256 + l31) * HLD;                              \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                           \
            e0[ks] = *(const h8*)(Ec + swz[ks]);                                                     \
            e1[ks] = *(const h8*)(Ec + APL + swz[ks]);                                               \
        }                                                                                            \
    } while (0)
#define PS_READ_B(buf_, s_)                                                                          \
    do {                                                                                             \
        const _Float16* Bc = smem + (buf_) * STAGE + 2 * APL + (wc * 64 + (s_) * 32 + l31) * HLD;    \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                           \
            b0[s_][ks] = *(const h8*)(Bc + swz[ks]);                                                 \
            b1[s_][ks] = *(const h8*)(Bc + BPL + swz[ks]);                                           \
        }                                                                                            \
    } while (0)
#define PS_QUAD(sa_, sb_)                                                                            \
    do {                                                                                             \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                           \
            _Pragma("unroll") for (int ib = 0; ib < 2; ++ib)                                         \
                acc[2 * (sa_) + ib][sb_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[ks][ib], b0[sb_][ks], acc[2 * (sa_) + ib][sb_], 0, 0, 0); \
            _Pragma("unroll") for (int ib = 0; ib < 2; ++ib)                                         \
                acc[2 * (sa_) + ib][sb_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[ks][ib], b1[sb_][ks], acc[2 * (sa_) + ib][sb_], 0, 0, 0); \
            _Pragma("unroll") for (int ib = 0; ib < 2; ++ib)                                         \
                acc[2 * (sa_) + ib][sb_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[ks][ib], b0[sb_][ks], acc[2 * (sa_) + ib][sb_], 0, 0, 0); \
        }                                                                                            \
    } while (0)
#define PS_EXTRA(sb_)                                                                                \
    do {                                                                                             \
        _Pragma("unroll") for (int ks = 0; ks < EX; ++ks) {                                          \
            acc8 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e1[ks], b0[sb_][ks], acc8, 0, 0, 0);       \
            acc8 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0[ks], b1[sb_][ks], acc8, 0, 0, 0);       \
            acc8 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0[ks], b0[sb_][ks], acc8, 0, 0, 0);       \
        }                                                                                            \
    } while (0)
#define PS_PHASE(P, BUF)                                                                             \
    do {                                                                                             \
        if (P == 0) { PS_READ_A(BUF, 0); PS_READ_B(BUF, 0); }                                        \
        if (P == 1) PS_READ_B(BUF, 1);                                                               \
        if (P == 2) PS_READ_A(BUF, 1);                                                               \
        if (P == 3) PS_READ_E(BUF);                                                                  \
        PP_FENCE();                                                                                  \
        {                                                                                            \
            constexpr int dq = (P) + LEAD;                     /* quarter 4 t + dq */                \
            const int tq = t + (dq >> 2);                                                            \
            if (tq < nk) {                                                                           \
                PS_ISSUE(tq, dq & 3, ((BUF) + (dq >> 2)) & 1);                                       \
                asm volatile("s_waitcnt vmcnt(9)" ::: "memory");  /* the 4 youngest quarters: 2+2+2+3 */ \
            } else {                                                                                 \
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   /* tail: nothing younger to count */ \
            }                                                                                        \
        }                                                                                            \
        PP_BAR();                                                                                    \
        __builtin_amdgcn_s_setprio(1);                                                               \
        if (P == 0) PS_QUAD(0, 0);                                                                   \
        if (P == 1) PS_QUAD(0, 1);                                                                   \
        if (P == 2) PS_QUAD(1, 1);                                                                   \
        if (P == 3) { PS_QUAD(1, 0); if (wr == 0) PS_EXTRA(0); else PS_EXTRA(1); }                   \
        __builtin_amdgcn_s_setprio(0);                                                               \
        PP_BAR();                                                                                    \
    } while (0)
    // prologue: quarters 0 .. 5 (k-tile 0 and types 0, 1 of k-tile 1) = 13 instructions; quarters 0, 1 landed once
    // only the 4 youngest (2 + 3 + 2 + 2 = 9) are outstanding
#pragma unroll
    for (int q = 0; q < LEAD; ++q)
        if ((q >> 2) < nk) PS_ISSUE(q >> 2, q & 3, (q >> 2) & 1);
    if (4 * nk >= LEAD) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PP_BAR();
    if (wr == 1) PP_BAR();   // the second wave row runs one barrier behind the first
    for (int t = 0; t < nk; t += 2) {
        PS_PHASE(0, 0); PS_PHASE(1, 0); PS_PHASE(2, 0); PS_PHASE(3, 0);
        ++t;
        PS_PHASE(0, 1); PS_PHASE(1, 1); PS_PHASE(2, 1); PS_PHASE(3, 1);
        --t;
    }
    if (wr == 0) PP_BAR();   // ... and the first row waits for it at the end
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + (wc * 2 + j) * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wr * 4 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (row >= row_lo && row < row_hi && col < N) C[(size_t)row * N + col] = acc[i][j][r];
            }
        }
    {
        const int col = n0 + (wc * 2 + wr) * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + 256 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (row >= row_lo && row < row_hi && col < N) C[(size_t)row * N + col] = acc8[r];
        }
    }
    if (tid == 0 && clk) {
        clk[2 * blockIdx.x] = __builtin_readcyclecounter() - t0;
        clk[2 * blockIdx.x + 1] = wall_clock64() - w0;
    }
}

template <int GM, int EX = 2>
static void run_ps(const char* name, const _Float16* A, const _Float16* W, float* C, int M, int N, int K,
                   unsigned long long* clk) {
    const size_t lds = (size_t)2 * 2 * (288 + 256) * HLD * 2;
    hipFuncSetAttribute((const void*)ps_kernel<GM, EX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int tiles = (M / 265) * ((N + 255) / 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i)
        hipLaunchKernelGGL((ps_kernel<GM, EX>), dim3(tiles), dim3(512), lds, 0, A, (long long)M * K, W, (long long)N * K, C, M, N, K, clk);
    const int reps = 30;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL((ps_kernel<GM, EX>), dim3(tiles), dim3(512), lds, 0, A, (long long)M * K, W, (long long)N * K, C, M, N, K, clk);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) printf("  launch error: %s\n", hipGetErrorString(err));
    std::vector<unsigned long long> h(2 * tiles);
    hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int i = 0; i < tiles; ++i) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
    const double us = ms * 1e3 / reps;
    printf("%-26s GM%-2d 288x256 8w per-sample M=%d N=%d K=%d: %8.1f us  %6.1f TF-eq  clock %.2f GHz  block life %.0f cyc  (%d tiles = %.2f rounds)\n",
           name, GM, M, N, K, us, 2.0 * M * N * K / us / 1e6, cyc / wall * 0.1, cyc / tiles, tiles, tiles / 256.0);
    fflush(stdout);
}

template <int NS, int GM>
static void run_sp(const char* name, const _Float16* A, const _Float16* W, float* C, int M, int N, int K,
                   unsigned long long* clk) {
    const size_t lds = (size_t)NS * 2 * 256 * HLD * 2;
    hipFuncSetAttribute((const void*)sp_kernel<NS, GM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i)
        hipLaunchKernelGGL((sp_kernel<NS, GM>), dim3(tiles), dim3(256), lds, 0, A, (long long)M * K, W, (long long)N * K, C,
                           M, N, K, clk);
    const int reps = 30;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL((sp_kernel<NS, GM>), dim3(tiles), dim3(256), lds, 0, A, (long long)M * K, W, (long long)N * K, C,
                           M, N, K, clk);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * tiles);
    hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int i = 0; i < tiles; ++i) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
    const double us = ms * 1e3 / reps;
    printf("%-26s NS%d GM%-2d 128x128 occ1 M=%d N=%d K=%d: %8.1f us  %6.1f TF-eq  clock %.2f GHz  block life %.0f cyc\n", name,
           NS, GM, M, N, K, us, 2.0 * M * N * K / us / 1e6, cyc / wall * 0.1, cyc / tiles);
    fflush(stdout);
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
    const int M = 32768, Nmax = 4096, Kmax = 4096;
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
    float* C2; hipMalloc(&C2, (size_t)M * Nmax * 4);
    std::vector<float> c1((size_t)M * 1024), c2((size_t)M * 1024);
    auto differ = [&](float* X, float* Y, int MM, int N) {
        hipDeviceSynchronize();
        const size_t n = (size_t)MM * N;
        std::vector<float> x(n), y(n);
        hipMemcpy(x.data(), X, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(y.data(), Y, n * 4, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < n; ++i) bad += (x[i] != y[i]);
        if (bad) printf("  outputs differ at %zu of %zu elements\n", bad, n);
    };
#define QCASE(MM, N, K)                                                                             \
    run<128, 128, 16, 2, 8>("packed 2-stage occ2", A, W, C, MM, N, K, clk);                         \
    run_reg<128, 128, 2, 8>("packed reg-staged 2-ahead", A, W, C2, MM, N, K, clk); differ(C, C2, MM, N); \
    run_big<256, 256, 2, 4, 2, 4>("big 256x256 8w", A, W, C2, MM, N, K, clk); differ(C, C2, MM, N); \
    run_big<256, 256, 4, 2, 2, 4>("big 256x256 8w (4x2)", A, W, C2, MM, N, K, clk); differ(C, C2, MM, N); \
    run_big<256, 256, 2, 4, 2, 4, 1>("big 256x256 8w sched1", A, W, C2, MM, N, K, clk); differ(C, C2, MM, N); \
    run_big<256, 256, 2, 4, 2, 4, 2>("big 256x256 8w sched2", A, W, C2, MM, N, K, clk); differ(C, C2, MM, N); \
    run_big<256, 128, 4, 2, 3, 8, 1>("big 256x128 8w sched1", A, W, C2, MM, N, K, clk); differ(C, C2, MM, N); \
    run_big<256, 128, 4, 2, 3, 8>("big 256x128 8w", A, W, C2, MM, N, K, clk); differ(C, C2, MM, N); \
    run_big<128, 256, 2, 4, 3, 8>("big 128x256 8w", A, W, C2, MM, N, K, clk); differ(C, C2, MM, N); \
    run_big<256, 128, 2, 2, 3, 8>("big 256x128 4w", A, W, C2, MM, N, K, clk); differ(C, C2, MM, N); \
    run_big<128, 128, 2, 2, 4, 8>("big 128x128 4w", A, W, C2, MM, N, K, clk); differ(C, C2, MM, N);
#define PCASE(MM, N, K)                                                                             \
    run<128, 128, 16, 2, 8>("packed 2-stage occ2", A, W, C, MM, N, K, clk);                         \
    run_pp<6, 4>("ping-pong 8-phase", A, W, C2, MM, N, K, clk); differ(C, C2, MM, N);                \
    run_pp<5, 4>("ping-pong 8-phase", A, W, C2, MM, N, K, clk); differ(C, C2, MM, N);                \
    run_pp<6, 4, 1>("8-phase, no setprio", A, W, C2, MM, N, K, clk); differ(C, C2, MM, N);           \
    run_pp<6, 4, 2>("8-phase, rows in lockstep", A, W, C2, MM, N, K, clk); differ(C, C2, MM, N);     \
    run_pp<6, 8>("ping-pong 8-phase", A, W, C2, MM, N, K, clk); differ(C, C2, MM, N);                \
    run_pp<6, 4, 7>("8-phase, balanced reads", A, W, C2, MM, N, K, clk); differ(C, C2, MM, N);       \
    run_pp<7, 4, 7>("8-phase, balanced reads", A, W, C2, MM, N, K, clk); differ(C, C2, MM, N);
    // M = 16384 = whole rounds for every tile (what a balanced launch would give the big tiles), 16960 = the real shape
    if (!getenv("PROBE_PP_ONLY")) {
    QCASE(16384, 1024, 1024)
    QCASE(16384, 3072, 1024)
    QCASE(16384, 1024, 4096)
    QCASE(16960, 1024, 1024)
    }
    if (getenv("PROBE_PS_EXTRA")) {      // what the 23 padding rows of the 288-row tile cost (timing model, see ps_kernel)
#define ECASE(MM, N, K)                                                                             \
    run_ps<4, 2>("per-sample, 288 rows", A, W, C2, MM, N, K, clk);                                   \
    run_ps<4, 1>("per-sample, 272-row model", A, W, C2, MM, N, K, clk);                              \
    run_ps<4, 0>("per-sample, 256-row model", A, W, C2, MM, N, K, clk);                              \
    run_ps<4, 2>("per-sample, 288 rows", A, W, C2, MM, N, K, clk);
    ECASE(16960, 1024, 1024)
    ECASE(16960, 3072, 1024)
    ECASE(16960, 4096, 1024)
    ECASE(16960, 1024, 4096)
    return 0;
    }
    if (getenv("PROBE_PS")) {
#define SCASE(MM, N, K)                                                                             \
    hipMemset(C, 0xff, (size_t)MM * N * 4); hipMemset(C2, 0xff, (size_t)MM * N * 4);                \
    run<128, 128, 16, 2, 8>("packed 2-stage occ2", A, W, C, MM, N, K, clk);                         \
    run_ps<4>("per-sample ping-pong", A, W, C2, MM, N, K, clk); differ(C, C2, MM, N);               \
    run_ps<8>("per-sample ping-pong", A, W, C2, MM, N, K, clk); differ(C, C2, MM, N);               \
    run_ps<2>("per-sample ping-pong", A, W, C2, MM, N, K, clk); differ(C, C2, MM, N);
    SCASE(16960, 1024, 1024)
    SCASE(16960, 3072, 1024)
    SCASE(16960, 4096, 1024)
    SCASE(16960, 1024, 4096)
    SCASE(2 * 265, 1024, 1024)
    SCASE(3 * 265, 256, 64)
    return 0;
    }
    if (!getenv("PROBE_NO_PP")) {
    PCASE(16384, 1024, 1024)
    PCASE(16384, 3072, 1024)
    PCASE(16384, 4096, 1024)
    PCASE(16384, 1024, 4096)
    PCASE(16960, 1024, 1024)
    PCASE(16960, 3072, 1024)
    }
    // the tail program of the balanced big-tile launch alone: 576 rows (B = 64) on 8-wave 128x128 tiles, 2 vs 4 stages
#define TCASE(N, K)                                                                                 \
    run_big<128, 128, 2, 4, 2, 8>("tail 128x128 8w", A, W, C, 576, N, K, clk);                      \
    run_big<128, 128, 2, 4, 4, 8>("tail 128x128 8w", A, W, C2, 576, N, K, clk); differ(C, C2, 576, N);
    TCASE(1024, 1024)
    TCASE(1024, 4096)
    TCASE(4096, 1024)
    return 0;
}
