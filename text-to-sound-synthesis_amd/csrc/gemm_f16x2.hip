// fp32-class dense GEMM on the fp16 matrix cores ("f16x2 split"): three MFMA passes per k-step.
//
//   C[m][n] = store( act( out_scale * sum_k A[m][k] * W'[n][k] + bias[n] ) + R[m][n] ),   W' = W * 2^s
//
// fp16 carries an 11-bit significand, so
//   a = a0 + a1,  a0 = fp16(a),  a1 = fp16(a - a0)           (a - a0 is exact in fp32)
// represents a to 22 bits, and  a*b = a0b0 + a0b1 + a1b0 + O(2^-22 |ab|): three v_mfma_f32_32x32x16_f16
// passes instead of six.  The per-product error (~3e-7 relative, random sign) is far below what the
// fp32 FMA chain loses to accumulation rounding over K = 1024..4096 terms (measured 2e-6), so the
// result is fp32-class; tests/test_hip_split_gemm.py measures both against float64.
// fp16's narrow exponent range is handled with exact power-of-two scaling: the weights are
// pre-multiplied by 2^s (s per matrix, max |W'| in [2^13, 2^14)), which puts w1 = fp16(W' - w0) in the
// normal range, and the epilogue multiplies by out_scale = 2^-s.  Activations are used unscaled: they
// must stay below 65504 (true for this network: LayerNorm outputs, attention outputs, GELU2 outputs);
// an a1 that falls into the fp16 subnormal range keeps an absolute precision of 2^-25.
//
// 256 threads = 4 waves (2x2), block tile BM x BN x 32, two fp16 planes per operand, unpadded 64-byte LDS
// rows with an XOR chunk swizzle (conflict-free ds_read_b128 and ds_write_b128), double-buffered LDS (one barrier per k-tile), two register sets
// so that global loads run two k-tiles ahead and the split + LDS write of the next tile is scheduled
// into the shadow of the current tile's MFMAs.
#include "common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define HBK 32
#define HLD 32  // halves per LDS row: unpadded 64-byte rows, 16-byte chunks XOR-swizzled by (row>>2)&3
// Swizzle: logical chunk c of row r lives at chunk c ^ ((r >> 2) & 3).  A ds_read_b128 lane group covers 16
// rows at one logical chunk -> 16 distinct 4-bank slots (conflict-free); a ds_write_b128 lane group covers
// two consecutive rows x 4 chunks -> 32 distinct banks.  (With the earlier 80-byte padded rows the reads were
// clean but every staging write was a 2-way conflict: SQ_LDS_BANK_CONFLICT = 1/3 of the LDS busy cycles.)

// AMODE 0: A is fp32 row-major and is split by the loader (register staging: global -> VGPR -> split -> ds_write);
//          W = two row-major fp16 planes [2][N][ldw].
// AMODE 2: A and W are packed split planes (common.h ds_packed_off: 16-row x 32-k tiles of 1 KB that already are
//          the LDS image) and are staged by LDS-DMA, one global_load_lds_dwordx4 per wave per KB.  Measured on the
//          denoiser's shapes (tools/probe): with row-major planes the loop was bound by the L2 -> CU path (64-byte
//          half-cacheline requests at a 2-8 KB row stride: ~13 TB/s delivered, as long as the whole kernel); packed
//          tiles fetch 1.6x faster and leave the MFMA pipe as the busiest unit.
typedef __attribute__((address_space(1))) const void* ds_gptr;
typedef __attribute__((address_space(3))) void* ds_lptr;

// the whole workgroup program for output tile `bid` of `nblk` (both as launched; remapped below): 4 waves (2 x 2),
// two LDS stages.  (Round 2 measured the 8-wave big-tile, register-staged and deeper-ring variants of this loop
// against it and against the per-sample ping-pong program of gemm_f16x2_ps.hip -- profiles/r02_probe_*.txt -- and
// removed them: none beat this loop by more than a few per cent on the shapes it still serves.)
// (Round 5 measured a 2-wave form of the 128 x 64 tile -- wave tile 64 x 64, a third fewer LDS fragment bytes per MFMA -- on the
// training step's packed shapes: 15-25 % SLOWER than this 4-wave form everywhere, profiles/r05n_*; removed.)
// (Round 6: WGM x WGN = the wave grid.  2 x 2 everywhere but the 96 x 128 tile, whose four waves sit side by side (1 x 4: wave
// tile 96 x 32) -- M = 5 300 training rows x N = 1024 columns are 448 such tiles = ONE round of the 512 resident slots, where
// 128 x 128 leaves a third of the slots empty and 128 x 64 needs two rounds.)
template <int BM, int BN, int AMODE, int WGM = 2, int WGN = 2>
__device__ __forceinline__ void ds_gemm_f16x2_body(const GemmParams& p, int bid, const int nblk,
                                                   unsigned char* smem_raw) {
    constexpr int NS = 2;
    constexpr int NW = WGM * WGN, NT = NW * 64;
    static_assert(NW == 4, "256 threads");
    constexpr int TM = BM / (32 * WGM), TN = BN / (32 * WGN);
    static_assert(BM % (32 * WGM) == 0 && BN % (32 * WGN) == 0, "wave tiles are made of 32x32 blocks");
    static_assert(AMODE == 0 || AMODE == 2, "0: fp32 A split by the loader, 2: packed split planes by LDS-DMA");
    static_assert(AMODE == 2 || BM % 64 == 0, "register staging covers 64 rows per pass");
    constexpr int SA = BM / 64;   // 8-element (2 x float4) staging chunks per thread (A)
    constexpr int SB = BN / 64;   // 16-byte staging chunks per thread per plane (B)
    constexpr int APL = BM * HLD, BPL = BN * HLD;       // plane strides (halves)
    constexpr int STAGE = 2 * (APL + BPL);              // halves per pipeline stage
    _Float16* smem = (_Float16*)smem_raw;               // [2 stages]{ A[2][BM][HLD], B[2][BN][HLD] }

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wave / WGN, wn = wave % WGN;

    const int tiles_n = (p.N + BN - 1) / BN;
    {
        const int xcd = bid & 7, idx = bid >> 3;
        const int q = nblk >> 3, r = nblk & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // grouped raster inside the XCD's contiguous range: 8 row tiles x all column tiles per group, column-major
    // inside the group, so the ~64 workgroups resident on an XCD share each A / W k-slice 8 ways in its L2
    int m0, n0;
    {
        const int tiles_m = (p.M + BM - 1) / BM;
        const int per = 8 * tiles_n, grp = bid / per, first = grp * 8;
        const int gsz = tiles_m - first < 8 ? tiles_m - first : 8;
        const int in = bid - grp * per;
        m0 = (first + in % gsz) * BM;
        n0 = (in / gsz) * BN;
    }

    // A and B use the same chunking: chunk c = tid + 256 i covers row c>>2 and the 8 consecutive k at (c&3)*8
    // (A: two float4 loads -> one 16-byte ds_write per fp16 plane; 4 lanes cover one 128-byte row segment)
    const float* a_base[SA];
    int a_row[SA], a_k8[SA];
#pragma unroll
    for (int i = 0; i < SA; ++i) {
        const int c = tid + 256 * i;
        a_row[i] = c >> 2;
        const int ck = c & 3;
        a_k8[i] = (ck ^ ((a_row[i] >> 2) & 3)) * 8;          // swizzled LDS position of this chunk
        int m = m0 + a_row[i];
        if (m >= p.M) m = p.M - 1;
        a_base[i] = p.A + (size_t)m * p.lda + ck * 8;
    }
    const unsigned short* w2 = (const unsigned short*)p.W;
    const unsigned short* b_base[SB];
    int b_row[SB], b_k8[SB];
#pragma unroll
    for (int j = 0; j < SB; ++j) {
        const int c = tid + 256 * j;
        b_row[j] = c >> 2;
        const int ck = c & 3;
        b_k8[j] = (ck ^ ((b_row[j] >> 2) & 3)) * 8;
        int n = n0 + b_row[j];
        if (n >= p.N) n = p.N - 1;
        b_base[j] = w2 + (size_t)n * p.ldw + ck * 8;
    }

    // fragment reads: k-step ks, lane half hh -> logical chunk 2ks+hh; every fragment row of this lane is
    // l31 plus a multiple of 32, so the swizzle term depends on l31 only
    const int swz[2] = {((0 + hh) ^ ((l31 >> 2) & 3)) * 8, ((2 + hh) ^ ((l31 >> 2) & 3)) * 8};
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Two register sets: while the MFMAs of k-tile t run, set X (tile t+1, loaded one iteration ago and
    // therefore already landed) is split and written to the other LDS stage, and set Y receives the
    // global loads of tile t+2.  The body is branch-free (past-the-end tiles re-read the last tile and
    // write an unused stage), so the scheduler can interleave the split/ds_write VALU work with the MFMAs.
    f32x4 raX[2 * SA], raY[2 * SA];
    u32x4 rb0X[SB], rb1X[SB], rb0Y[SB], rb1Y[SB];
    const size_t pl1 = (size_t)p.w3_plane;
#define H_ISSUE_LOADS(RA, RB0, RB1, k0_)                                                            \
    do {                                                                                            \
        _Pragma("unroll") for (int i = 0; i < SA; ++i) {                                            \
            RA[2 * i] = *(const f32x4*)(a_base[i] + (k0_));                                         \
            RA[2 * i + 1] = *(const f32x4*)(a_base[i] + (k0_) + 4);                                 \
        }                                                                                           \
        _Pragma("unroll") for (int j = 0; j < SB; ++j) {                                            \
            RB0[j] = *(const u32x4*)(b_base[j] + (k0_));                                            \
            RB1[j] = *(const u32x4*)(b_base[j] + pl1 + (k0_));                                      \
        }                                                                                           \
    } while (0)
#define H_WRITE_LDS(RA, RB0, RB1, stage_)                                                           \
    do {                                                                                            \
        _Float16* As_ = smem + (stage_) * STAGE;                                                    \
        _Float16* Bs_ = As_ + 2 * APL;                                                              \
        _Pragma("unroll") for (int i = 0; i < SA; ++i) {                                            \
            _Float16* dst = As_ + a_row[i] * HLD + a_k8[i];                                         \
            h8 s0, s1;                                                                              \
            _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                         \
                const float a = RA[2 * i + (e >> 2)][e & 3];                                        \
                s0[e] = ds_split_hi(a);                                                             \
                s1[e] = ds_split_lo(a, s0[e]);                                                      \
            }                                                                                       \
            *(h8*)(dst) = s0;                                                                       \
            *(h8*)(dst + APL) = s1;                                                                 \
        }                                                                                           \
        _Pragma("unroll") for (int j = 0; j < SB; ++j) {                                            \
            _Float16* dst = Bs_ + b_row[j] * HLD + b_k8[j];                                         \
            *(u32x4*)(dst) = RB0[j];                                                                \
            *(u32x4*)(dst + BPL) = RB1[j];                                                          \
        }                                                                                           \
    } while (0)
#define H_COMPUTE(cur_)                                                                             \
    do {                                                                                            \
        const _Float16* Ac = smem + (cur_) * STAGE + (wm * TM * 32 + l31) * HLD;                    \
        const _Float16* Bc = smem + (cur_) * STAGE + 2 * APL + (wn * TN * 32 + l31) * HLD;          \
        _Pragma("unroll") for (int ks = 0; ks < HBK / 16; ++ks) {                                   \
            h8 fa0[TM], fa1[TM], fb0[TN], fb1[TN];                                                  \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                        \
                fa0[i] = *(const h8*)(Ac + i * 32 * HLD + swz[ks]);                                 \
                fa1[i] = *(const h8*)(Ac + APL + i * 32 * HLD + swz[ks]);                           \
            }                                                                                       \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                        \
                fb0[j] = *(const h8*)(Bc + j * 32 * HLD + swz[ks]);                                 \
                fb1[j] = *(const h8*)(Bc + BPL + j * 32 * HLD + swz[ks]);                           \
            }                                                                                       \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                          \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                    \
                    f32x16 c = acc[i][j];                                                           \
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa1[i], fb0[j], c, 0, 0, 0);         \
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0[i], fb1[j], c, 0, 0, 0);         \
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0[i], fb0[j], c, 0, 0, 0);         \
                    acc[i][j] = c;                                                                  \
                }                                                                                   \
        }                                                                                           \
    } while (0)
    // one k-tile: X holds tile kt+1, Y receives tile kt+2
#define H_BODY(RAX, RB0X, RB1X, RAY, RB0Y, RB1Y)                                                    \
    do {                                                                                            \
        const int kn2 = (kt + 2 < nk ? kt + 2 : nk - 1) * HBK;                                      \
        H_ISSUE_LOADS(RAY, RB0Y, RB1Y, kn2);                                                        \
        H_COMPUTE(cur);                                                                             \
        H_WRITE_LDS(RAX, RB0X, RB1X, cur ^ 1);                                                      \
        INTERLEAVE_HINTS();                                                                         \
        __syncthreads();                                                                            \
        cur ^= 1;                                                                                   \
        ++kt;                                                                                       \
    } while (0)
    // ask the scheduler for  MFMA, 3 x VALU, (DS write)  groups: the split runs in the MFMA shadow
#define INTERLEAVE_HINTS()                                                                          \
    do {                                                                                            \
        _Pragma("unroll") for (int q = 0; q < 3 * TM * TN * (HBK / 16); ++q) {                      \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); /* MFMA */                           \
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0); /* VALU */                           \
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); /* DS write */                       \
        }                                                                                           \
    } while (0)

    const int nk = p.K / HBK;
    if constexpr (AMODE == 2) {
        // LDS image of a stage = 2*(BM+BN) rows of 64 B: A hi rows, A lo rows, B hi rows, B lo rows.  One DMA
        // instruction of a wave fills 16 consecutive rows = one packed 16-row x 32-k tile (lane l -> bytes 16 l);
        // the wave owns the 16-row groups g = wave + NW i.
        constexpr int G = 2 * (BM + BN) / 16 / NW;
        static_assert(G * NW * 16 == 2 * (BM + BN), "16-row groups must divide over the waves");
        const _Float16* src[G];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            int r = 16 * (wave + NW * i);                    // first row of the group within the stage image
            const _Float16* base;
            int rg, rgs;
            if (r < 2 * BM) {
                base = (const _Float16*)p.A;
                if (r >= BM) { r -= BM; base += p.a_plane; }
                rg = (m0 + r) >> 4; rgs = (p.M + 15) >> 4;
            } else {
                r -= 2 * BM;
                base = (const _Float16*)p.W;
                if (r >= BN) { r -= BN; base += pl1; }
                rg = (n0 + r) >> 4; rgs = (p.N + 15) >> 4;
            }
            if (rg >= rgs) rg = rgs - 1;                     // tail groups re-read the last group (never stored)
            // row-group stride = the k-tiles the planes were PACKED with (lda = ldw; > K for a K-range of a split-K launch)
            src[i] = base + (size_t)rg * (p.lda / HBK) * 512 + lane * 8;
        }
#define H_DMA(stage_, k0_)                                                                          \
    do {                                                                                            \
        unsigned char* d_ = smem_raw + (stage_) * (STAGE * 2) + wave * 1024;                        \
        _Pragma("unroll") for (int i = 0; i < G; ++i)                                               \
            __builtin_amdgcn_global_load_lds((ds_gptr)(src[i] + (k0_)), (ds_lptr)(d_ + i * (NW * 1024)), 16, 0, 0); \
    } while (0)
        // all 4 (TM + TN) fragment reads of the k-tile are issued back to back, then its 6 TM TN MFMAs: the LDS
        // latency is paid once per tile and the other resident workgroups' MFMAs fill it
#define H_COMPUTE_ALL(cur_)                                                                         \
    do {                                                                                            \
        const _Float16* Ac = smem + (cur_) * STAGE + (wm * TM * 32 + l31) * HLD;                    \
        const _Float16* Bc = smem + (cur_) * STAGE + 2 * APL + (wn * TN * 32 + l31) * HLD;          \
        h8 fa0[2][TM], fa1[2][TM], fb0[2][TN], fb1[2][TN];                                          \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                          \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                        \
                fa0[ks][i] = *(const h8*)(Ac + i * 32 * HLD + swz[ks]);                             \
                fa1[ks][i] = *(const h8*)(Ac + APL + i * 32 * HLD + swz[ks]);                       \
            }                                                                                       \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                        \
                fb0[ks][j] = *(const h8*)(Bc + j * 32 * HLD + swz[ks]);                             \
                fb1[ks][j] = *(const h8*)(Bc + BPL + j * 32 * HLD + swz[ks]);                       \
            }                                                                                       \
        }                                                                                           \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                            \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                          \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                    \
                    f32x16 c = acc[i][j];                                                           \
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa1[ks][i], fb0[ks][j], c, 0, 0, 0); \
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0[ks][i], fb1[ks][j], c, 0, 0, 0); \
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0[ks][i], fb0[ks][j], c, 0, 0, 0); \
                    acc[i][j] = c;                                                                  \
                }                                                                                   \
        __builtin_amdgcn_sched_group_barrier(0x100, 4 * (TM + TN), 0); /* DS reads */               \
        __builtin_amdgcn_sched_group_barrier(0x008, 6 * TM * TN, 0);   /* MFMA */                   \
    } while (0)
        static_assert(HBK == 32, "two 16-wide k-steps per tile");
        H_DMA(0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            // Tile kt must have landed before anyone crosses the barrier.  The wait is written out: hipcc adds a
            // vmcnt(0) for an in-flight LDS-DMA on its own in some instantiations of this body but NOT in others
            // (the 128x128 program inlined into ds_gemm_f16x2_hybrid_kernel got `s_waitcnt lgkmcnt(0)` only -- a
            // silent race that corrupted large grids; tests/test_hip_split_gemm.py sweeps batch sizes for it).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();   // ... and stage (kt+1)&1 is free
            if (kt + 1 < nk) H_DMA((kt + 1) & 1, (kt + 1) * 512);
            H_COMPUTE_ALL(kt & 1);
        }
    } else {
        const int nk = p.K / HBK;
        H_ISSUE_LOADS(raX, rb0X, rb1X, 0);
        H_WRITE_LDS(raX, rb0X, rb1X, 0);
        H_ISSUE_LOADS(raX, rb0X, rb1X, (nk > 1 ? 1 : 0) * HBK);
        __syncthreads();

        int cur = 0, kt = 0;
        while (kt + 1 < nk) {
            H_BODY(raX, rb0X, rb1X, raY, rb0Y, rb1Y);
            H_BODY(raY, rb0Y, rb1Y, raX, rb0X, rb1X);
        }
        if (kt < nk) H_BODY(raX, rb0X, rb1X, raY, rb0Y, rb1Y);
    }

    // Epilogue.  One fully unrolled, branch-nested (no `continue`) copy of the loop per store family: with the
    // destinations of all families inside one loop body the accumulators were demoted to scratch (320 B/lane,
    // every GEMM 3x slower).
    const float osc = p.out_scale;
#define H_EPILOGUE(...)                                                                             \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                            \
            const int col = n0 + (wn * TN + j) * 32 + l31;                                          \
            if (col < p.N) {                                                                        \
                const float bv = p.bias ? p.bias[col] : 0.f;                                        \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                    \
                    const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;      \
                    if (row < p.M) {                                                                \
                        float v = acc[i][j][r] * osc + bv;                                          \
                        if (p.act == DS_ACT_GELU2) v = ds_gelu2_fast(v);                            \
                        __VA_ARGS__                                                                 \
                    }                                                                               \
                }                                                                                   \
            }                                                                                       \
        }                                                                                           \
    }
    // The staged epilogues reuse the operand stages (NS * STAGE halves) as the tile buffer.  A tile that does not fit
    // (256x256: 256 KB of fp32 against 128 KB of stages) goes through in SLABS row slabs of SR rows, each filled by
    // the waves whose rows lie in it; SLABS = 1 for every 4-wave instantiation.
    constexpr int LDS_BYTES = NS * STAGE * 2;
    constexpr int SLABS = (BM * BN * 4 + LDS_BYTES - 1) / LDS_BYTES;
    static_assert(WGM % SLABS == 0, "a wave's rows must lie in one slab");
    constexpr int SR = BM / SLABS;
    const int my_slab = SLABS == 1 ? 0 : wm / (WGM / SLABS);
    if (p.store == DS_STORE_ATTN || p.c_split) {
        // fp16 split outputs (packed planes for the next GEMM, or attention-ready Q / K / V^T): the tile is split,
        // staged in LDS as T[plane][SR][BN] halves and leaves as 16-byte stores -- 2-byte stores straight from the
        // accumulator layout cost +3..+25 % of the GEMM (one L2 write request per few bytes)
      for (int sl = 0; sl < SLABS; ++sl) {
        const int ms = m0 + sl * SR;                      // first row of this slab
        __syncthreads();                                  // every wave is done with the operand stages / the last slab
        _Float16* T = smem;
        if (my_slab == sl) {
            H_EPILOGUE({
                const _Float16 hi = ds_split_hi(v);
                const int tl = (row - ms) * BN + (col - n0);
                T[tl] = hi;
                T[SR * BN + tl] = ds_split_lo(v, hi);
            })
        }
        __syncthreads();
        const int hw = p.attn_heads * 64;
        const int which = p.c_split ? 0 : n0 / hw;        // block-uniform: Q, K or V columns
        if (p.c_split || which < 2) {                     // 8 consecutive columns of a row per store
            constexpr int CPR = BN / 8;
            for (int c = tid; c < 2 * SR * CPR; c += NT) {
                const int cc = c % CPR, rl = (c / CPR) % SR, pl = c / (CPR * SR);
                const int row = ms + rl, col = n0 + cc * 8;
                if (row < p.M && col < p.N) {
                    const u32x4 val = *(const u32x4*)(T + (pl * SR + rl) * BN + cc * 8);
                    _Float16* dst;
                    if (p.c_split) {
                        dst = (_Float16*)p.C + (size_t)pl * p.c_plane + ds_packed_off(row, col, p.ldc >> 5);
                    } else {
                        const int arow = row + p.row_off;
                        const int b = arow / p.rows_per_sample, pos = arow - b * p.rows_per_sample;
                        const int hc = col - which * hw, head = hc >> 6, d = hc & 63;
                        const size_t bh = (size_t)b * p.attn_heads + head;
                        if (which == 0)
                            dst = (_Float16*)p.C + (size_t)pl * p.attn_qplane + (bh * p.rows_per_sample + pos) * 64 + d;
                        else
                            dst = (_Float16*)p.attn_kv + (bh * 4 + pl) * ((size_t)p.attn_nkey * 64) + ds_attn_k_off(pos, d);
                    }
                    *(u32x4*)dst = val;
                }
            }
        } else {                                           // V^T: 8 consecutive keys of one d per store
            // The tile's rows belong to at most two samples; units of 8 keys are aligned in a sample's own key
            // index, so the first / last unit of each sample segment can be partial (2-byte stores for those).
            const int L = p.rows_per_sample, g0 = ms + p.row_off;        // absolute first row
            const int b0 = g0 / L, pos0 = g0 - b0 * L;
            const int rows_here = (p.M - ms < SR ? p.M - ms : SR);       // valid rows of this slab (may be <= 0)
            const int seg0 = (L - pos0 < rows_here ? L - pos0 : rows_here);   // rows in sample b0
            const int u0 = ((pos0 + seg0 + 7) >> 3) - (pos0 >> 3);       // units touching sample b0
            const int seg1 = rows_here - seg0;                           // rows in sample b0 + 1 (from key 0)
            const int units = (SLABS > 1 && rows_here <= 0) ? 0 : u0 + ((seg1 + 7) >> 3);   // slab past the last row
            const int pln = p.attn_nkey * 64;
            for (int c = tid; c < 2 * BN * units; c += NT) {
                const int cl = c % BN, u = (c / BN) % units, pl = c / (BN * units);
                const int col = n0 + cl;
                if (col < p.N) {
                    const bool first = u < u0;
                    const int b = first ? b0 : b0 + 1;
                    const int k0 = first ? ((pos0 >> 3) + u) * 8 : (u - u0) * 8;   // first key of the unit
                    const int rl0 = first ? k0 - pos0 : seg0 + k0;                 // its tile-local row (may be < 0)
                    const int lo_ok = first ? 0 : seg0, hi_ok = first ? seg0 : rows_here;   // local rows of this sample
                    const int hc = col - 2 * hw, head = hc >> 6, d = hc & 63;
                    _Float16* dst = (_Float16*)p.attn_kv + (((size_t)b * p.attn_heads + head) * 4 + 2 + pl) * (size_t)pln +
                                    ds_attn_vt_off(k0, d, p.attn_nkey);
                    const _Float16* src = T + (pl * SR) * BN + cl;
                    if (rl0 >= lo_ok && rl0 + 8 <= hi_ok) {
                        h8 val;
#pragma unroll
                        for (int e = 0; e < 8; ++e) val[e] = src[(rl0 + e) * BN];
                        *(h8*)dst = val;
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (rl0 + e >= lo_ok && rl0 + e < hi_ok) dst[e] = src[(rl0 + e) * BN];
                    }
                }
            }
        }
      }
    } else if (p.store == DS_STORE_ROW && ((p.N | p.ldc | p.ldr) & 3) == 0 &&
               (((uintptr_t)p.C | (uintptr_t)p.R) & 15) == 0) {
        // row-major fp32 (+ residual): the same staging, T[SR][BN] floats, 16-byte residual loads and stores
      for (int sl = 0; sl < SLABS; ++sl) {
        const int ms = m0 + sl * SR;
        __syncthreads();
        float* Tf = (float*)smem;
        if (my_slab == sl) {
            H_EPILOGUE({ Tf[(row - ms) * BN + (col - n0)] = v; })
        }
        __syncthreads();
        constexpr int CPR = BN / 4;
        for (int c = tid; c < SR * CPR; c += NT) {
            const int cc = c % CPR, rl = c / CPR;
            const int row = ms + rl, col = n0 + cc * 4;
            if (row < p.M && col < p.N) {
                f32x4 val = *(const f32x4*)(Tf + rl * BN + cc * 4);
                if (p.R) val += *(const f32x4*)(p.R + (size_t)row * p.ldr + col);
                *(f32x4*)(p.C + (size_t)row * p.ldc + col) = val;
            }
        }
      }
    } else if (p.store == DS_STORE_ROW) {
        H_EPILOGUE({
            if (p.R) v += p.R[(size_t)row * p.ldr + col];
            p.C[(size_t)row * p.ldc + col] = v;
        })
    } else {                                    // DS_STORE_BATCH_T
        H_EPILOGUE({
            const int b = row / p.rows_per_sample, pp = row - b * p.rows_per_sample;
            p.C[((size_t)b * p.N + col) * p.ldc + pp] = v;
        })
    }
}

// blockIdx.y = group: A / W / C advance by the group strides -- the training step's split-K dW GEMMs are `groups` K-ranges
// of one product (a_gstride = w_gstride = K per group for row-major operands; 16 K halves = K / 32 k-tiles for packed
// operands, whose row groups stay lda / 32 k-tiles apart; partial results c_gstride apart)
template <int BM, int BN, int AMODE, int WGM = 2, int WGN = 2>
__global__ __launch_bounds__(256, 2) void ds_gemm_f16x2_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
    if (blockIdx.y != 0) {                // (0 in an ungrouped launch)
        GemmParams q = p;
        const size_t g = blockIdx.y;
        q.A = AMODE == 0 ? p.A + g * (size_t)p.a_gstride : (const float*)((const _Float16*)p.A + g * (size_t)p.a_gstride);
        q.W = (const float*)((const _Float16*)p.W + g * (size_t)p.w_gstride);
        q.C = p.C + g * (size_t)p.c_gstride;
        ds_gemm_f16x2_body<BM, BN, AMODE, WGM, WGN>(q, blockIdx.x, gridDim.x, smem_dyn);
        return;
    }
    ds_gemm_f16x2_body<BM, BN, AMODE, WGM, WGN>(p, blockIdx.x, gridDim.x, smem_dyn);
}

// Balanced launch for packed operands: the first `nbig` workgroups compute 128x128 tiles of the leading rows
// (pb: a whole number of rounds of the chip's resident-workgroup slots), the rest 64x64 tiles of the remaining
// rows (ps: the same problem with the row origin moved).  At M = 16960, N = 1024 a plain 128x128 grid is 1064
// tiles = 2.08 rounds of 512 slots and the 40 stragglers cost a full extra tile time (255 vs 303 TF-eq measured
// against an exactly divisible M, tools/probe); the quarter-size tail tiles fill that hole instead.
__global__ __launch_bounds__(256, 2) void ds_gemm_f16x2_hybrid_kernel(const GemmParams pb, const GemmParams ps,
                                                                     const int nbig) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
    const int bid = blockIdx.x;   // uniform branch: a workgroup runs one of the two programs
    if (bid < nbig) ds_gemm_f16x2_body<128, 128, 2>(pb, bid, nbig, smem_dyn);
    else ds_gemm_f16x2_body<64, 64, 2>(ps, bid - nbig, (int)gridDim.x - nbig, smem_dyn);
}

// Several INDEPENDENT products of one tile configuration in one grid (round 6: the training step's weight gradients).  Every dW
// launch of the step is sized to <= 256 workgroups -- one per CU -- and a workgroup alone on a CU runs its tile in ~0.6 of the
// time two co-resident ones take (profiles/r06x_gemm_tile_batch_sweep.txt): two such launches back to back cost 2.0 units,
// their 512 workgroups side by side ~1.7.  The weight gradients are off the backward's critical path, so the step collects
// them per block and launches the ones of equal tile / K-range count together.  blockIdx.x ranges [first[i], first[i + 1])
// belong to problem i; blockIdx.y = the K-range (all problems of a launch have the same count).
extern int g_last_tile;
#define DS_GEMM_MULTI_MAX 4
struct GemmMulti {
    GemmParams p[DS_GEMM_MULTI_MAX];
    int first[DS_GEMM_MULTI_MAX + 1];
    int n;
};
template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256, 2) void ds_gemm_f16x2_multi_kernel(const GemmMulti m) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
    int i = 0;
    while (i + 1 < m.n && (int)blockIdx.x >= m.first[i + 1]) ++i;        // (block-uniform: scalar loads of the argument block)
    GemmParams q = m.p[i];
    const size_t g = blockIdx.y;
    q.A = (const float*)((const _Float16*)q.A + g * (size_t)q.a_gstride);
    q.W = (const float*)((const _Float16*)q.W + g * (size_t)q.w_gstride);
    q.C = q.C + g * (size_t)q.c_gstride;
    ds_gemm_f16x2_body<BM, BN, 2, WGM, WGN>(q, (int)blockIdx.x - m.first[i], m.first[i + 1] - m.first[i], smem_dyn);
}

template <int BM, int BN, int WGM, int WGN>
static int launch_multi(const GemmMulti& m, int groups, hipStream_t s) {
    const size_t lds = (size_t)2 * 2 * (BM + BN) * HLD * sizeof(unsigned short);
    static DsOnce attr_set;
    if (attr_set.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)ds_gemm_f16x2_multi_kernel<BM, BN, WGM, WGN>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            ds_set_error("gemm_f16x2 multi: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return -2;
        }
        attr_set.done();
    }
    hipLaunchKernelGGL((ds_gemm_f16x2_multi_kernel<BM, BN, WGM, WGN>), dim3(m.first[m.n], groups), dim3(256), lds, s, m);
    DS_CHECK_LAUNCH();
    return 0;
}

// n <= 4 packed-operand products (row store, no bias / residual / activation, the same number of K-ranges each) as ONE grid of
// tile configuration cfg (0: 128 x 128, 1: 128 x 64, 3: 96 x 128); each product is what ds_launch_gemm_f16x2 computes for it
// with that tile forced -- the same bits.
int ds_launch_gemm_f16x2_multi(const GemmParams* ps, int n, int cfg, hipStream_t stream) {
    DS_CHECK_ARG(ps && n >= 1 && n <= DS_GEMM_MULTI_MAX, "1 .. 4 products per launch");
    DS_CHECK_ARG(cfg == 0 || cfg == 1 || cfg == 3, "tile configuration: 0 (128 x 128), 1 (128 x 64) or 3 (96 x 128)");
    const int bm = cfg == 3 ? 96 : 128, bn = cfg == 1 ? 64 : 128;
    GemmMulti m = {};
    m.n = n;
    const int groups = ps[0].groups > 1 ? ps[0].groups : 1;
    for (int i = 0; i < n; ++i) {
        const GemmParams& p = ps[i];
        DS_CHECK_ARG(p.A && p.W && p.C && p.M > 0 && p.N > 0 && p.K > 0 && p.K % HBK == 0, "K must be a positive multiple of 32");
        DS_CHECK_ARG(p.a_split == 1 && !p.c_split && p.store == DS_STORE_ROW && !p.R && !p.bias && p.act == DS_ACT_NONE && p.out_scale > 0.f,
                     "multi: packed operands, plain row store, no bias / residual / activation, out_scale set");
        DS_CHECK_ARG((p.groups > 1 ? p.groups : 1) == groups, "multi: every product has the same number of K-ranges");
        DS_CHECK_ARG(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.W & 15) == 0 && p.lda >= p.K && p.lda % HBK == 0 && p.ldw == p.lda &&
                         p.a_plane >= (long long)((p.M + 15) / 16) * 16 * p.lda && p.a_plane % 8 == 0 &&
                         p.w3_plane >= (long long)((p.N + 15) / 16) * 16 * p.lda && p.w3_plane % 8 == 0,
                     "multi: lda = ldw = the packed contraction length (>= K), planes of ceil16(rows) * lda halves");
        DS_CHECK_ARG(p.lda == p.K || groups > 1, "packed operands: lda > K only for the K-ranges of a split-K launch");
        DS_CHECK_ARG(groups == 1 || (p.a_gstride % 512 == 0 && p.w_gstride % 512 == 0 && p.c_gstride % 4 == 0),
                     "multi: K-range strides are whole k-tiles, 16-byte aligned partial results");
        DS_CHECK_ARG(((p.N | p.ldc) & 3) == 0 && ((uintptr_t)p.C & 15) == 0, "multi: N, ldc multiples of 4, C 16-byte aligned");
        m.p[i] = p;
        m.first[i + 1] = m.first[i] + ((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn);
    }
    g_last_tile = cfg == 3 ? 5 : cfg;
    switch (cfg) {
        case 0: return launch_multi<128, 128, 2, 2>(m, groups, stream);
        case 1: return launch_multi<128, 64, 2, 2>(m, groups, stream);
        default: return launch_multi<96, 128, 1, 4>(m, groups, stream);
    }
}

// Row partition of a balanced launch: the first m_off rows go to BM x BN tiles (nbig of them = a whole number of rounds
// of `slots` resident workgroups), the rows after them to tbm x tbn tail tiles (nsmall of them) in the same grid.
// nsmall == 0: one program over all rows.  Pure arithmetic (ds_gemm_f16x2_plan exposes it to the CPU tests).
struct BalancePlan { int m_off, nbig, nsmall; };
static BalancePlan ds_balance_plan(int M, int N, int BM, int BN, int slots, int tbm, int tbn, bool tail_ok) {
    const int tn = (N + BN - 1) / BN;
    int rb = M / BM;                                 // full BM-row tiles available
    while (rb > 0 && ((long)rb * tn) % slots != 0) --rb;
    BalancePlan pl;
    if (rb == 0 || rb * BM == M || !tail_ok) {
        pl.m_off = M;
        pl.nbig = ((M + BM - 1) / BM) * tn;
        pl.nsmall = 0;
    } else {
        pl.m_off = rb * BM;
        pl.nbig = rb * tn;
        pl.nsmall = ((M - pl.m_off + tbm - 1) / tbm) * ((N + tbn - 1) / tbn);
    }
    return pl;
}

template <int BM, int BN, int AMODE, int WGM = 2, int WGN = 2>
static int launch_h2(const GemmParams& p, hipStream_t s) {
    const size_t lds = (size_t)2 * 2 * (BM + BN) * HLD * sizeof(unsigned short);
    static DsOnce attr_set;
    if (attr_set.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)ds_gemm_f16x2_kernel<BM, BN, AMODE, WGM, WGN>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            ds_set_error("gemm_f16x2: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return -2;
        }
        attr_set.done();
    }
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    hipLaunchKernelGGL((ds_gemm_f16x2_kernel<BM, BN, AMODE, WGM, WGN>), dim3(tiles, p.groups > 1 ? p.groups : 1),
                       dim3(256), lds, s, p);
    DS_CHECK_LAUNCH();
    return 0;
}
static int g_force_tile_h = -1;
template <int BM, int BN>
static int launch_h(const GemmParams& p, hipStream_t s) {
    return p.a_split ? launch_h2<BM, BN, 2>(p, s) : launch_h2<BM, BN, 0>(p, s);
}

extern int g_last_tile;
extern "C" void ds_gemm_f16x2_force_tile(int t) { g_force_tile_h = t; }
int ds_gemm_f16x2_ps_choice(long B, long N, bool half_ok);                   // gemm_f16x2_ps.hip: the grid rule
// Whether ds_launch_gemm_f16x2 would run a per-sample program (full or half tiles) for a packed-operand GEMM over B samples
// of 272 rows and N columns (every other precondition met): the forced tile decides first, then the grid rule.  The
// denoiser driver sizes its activation rows with this (padded-row mode), so it cannot drift from the dispatch below.
bool ds_gemm_f16x2_ps_taken(int B, int N) {
    if (g_force_tile_h == 9 || g_force_tile_h == 10) return true;
    if (g_force_tile_h >= 0) return false;
    return ds_gemm_f16x2_ps_choice(B, N, true) != 0;
}
bool ds_gemm_f16x2_ps_applies(const GemmParams& p, bool need_full_grid);   // gemm_f16x2_ps.hip
bool ds_gemm_f16x2_ph_applies(const GemmParams& p);
int ds_gemm_f16x2_ps_pick(const GemmParams& p);
int ds_launch_gemm_f16x2_ps(const GemmParams& p, hipStream_t s);
int ds_launch_gemm_f16x2_ph(const GemmParams& p, hipStream_t s);

// resident 128x128 workgroups on the chip (256 CUs x 2); a test hook shrinks it so small shapes take the hybrid path
static int g_balance_slots = 512;
extern "C" void ds_gemm_f16x2_set_balance_slots(int n) { g_balance_slots = n > 0 ? n : 512; }

// 128x128 tiles for the largest row range that fills whole rounds of slots, 64x64 tiles for the rows after it
static int launch_hybrid(const GemmParams& p, hipStream_t s) {
    const BalancePlan pl = ds_balance_plan(p.M, p.N, 128, 128, g_balance_slots, 64, 64,
                                           p.store == DS_STORE_ROW || p.store == DS_STORE_ATTN);
    if (pl.nsmall == 0) return launch_h2<128, 128, 2>(p, s);
    const int m_off = pl.m_off;
    GemmParams pb = p, ps = p;
    pb.M = m_off;
    ps.M = p.M - m_off;
    const size_t rg = (size_t)m_off / 16;
    ps.A = (const float*)((const _Float16*)p.A + rg * (p.lda / HBK) * 512);         // packed planes: row-group offset
    if (p.store == DS_STORE_ATTN) ps.row_off = p.row_off + m_off;   // destinations are computed from absolute rows
    else if (p.c_split) ps.C = (float*)((_Float16*)p.C + rg * (p.ldc / 32) * 512);
    else ps.C = p.C + (size_t)m_off * p.ldc;
    if (p.R) ps.R = p.R + (size_t)m_off * p.ldr;
    const int nbig = pl.nbig, nsmall = pl.nsmall;
    const size_t lds = (size_t)2 * 2 * (128 + 128) * HLD * sizeof(unsigned short);
    static DsOnce attr_set;
    if (attr_set.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)ds_gemm_f16x2_hybrid_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            ds_set_error("gemm_f16x2: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return -2;
        }
        attr_set.done();
    }
    hipLaunchKernelGGL(ds_gemm_f16x2_hybrid_kernel, dim3(nbig + nsmall), dim3(256), lds, s, pb, ps, nbig);
    DS_CHECK_LAUNCH();
    return 0;
}

// the row partition a packed-operand launch of configuration cfg would use (no device work; CPU tests call it)
extern "C" int ds_gemm_f16x2_plan(int cfg, int M, int N, int store, int* m_off, int* nbig, int* nsmall) {
    DS_CHECK_ARG(M > 0 && N > 0 && m_off && nbig && nsmall, "bad arguments");
    const bool tail_ok = store == DS_STORE_ROW || store == DS_STORE_ATTN;
    BalancePlan pl;
    DS_CHECK_ARG(cfg == 0, "only configuration 0 (128x128 + 64x64 tail tiles) has a balanced launch");
    pl = ds_balance_plan(M, N, 128, 128, g_balance_slots, 64, 64, tail_ok);
    *m_off = pl.m_off; *nbig = pl.nbig; *nsmall = pl.nsmall;
    return 0;
}

// The tile a packed-operand launch of M x N (x groups K-ranges) takes when nothing is forced: 0 128x128 (balanced), 1 128x64,
// 2 64x64, 3 96x128.  Written from the batch sweep of round 6 (profiles/r06x_gemm_tile_batch_sweep.txt: M = 265 B rows,
// B = 4 .. 48, on the four layer shapes).  One workgroup alone on a CU runs a tile in ~0.6 of the time two co-resident ones
// take, so what counts is the most loaded CU: up to 256 tiles -> one each, up to 512 -> some CUs carry two.  The 96 x 128 tile
// turns 257 .. 341 tiles of 128 x 128 (two per CU somewhere, a third of the slots empty) into <= 512 tiles of 3/4 the work
// (-9 .. -18 %: every N = 1024 layer of the training step), and 129 .. 192 into <= 256 (-11 .. -19 %); 193 .. 256 tiles of
// 128 x 128 -- one per CU, most CUs busy -- beat 128 x 64 by 0 .. 6 % (the fc1 / fc2 dW); between 342 and 999 tiles the plain /
// balanced 128 x 128 launch beats 128 x 64 (0 .. -23 %).  K-range launches (groups > 1: the dW products) keep the round-5
// rule (profiles/r05g_train_gemm_packed_sweep.txt).  Exported: the training step asks it when it batches weight gradients.
extern "C" int ds_gemm_f16x2_auto_tile(int M, int N, int groups) {
    const int G = groups > 1 ? groups : 1;
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128) * G;
    if (G > 1) return t128 >= 700 ? 0 : (t128 >= 128 ? 1 : 2);
    if (t128 < 128) return 2;
    const long t96 = (long)((M + 95) / 96) * ((N + 127) / 128);
    return t96 <= 256 ? 3 : t128 <= 256 ? (t128 > 192 ? 0 : 1) : t96 <= 512 ? 3 : 0;
}

// p.W: 2 fp16 planes holding W * 2^s, plane stride p.w3_plane (halves); p.out_scale = 2^-s.  Row-major planes
// [N][ldw] with an fp32 A (a_split 0); packed split planes for both A and W when a_split is set.
int ds_launch_gemm_f16x2(const GemmParams& p, hipStream_t stream) {
    DS_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0 && p.K % HBK == 0, "K must be a positive multiple of 32");
    DS_CHECK_ARG(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.W & 15) == 0 && p.lda % 4 == 0, "alignment");
    DS_CHECK_ARG(p.ldw >= p.K && p.ldw % 8 == 0 && p.w3_plane % 8 == 0, "split-weight strides must be multiples of 8");
    DS_CHECK_ARG(!p.a_split || (p.lda >= p.K && p.lda % HBK == 0 && p.ldw == p.lda &&
                                p.a_plane >= (long long)((p.M + 15) / 16) * 16 * p.lda && p.a_plane % 8 == 0 &&
                                p.w3_plane >= (long long)((p.N + 15) / 16) * 16 * p.lda),
                 "packed operands: lda = ldw = the packed contraction length (>= K), planes of ceil16(rows) * lda halves");
    DS_CHECK_ARG(!p.a_split || p.lda == p.K || p.groups > 1, "packed operands: lda > K only for the K-ranges of a split-K launch");
    DS_CHECK_ARG(!p.c_split || (p.ldc % 32 == 0 && p.N <= p.ldc && p.c_plane >= (long long)((p.M + 15) / 16) * 16 * p.ldc &&
                                p.store == DS_STORE_ROW && !p.R),
                 "packed output: ldc % 32 == 0, plane of ceil16(M) * ldc halves, row store, no residual");
    DS_CHECK_ARG(p.store == DS_STORE_ROW || p.store == DS_STORE_BATCH_T || p.store == DS_STORE_ATTN,
                 "unsupported store mode");
    DS_CHECK_ARG(p.store != DS_STORE_ATTN ||
                     (p.a_split && p.attn_heads > 0 && p.N % (p.attn_heads * 64) == 0 && p.N / (p.attn_heads * 64) <= 3 &&
                      p.rows_per_sample >= 128 && p.attn_qplane > 0 && !p.R && !p.c_split &&
                      (p.N == p.attn_heads * 64 || (p.attn_kv && p.attn_nkey >= p.rows_per_sample && p.attn_nkey % 32 == 0))),
                 "attention store: packed operands, N = (1..3) * heads * 64, >= 128 rows per sample, Q plane stride, "
                 "K/V images with nkey >= rows per sample");
    DS_CHECK_ARG(!p.c_split || p.N % 8 == 0, "packed output needs N % 8 == 0");
    DS_CHECK_ARG(p.act == DS_ACT_NONE || p.act == DS_ACT_GELU2, "unsupported activation");
    DS_CHECK_ARG(p.out_scale > 0.f, "out_scale must be set (2^-s of the weight pre-scale)");
    DS_CHECK_ARG(p.groups <= 1 || (!p.c_split && p.store == DS_STORE_ROW && !p.R && !p.bias && p.w_gstride % 8 == 0 &&
                                   p.c_gstride % 4 == 0 && (p.a_split ? p.a_gstride % 512 == 0 && p.w_gstride % 512 == 0
                                                                      : p.a_gstride % 4 == 0)),
                 "groups: plain row store, no bias / residual, 16-byte aligned group strides (whole k-tiles for packed operands)");
    // Full-batch denoiser GEMMs (packed operands, one sample = 265 rows, N in 256-column tiles, a grid of whole
    // rounds of the 256 CUs): the per-sample ping-pong program of gemm_f16x2_ps.hip.  force_tile(9) takes it for
    // every shape it can compute (tests: small batches), force_tile(0 / 1 / 2) never.
    // force_tile(10): the half-tile program (272-row samples).
    {
        const int pick = p.groups > 1 ? 0 : g_force_tile_h < 0 ? ds_gemm_f16x2_ps_pick(p)
                       : g_force_tile_h == 9 ? (ds_gemm_f16x2_ps_applies(p, false) ? 1 : 0)
                       : g_force_tile_h == 10 ? (ds_gemm_f16x2_ph_applies(p) ? 2 : 0) : 0;
        if (pick == 1) {
            g_last_tile = 3;
            return ds_launch_gemm_f16x2_ps(p, stream);
        }
        if (pick == 2) {
            g_last_tile = 4;
            return ds_launch_gemm_f16x2_ph(p, stream);
        }
        // Round 6: the grid rule said no because the samples do not fill whole rounds of the chip -- but a LEADING run of
        // samples may: B = 20 samples x N = 4096 columns are 320 per-sample tiles = 1.25 rounds (0.62), while the first 16
        // samples are exactly one round.  Those go to the per-sample program, the rest (its own, smaller problem: the rules
        // below) follows in a second launch: the training step's fc1 forward / fc2 dX 148 -> ~130 us.  Row-major stores only
        // (the second launch's operand / output offsets are whole 16-row groups: 16 samples x 265 rows).
        if (pick == 0 && g_force_tile_h < 0 && p.groups <= 1 && p.a_split && p.rows_per_sample > 0 && p.store == DS_STORE_ROW &&
            !p.c_split && p.M % p.rows_per_sample == 0 && p.N % 256 == 0 && 256 % (p.N / 256) == 0) {
            const int L = p.rows_per_sample, B = p.M / L, per_round = 256 / (p.N / 256);
            const int nfull = B / per_round * per_round;
            GemmParams pa = p;
            pa.M = nfull * L;
            if (nfull > 0 && nfull < B && pa.M % 16 == 0 && ds_gemm_f16x2_ps_pick(pa) == 1) {
                g_last_tile = 3;
                const int rc = ds_launch_gemm_f16x2_ps(pa, stream);
                if (rc != 0) return rc;
                GemmParams pb = p;
                pb.M = p.M - pa.M;
                pb.A = (const float*)((const _Float16*)p.A + (size_t)(pa.M / 16) * (p.lda / HBK) * 512);
                pb.C = p.C + (size_t)pa.M * p.ldc;
                if (p.R) pb.R = p.R + (size_t)pa.M * p.ldr;
                return ds_launch_gemm_f16x2(pb, stream);
            }
        }
    }
    // Tile choice from the measured sweep (profiles/r01_gemm_tile_sweep_f16x2.txt, B=64): 128x128 reaches
    // ~235-250 TF-eq once the grid has >= 3 rounds of 512 resident blocks; below that 128x64 (3 blocks/CU,
    // ~205-230 TF-eq) quantises better; 64x64 (~190) only wins for tiny grids.
    int best;
    if (g_force_tile_h >= 0 && g_force_tile_h <= 3) {
        best = g_force_tile_h == 3 && !p.a_split ? 1 : g_force_tile_h;     // (the 96-row tile exists for packed operands only)
    } else if (p.a_split) {
        best = ds_gemm_f16x2_auto_tile(p.M, p.N, p.groups);
    } else {
        const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * (p.groups > 1 ? p.groups : 1);
        best = t128 >= 1500 ? 0 : (t128 >= 128 ? 1 : 2);         // (fp32 A split by the loader: register staging)
    }
    g_last_tile = best == 3 ? 5 : best;          // (3 and 4 name the per-sample programs above)
    switch (best) {
        case 0: return p.a_split && p.groups <= 1 ? launch_hybrid(p, stream) : launch_h<128, 128>(p, stream);
        case 1: return launch_h<128, 64>(p, stream);
        case 3: return launch_h2<96, 128, 2, 1, 4>(p, stream);
        default: return launch_h<64, 64>(p, stream);
    }
}
