// The denoiser's f16x2 GEMM at full batch: per-sample 288 x 256 tiles on an 8-phase ping-pong main loop.
//
//   C[m][n] = store( act( out_scale * sum_k A[m][k] * W'[n][k] + bias[n] ) + R[m][n] )        (gemm_f16x2.hip)
//
// Same arithmetic and, per accumulator, the same MFMA order as every program of gemm_f16x2.hip (ks 0 {a1 w0, a0 w1,
// a0 w0}, ks 1 {...}) -- results are bit-identical (tests/test_hip_split_gemm.py).  What differs is the decomposition:
//
// * Tile = ONE SAMPLE.  The activation matrices of the denoiser have M = B * L rows, L = 265 (the 5 x 53 token grid).
//   265 never divides into 128- or 256-row tiles: at B = 64, 256-row tiles are 66.25 row tiles = 1.05 / 3.14 / 4.19
//   rounds of the 256 CUs for N = 1024 / 3072 / 4096 (measured: the 256 x 256 ping-pong loop drops from 344 to 245
//   TF-eq at N = 1024).  A tile of 288 rows = 9 MFMA blocks that covers exactly the rows of one sample gives
//   B * N / 256 tiles: whole rounds at B = 64 for every GEMM of the network, no tail program, 265 / 288 = 92 % of the
//   MFMA work useful.  Tile rows are [m0, m0 + 288), m0 = 16 floor(L b / 16) (packed planes come in 16-row groups);
//   they contain the sample's rows [L b, L b + L) (needs (L b mod 16) + L <= 288: L <= 273); only those are stored.
// * 8 waves = 2 (M) x 4 (N), one workgroup per CU.  Blocks 0..7 of the tile: wave (wr, wc) owns rows wr 128 + [0, 128),
//   columns wc 64 + [0, 64) (4 x 2 blocks).  The NINTH block row (tile rows 256..287, 32 x 256) is split by COLUMNS over
//   all eight waves: wave (wr, wc) adds the 32 x 32 block at columns (2 wc + wr) 32 -- B-sub `wr` of its own column
//   range, whose fragments it already holds -- so every wave runs the same program: 54 MFMAs per k-tile, 144
//   accumulator registers.
// * Main loop: the "256^2 8-phase" structure of cdna_hip_programming.md section 5 for two fp16 planes x 32 k per
//   k-tile (= the bytes of a 64-wide bf16 k-step).  A k-tile is consumed in four phases (quadrants of the wave tile,
//   12 MFMAs each; phase 3 also reads and multiplies the ninth block) and staged by LDS-DMA in four quarters (A-sub0,
//   B-sub0, B-sub1, A-sub1 + the ninth block's rows), one per phase, LEAD = 6 quarters ahead of the phase that
//   computes.  The wait of a phase is a COUNTED vmcnt(9) -- the four youngest quarters (2 + 2 + 2 + 3 instructions per
//   wave) stay in flight across the barriers, never vmcnt(0) in the steady state.  Each phase = { ds_read a sub-tile;
//   issue a quarter; vmcnt; barrier; MFMAs under s_setprio(1); barrier }, and the two wave rows run one barrier
//   apart, so on every SIMD one wave issues MFMAs while its partner reads LDS and issues DMA.
//   Hazards (phase g = 4 tile + p; quarter q = 4 tile + type is first needed at phase 4 tile + {0, 0, 1, 2}[type]):
//     RAW  the wait of phase g retires this wave's quarters <= g + 2, the barrier behind it does so for every wave of
//          its row, the other row is at most one barrier away -> a quarter is read in a phase AFTER the one that
//          retires it (quarter 4t+3 incl. the ninth block's rows: retired in phase 4t+1, read in phases 4t+2 / 4t+3);
//     WAR  quarter g + 6 lands on the region of quarter g - 2, last read two or more phases before phase g.
//   Waves 4..7 repeat the ninth-block loads of waves 0..3 (same bytes to the same LDS address) so that every wave
//   counts the same number of instructions.
// * Measured (round 2, a standalone probe of this loop that has since been removed; M = 16960): 343 / 374 / 385 / 412 TF-eq at
//   (N, K) = (1024, 1024) / (3072, 1024) / (4096, 1024) / (1024, 4096) against 253 / 286 / 290 / 283 for the
//   128 x 128 two-workgroups-per-CU program on the same shapes (profiles/r02_probe_per_sample.txt).
// Epilogue: the three store families of gemm_f16x2.hip (row-major fp32 + residual; packed split planes; attention-ready
// Q / K / V^T), staged through the operand stages in three row slabs (tile rows 0..127: wave row 0, 128..255: wave row
// 1, 256..287: every wave's ninth block) and written with 16-byte stores.  One kernel instantiation per family.
#include <stdlib.h>

#include "common.h"

// (The ablation switches and in-kernel time stamps this file carried through rounds 3-5 -- PS_ABLATE / PS_TIMING -- live in
// tools/probe/ps_probe.patch: the probe builds apply it to a scratch copy of this file, tools/probe/README.md.  Nothing of
// them is left in the product source.)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* ds_gptr;
typedef __attribute__((address_space(3))) void* ds_lptr;

#define PS_HLD 32          // halves per LDS row (64-byte rows, chunks swizzled by (row >> 2) & 3: the packed-plane image)
#define PS_BM 288
#define PS_BN 256
#define PS_LEAD 6
#define PS_GM 4            // raster: groups of 4 sample tiles x all column tiles (measured: 4 beats 2 and 8 by ~10 %)
#define PS_STAGE_BYTES (2 * 2 * (PS_BM + PS_BN) * PS_HLD * 2)   // the main loop's two operand stages of 68 KB
#define PS_VT_BYTES (2 * 2 * PS_BN * 36 * 4)                    // the epilogue's transposed V^T staging: two buffers of 72 KB
#define PS_LDS_BYTES (PS_VT_BYTES > PS_STAGE_BYTES ? PS_VT_BYTES : PS_STAGE_BYTES)
// NB16 operands: lane quad q of the 16-row MFMA operand reads tile-row quad PS_SIG(q) = {0, 2, 3, 1}[q].  With the natural
// order the 16-lane service groups of ds_read_b128 ({0-3, 12-15, 20-27}, ...) hit each 16-byte slot of the packed image
// twice (rows 0-3 and 4-7 of k-chunks 0 and 1 share a slot: SQ_LDS_BANK_CONFLICT = 17 % of the loop's LDS cycles, round-3
// PMC pass); with this order every group covers the 16 slots once.  A permutation of the operand's ROWS only: each
// output element is the same sum in the same order, it just lives in another lane (the epilogue applies PS_SIG again).
#define PS_SIG(q_) ((0x78 >> (2 * (q_))) & 3)

enum { PS_EPI_ROW = 0, PS_EPI_SPLIT = 1, PS_EPI_ATTN = 2 };

// NB16: the sample has L = 272 rows (a whole number of 16-row packed groups: the denoiser's padded-row mode, api.hip), so
// the tile's ninth block row is the 16 rows 256..271 and runs on v_mfma_f32_16x16x32_f16 -- 6 MFMAs of half the cost per
// k-tile and wave instead of 6 of 32x32x16 (measured with a timing model in the probe: -4.6..5.4 % per launch).  The wave
// (wr, wc) owns the two 16 x 16 tiles at columns (2 wc + wr) 32 + {0, 16}; its B fragments in that instruction's layout are
// read in phase 1 (B-sub0 is re-staged from phase 3 on, B-sub1 from the next phase 0), the sixteen A rows in phase 2 (their
// quarter is retired in phase 1), and the MFMAs run in phase 2: 15 / 12 MFMA-equivalents in phases 2 / 3 instead of 12 / 18.
// One 32-k MFMA per product and k-tile instead of two 16-k ones: rows 256.. are NOT bit-identical to the 4-wave programs.
template <int EPI, bool NB16>
__global__ __launch_bounds__(512, 1) void ds_gemm_f16x2_ps_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int BM = PS_BM, BN = PS_BN, HLD = PS_HLD, LEAD = PS_LEAD;
    constexpr int APL = BM * HLD, BPL = BN * HLD, STAGE = 2 * (APL + BPL);   // halves; STAGE * 2 bytes = 68 KB
    _Float16* smem = (_Float16*)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int wr = wave >> 2, wc = wave & 3;
    const int L = p.rows_per_sample;
    const int tiles_n = p.N / BN, nblk = gridDim.x;
    int bid = blockIdx.x;
    {   // each XCD works a contiguous run of tiles
        const int xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tm_, tn_;
    {
        const int tiles_m = p.M / L;
        const int per = PS_GM * tiles_n, grp = bid / per, first = grp * PS_GM;
        const int gsz = tiles_m - first < PS_GM ? tiles_m - first : PS_GM;
        const int in = bid - grp * per;
        tm_ = first + in % gsz;
        tn_ = in / gsz;
    }
    const int row_lo = tm_ * L;                     // the sample's rows [row_lo, row_lo + L) are what this tile stores
    const int m0 = (row_lo >> 4) << 4, n0 = tn_ * BN;
    const int nk = p.K / 32;                        // even (K % 64 == 0)
    const _Float16* Ap = (const _Float16*)p.A;
    const _Float16* Wp = (const _Float16*)p.W;
    // wave-uniform 64-bit bases (SGPRs) + one per-lane byte offset keep the tile pointers out of the VGPR file
    unsigned long long src[4][2], src8;             // quarter types: 0 = A-sub0, 1 = B-sub0, 2 = B-sub1, 3 = A-sub1 (+ block 8)
    int ldsoff[4][2], ldsoff8;
    const unsigned lane16 = lane * 16;
    const int rgsA = (p.M + 15) >> 4, rgsB = (p.N + 15) >> 4;
    // (unsigned) on the builtin's result: it returns int, and the conversion to 64 bits would SIGN-extend the low word
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
            if (rg >= rgs) rg = rgs - 1;                               // groups past the end re-read the last one (never stored)
            PS_BASE(src[ty][k], (isA ? Ap + plane * p.a_plane : Wp + plane * p.w3_plane) + (size_t)rg * nk * 512);
            ldsoff[ty][k] = __builtin_amdgcn_readfirstlane((isA ? plane * 18 + gip : 36 + plane * 16 + gip) * 1024);
        }
    {   // block 8: groups 16, 17 of both planes = 4 KB; wave w (and w + 4) loads piece e = w & 3
        const int e = wave & 3, plane = e >> 1, gip = 16 + (e & 1);
        int rg = (m0 >> 4) + gip;
        if (rg >= rgsA) rg = rgsA - 1;
        PS_BASE(src8, Ap + plane * p.a_plane + (size_t)rg * nk * 512);
        ldsoff8 = __builtin_amdgcn_readfirstlane((plane * 18 + gip) * 1024);
    }
#define PS_ISSUE(tile_, ty_, buf_)                                                                   \
    do {                                                                                             \
        _Pragma("unroll") for (int k = 0; k < 2; ++k)                                                \
            __builtin_amdgcn_global_load_lds((ds_gptr)((const unsigned char*)(src[ty_][k] + (unsigned long long)(tile_) * 1024) + lane16), \
                                             (ds_lptr)(smem_raw + (buf_) * (STAGE * 2) + ldsoff[ty_][k]), 16, 0, 0); \
        if ((ty_) == 3)                                                                              \
            __builtin_amdgcn_global_load_lds((ds_gptr)((const unsigned char*)(src8 + (unsigned long long)(tile_) * 1024) + lane16), \
                                             (ds_lptr)(smem_raw + (buf_) * (STAGE * 2) + ldsoff8), 16, 0, 0); \
    } while (0)
    // The same transfers inside the main loop, written out: per transfer the builtin form costs ~7 scalar instructions (a
    // 64-bit tile address from the k-tile index: 4 adds + v_lshl_add_u64, the LDS address into M0, a wait state) -- 1.6e7
    // SALU instructions per FC1 launch, and the loop was 9 % longer with the DMA than without (round-3 SQ counters).  Here
    // every transfer keeps a RUNNING 64-bit base in SGPRs (it moves one k-tile = 1 KB per use), addresses  SGPR base +
    // lane offset, and gets its LDS address by one s_add into M0: four scalar instructions, no vector one.  (No immediate
    // offset: the instruction adds it to the LDS address as well.  hipcc does not use M0 in this loop otherwise.)
    // Same-run A/B against the builtin form (profiles/r03h_*): -2..4 % launch time on every shape.
#define PS_DMA_ASM(base_, lds_, stage_)                                                              \
    do {                                                                                             \
        asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"            \
                     :: "v"(lane16), "s"(base_), "s"(lds_), "n"(stage_) : "memory", "scc", "m0");    \
        base_ += 1024;                                                                               \
    } while (0)
    // tq_: the k-tile (run time, used by the builtin form)
#define PS_ISSUE_LOOP(tq_, ty_, buf_)                                                                \
    do {                                                                                             \
        PS_DMA_ASM(cur[ty_][0], ldsabs[ty_][0], (buf_) * (STAGE * 2));                               \
        PS_DMA_ASM(cur[ty_][1], ldsabs[ty_][1], (buf_) * (STAGE * 2));                               \
        if ((ty_) == 3) PS_DMA_ASM(cur8, ldsabs8, (buf_) * (STAGE * 2));                             \
    } while (0)
#define PS_FENCE()                                                                                   \
    do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PS_BAR()                                                                                     \
    do { PS_FENCE(); __builtin_amdgcn_s_barrier(); PS_FENCE(); } while (0)
#define PS_LBAR() PS_BAR()
    const int swz[2] = {((0 + hh) ^ ((l31 >> 2) & 3)) * 8, ((2 + hh) ^ ((l31 >> 2) & 3)) * 8};
    f32x16 acc[4][2], acc8;
    f32x4 acc9[2];                  // NB16: the two 16 x 16 tiles of the ninth block row
#pragma unroll
    for (int r = 0; r < 16; ++r) acc8[r] = 0.f;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) acc9[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    h8 a0[2][2], a1[2][2];          // [ks][row block of the current A-sub]: hi, lo planes
    h8 b0[2][2], b1[2][2];          // [B-sub][ks]: hi, lo planes
    h8 e0[2], e1[2];                // block 8 [ks]: hi, lo planes
    h8 ea0, ea1, eb0[2], eb1[2];    // NB16: A rows 256..271 and the wave's two 16-column B tiles, 16x16x32 operand layout
    const int l15 = PS_SIG((lane >> 2) & 3) * 4 + (lane & 3);   // the tile row / column this lane's operand row is (see PS_SIG)
    const int swzq = ((lane >> 4) ^ ((l15 >> 2) & 3)) * 8;     // lane group kq = lane >> 4 holds k = 8 kq .. 8 kq + 7
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
    // quadrant (A-sub sa, B-sub sb): per accumulator ks 0 {a1 b0, a0 b1, a0 b0}, ks 1 {...}
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
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                           \
            acc8 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e1[ks], b0[sb_][ks], acc8, 0, 0, 0);       \
            acc8 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0[ks], b1[sb_][ks], acc8, 0, 0, 0);       \
            acc8 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0[ks], b0[sb_][ks], acc8, 0, 0, 0);       \
        }                                                                                            \
    } while (0)
#define PS_READ_EB16(buf_)                                                                           \
    do {                                                                                             \
        const _Float16* Bc = smem + (buf_) * STAGE + 2 * APL + ((2 * wc + wr) * 32 + l15) * HLD;     \
        _Pragma("unroll") for (int tt = 0; tt < 2; ++tt) {                                           \
            eb0[tt] = *(const h8*)(Bc + tt * 16 * HLD + swzq);                                       \
            eb1[tt] = *(const h8*)(Bc + BPL + tt * 16 * HLD + swzq);                                 \
        }                                                                                            \
    } while (0)
#define PS_READ_EA16(buf_)                                                                           \
    do {                                                                                             \
        const _Float16* Ec = smem + (buf_) * STAGE + (256 + l15) * HLD;                              \
        ea0 = *(const h8*)(Ec + swzq);                                                               \
        ea1 = *(const h8*)(Ec + APL + swzq);                                                         \
    } while (0)
#define PS_EXTRA16()                                                                                 \
    do {                                                                                             \
        _Pragma("unroll") for (int tt = 0; tt < 2; ++tt) {                                           \
            acc9[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ea1, eb0[tt], acc9[tt], 0, 0, 0);      \
            acc9[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ea0, eb1[tt], acc9[tt], 0, 0, 0);      \
            acc9[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ea0, eb0[tt], acc9[tt], 0, 0, 0);      \
        }                                                                                            \
    } while (0)
    // one phase: P = phase within the k-tile (compile time), BUF = parity of the k-tile t (compile time)
#define PS_PHASE(P, BUF)                                                                             \
    do {                                                                                             \
        if (P == 0) { PS_READ_A(BUF, 0); PS_READ_B(BUF, 0); }                                        \
        if (P == 1) { PS_READ_B(BUF, 1); if (NB16) PS_READ_EB16(BUF); }                              \
        if (P == 2) { PS_READ_A(BUF, 1); if (NB16) PS_READ_EA16(BUF); }                              \
        if (P == 3 && !NB16) PS_READ_E(BUF);                                                         \
        PS_FENCE();                                                                                  \
        {                                                                                            \
            constexpr int dq = (P) + LEAD;                     /* quarter 4 t + dq */                \
            const int tq = t + (dq >> 2);                                                            \
            if (tq < nk) {                                                                           \
                PS_ISSUE_LOOP(tq, dq & 3, ((BUF) + (dq >> 2)) & 1);                                  \
                asm volatile("s_waitcnt vmcnt(9)" ::: "memory");   /* the 4 youngest quarters: 2+2+2+3 */ \
            } else {                                                                                 \
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   /* tail: nothing younger to count */ \
            }                                                                                        \
        }                                                                                            \
        PS_LBAR();                                                                                   \
        __builtin_amdgcn_s_setprio(1);                                                               \
        if (P == 0) PS_QUAD(0, 0);                                                                   \
        if (P == 1) PS_QUAD(0, 1);                                                                   \
        if (P == 2) { PS_QUAD(1, 1); if (NB16) PS_EXTRA16(); }                                       \
        if (P == 3) { PS_QUAD(1, 0); if (!NB16) { if (wr == 0) PS_EXTRA(0); else PS_EXTRA(1); } }    \
        __builtin_amdgcn_s_setprio(0);                                                               \
        PS_LBAR();                                                                                   \
    } while (0)
    // prologue: quarters 0 .. 5 (k-tile 0 and types 0, 1 of k-tile 1) = 13 instructions per wave; quarters 0 and 1 have
    // landed once only the 4 youngest (2 + 3 + 2 + 2 = 9) are outstanding
#pragma unroll
    for (int q = 0; q < LEAD; ++q)
        if ((q >> 2) < nk) PS_ISSUE(q >> 2, q & 3, (q >> 2) & 1);
    if (4 * nk >= LEAD) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PS_BAR();
    if (wr == 1) PS_LBAR();         // the second wave row runs one barrier behind the first
    // running tile bases of the written-out transfers: the first k-tile each quarter type is issued for inside the loop is
    // 1 (types 2, 3: phases 0, 1 of k-tile 0 issue quarters 6, 7) or 2 (types 0, 1: quarters 8, 9)
    unsigned long long cur[4][2], cur8 = src8 + 1024;
#pragma unroll
    for (int ty = 0; ty < 4; ++ty) {
        cur[ty][0] = src[ty][0] + (ty < 2 ? 2048 : 1024);
        cur[ty][1] = src[ty][1] + (ty < 2 ? 2048 : 1024);
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw);
    unsigned ldsabs[4][2], ldsabs8 = lds0 + ldsoff8;          // LDS byte address of every transfer in stage 0
#pragma unroll
    for (int ty = 0; ty < 4; ++ty) { ldsabs[ty][0] = lds0 + ldsoff[ty][0]; ldsabs[ty][1] = lds0 + ldsoff[ty][1]; }
    for (int t = 0; t < nk; t += 2) {
        PS_PHASE(0, 0); PS_PHASE(1, 0); PS_PHASE(2, 0); PS_PHASE(3, 0);
        ++t;
        PS_PHASE(0, 1); PS_PHASE(1, 1); PS_PHASE(2, 1); PS_PHASE(3, 1);
        --t;
    }
    if (wr == 0) PS_LBAR();         // ... and the first row waits for it at the end
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    constexpr int WROW = 128, EROW0 = 256;         // tile geometry of the full tile (gemm_f16x2_ps_epilogue.inc)
    constexpr bool HALF_TILE = false;
    const bool hasE = true;
    const int pos0 = 0, tile_rows = L;
#include "gemm_f16x2_ps_epilogue.inc"
}

// ---- half tiles: the same program for grids that full tiles cannot fill ------------------------------------------------
// BASELINE configs[1] runs 32 captions: the N = 1024 GEMMs are then 128 full tiles -- half of the CUs idle -- and the QKV
// GEMM 384 = 1.5 rounds (the round-2 build fell back to the 4-wave programs there: 0.29 of the pipe against 0.44 at batch
// 64).  A sample of 272 rows (padded-row mode) is cut into a 144-row tile (128 + the 16-row block on the 16x16x32 MFMA) and
// a 128-row tile: 2 B N / 256 tiles of HALF the accumulators (72 registers) -- whole rounds again at B = 32 (256 / 768).
// Same arithmetic, same MFMA order per accumulator as the full tile (rows 0..127 and 144..271 of a sample keep the bits of
// the 4-wave programs, rows 128..143 are the 16-row block).  What changes:
//   * a wave row is 64 tile rows (two 32-row blocks): ONE A sub-tile, so a k-tile is two phases -- (A, B-sub0) and
//     (A, B-sub1) + the 16-row block -- of 12 (+3) MFMAs, still alternating between the two wave rows;
//   * a stage is 144 + 256 rows = 50 KB, so THREE stages fit: the LDS-DMA of a k-tile is issued in two halves, H0 =
//     A + B-sub0 (4 instructions per wave) in phase 0 and H1 = B-sub1 + the 16 extra A rows (3) in phase 1, two k-tiles
//     ahead.  Hazards (phase g = 2 tile + p reads half g): RAW  the wait of phase g (after issuing half g + 4) is
//     vmcnt(11 / 10) = the instructions of halves g + 2 .. g + 4, so half g + 1 has landed a phase before it is read;
//     WAR  half g + 4 lands on the region of half g - 2, last read two phases earlier;
//   * the second tile of a sample has no 16-row block: it still issues (and never reads) the seventh DMA of H1 so that
//     every wave of every tile counts the same instructions, and skips the 16x16 MFMAs;
//   * more operand traffic per MFMA (the B tile is re-staged for 144 rows instead of 272: x1.46 L2 -> LDS bytes, x1.3
//     fragment reads) -- which is why full tiles keep every grid they can fill.
#define PH_AROWS 144
#define PH_STAGE_BYTES (2 * (PH_AROWS + PS_BN) * PS_HLD * 2)    // 50 KB
#define PH_LDS_BYTES (3 * PH_STAGE_BYTES > PS_VT_BYTES ? 3 * PH_STAGE_BYTES : PS_VT_BYTES)

template <int EPI>
__global__ __launch_bounds__(512, 1) void ds_gemm_f16x2_ph_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr bool NB16 = true;
    constexpr int BN = PS_BN, HLD = PS_HLD;
    constexpr int APL = PH_AROWS * HLD, BPL = BN * HLD, STAGE = 2 * (APL + BPL);   // halves
    _Float16* smem = (_Float16*)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int wr = wave >> 2, wc = wave & 3;
    const int L = p.rows_per_sample;                // 272
    const int tiles_n = p.N / BN, nblk = gridDim.x;
    int bid = blockIdx.x;
    {   // each XCD works a contiguous run of tiles
        const int xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int th_, tn_;
    {
        const int tiles_m = 2 * (p.M / L);
        const int per = PS_GM * tiles_n, grp = bid / per, first = grp * PS_GM;
        const int gsz = tiles_m - first < PS_GM ? tiles_m - first : PS_GM;
        const int in = bid - grp * per;
        th_ = first + in % gsz;
        tn_ = in / gsz;
    }
    const int tm_ = th_ >> 1;                       // the sample
    const bool hasE = (th_ & 1) == 0;               // first half: rows 0..143 (128 + the 16-row block); second: 144..271
    const int pos0 = hasE ? 0 : PH_AROWS, tile_rows = hasE ? PH_AROWS : 128;
    const int row_lo = tm_ * L + pos0;              // a multiple of 16
    const int m0 = row_lo, n0 = tn_ * BN;
    const int nk = p.K / 32;
    const _Float16* Ap = (const _Float16*)p.A;
    const _Float16* Wp = (const _Float16*)p.W;
    unsigned long long srcA[2], srcB[2][2], srcE;
    int offA[2], offB[2][2], offE;
    const unsigned lane16 = lane * 16;
    const int rgsA = (p.M + 15) >> 4, rgsB = (p.N + 15) >> 4;
#define PH_BASE(dst_, ptr_)                                                                          \
    do {                                                                                             \
        const unsigned long long a_ = (unsigned long long)(ptr_);                                    \
        dst_ = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a_ >> 32)) << 32) | \
               (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a_);           \
    } while (0)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int idx = 2 * wave + k, plane = idx >> 3, r = idx & 7;
        int rg = (m0 >> 4) + r;                                        // the 8 row groups of the tile's 128 main rows
        if (rg >= rgsA) rg = rgsA - 1;
        PH_BASE(srcA[k], Ap + plane * p.a_plane + (size_t)rg * nk * 512);
        offA[k] = __builtin_amdgcn_readfirstlane((plane * 9 + r) * 1024);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int gip = (r >> 1) * 4 + s * 2 + (r & 1);            // wave column r >> 1, B-sub s, 16-row group r & 1
            int rb = (n0 >> 4) + gip;
            if (rb >= rgsB) rb = rgsB - 1;
            PH_BASE(srcB[s][k], Wp + plane * p.w3_plane + (size_t)rb * nk * 512);
            offB[s][k] = __builtin_amdgcn_readfirstlane((18 + plane * 16 + gip) * 1024);
        }
    }
    {   // the 16 extra A rows (group 8 of both planes): wave w loads plane w & 1; a tile without them re-reads group 7
        const int plane = wave & 1;
        int rg = (m0 >> 4) + (hasE ? 8 : 7);
        if (rg >= rgsA) rg = rgsA - 1;
        PH_BASE(srcE, Ap + plane * p.a_plane + (size_t)rg * nk * 512);
        offE = __builtin_amdgcn_readfirstlane((plane * 9 + 8) * 1024);
    }
#define PH_DMA(src_, off_, tile_, st_)                                                               \
    __builtin_amdgcn_global_load_lds((ds_gptr)((const unsigned char*)((src_) + (unsigned long long)(tile_) * 1024) + lane16), \
                                     (ds_lptr)(smem_raw + (st_) * (STAGE * 2) + (off_)), 16, 0, 0)
#define PH_ISSUE(tile_, half_, st_)                                                                  \
    do {                                                                                             \
        if ((half_) == 0) {                                                                          \
            PH_DMA(srcA[0], offA[0], tile_, st_); PH_DMA(srcA[1], offA[1], tile_, st_);              \
            PH_DMA(srcB[0][0], offB[0][0], tile_, st_); PH_DMA(srcB[0][1], offB[0][1], tile_, st_);  \
        } else {                                                                                     \
            PH_DMA(srcB[1][0], offB[1][0], tile_, st_); PH_DMA(srcB[1][1], offB[1][1], tile_, st_);  \
            PH_DMA(srcE, offE, tile_, st_);                                                          \
        }                                                                                            \
    } while (0)
    const int swz[2] = {((0 + hh) ^ ((l31 >> 2) & 3)) * 8, ((2 + hh) ^ ((l31 >> 2) & 3)) * 8};
    const int l15 = PS_SIG((lane >> 2) & 3) * 4 + (lane & 3);
    const int swzq = ((lane >> 4) ^ ((l15 >> 2) & 3)) * 8;
    f32x16 acc[2][2], acc8;             // acc8: the 32-row ninth block of the full tile (named by the shared epilogue, unused)
    f32x4 acc9[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) acc9[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc8[r] = 0.f;
    h8 a0[2][2], a1[2][2];              // [ks][row block]: hi, lo planes
    h8 b0[2][2], b1[2][2];              // [B-sub][ks]
    h8 ea0, ea1, eb0[2], eb1[2];        // the 16 extra A rows and the wave's two 16-column B tiles, 16x16x32 operand layout
#define PH_READ_A(st_)                                                                               \
    do {                                                                                             \
        const _Float16* Ac = smem + (st_) * STAGE + (wr * 64 + l31) * HLD;                           \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                             \
            _Pragma("unroll") for (int ib = 0; ib < 2; ++ib) {                                       \
                a0[ks][ib] = *(const h8*)(Ac + ib * 32 * HLD + swz[ks]);                             \
                a1[ks][ib] = *(const h8*)(Ac + APL + ib * 32 * HLD + swz[ks]);                       \
            }                                                                                        \
    } while (0)
#define PH_READ_B(st_, s_)                                                                           \
    do {                                                                                             \
        const _Float16* Bc = smem + (st_) * STAGE + 2 * APL + (wc * 64 + (s_) * 32 + l31) * HLD;     \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                           \
            b0[s_][ks] = *(const h8*)(Bc + swz[ks]);                                                 \
            b1[s_][ks] = *(const h8*)(Bc + BPL + swz[ks]);                                           \
        }                                                                                            \
    } while (0)
#define PH_READ_EB16(st_)                                                                            \
    do {                                                                                             \
        const _Float16* Bc = smem + (st_) * STAGE + 2 * APL + ((2 * wc + wr) * 32 + l15) * HLD;      \
        _Pragma("unroll") for (int tt = 0; tt < 2; ++tt) {                                           \
            eb0[tt] = *(const h8*)(Bc + tt * 16 * HLD + swzq);                                       \
            eb1[tt] = *(const h8*)(Bc + BPL + tt * 16 * HLD + swzq);                                 \
        }                                                                                            \
    } while (0)
#define PH_READ_EA16(st_)                                                                            \
    do {                                                                                             \
        const _Float16* Ec = smem + (st_) * STAGE + (128 + l15) * HLD;                               \
        ea0 = *(const h8*)(Ec + swzq);                                                               \
        ea1 = *(const h8*)(Ec + APL + swzq);                                                         \
    } while (0)
    // per accumulator ks 0 {a1 b0, a0 b1, a0 b0}, ks 1 {...}: the order of every f16x2 program
#define PH_QUAD(sb_)                                                                                 \
    do {                                                                                             \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                           \
            _Pragma("unroll") for (int ib = 0; ib < 2; ++ib)                                         \
                acc[ib][sb_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[ks][ib], b0[sb_][ks], acc[ib][sb_], 0, 0, 0); \
            _Pragma("unroll") for (int ib = 0; ib < 2; ++ib)                                         \
                acc[ib][sb_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[ks][ib], b1[sb_][ks], acc[ib][sb_], 0, 0, 0); \
            _Pragma("unroll") for (int ib = 0; ib < 2; ++ib)                                         \
                acc[ib][sb_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[ks][ib], b0[sb_][ks], acc[ib][sb_], 0, 0, 0); \
        }                                                                                            \
    } while (0)
#define PH_EXTRA16()                                                                                 \
    do {                                                                                             \
        _Pragma("unroll") for (int tt = 0; tt < 2; ++tt) {                                           \
            acc9[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ea1, eb0[tt], acc9[tt], 0, 0, 0);      \
            acc9[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ea0, eb1[tt], acc9[tt], 0, 0, 0);      \
            acc9[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ea0, eb0[tt], acc9[tt], 0, 0, 0);      \
        }                                                                                            \
    } while (0)
    // one phase of k-tile `tt` (run time) in stage ST = tt % 3 (compile time): P = 0: (A, B-sub0), P = 1: (A, B-sub1) + block
#define PH_PHASE(P, ST)                                                                              \
    do {                                                                                             \
        if (P == 0) { PH_READ_A(ST); PH_READ_B(ST, 0); if (hasE && wr == 0) PH_READ_EB16(ST); }      \
        else { PH_READ_B(ST, 1); if (hasE) { PH_READ_EA16(ST); if (wr == 1) PH_READ_EB16(ST); } }    \
        PS_FENCE();                                                                                  \
        if (tt + 2 < nk) {                                                                           \
            PH_ISSUE_LOOP(P, ((ST) + 2) % 3);                                                        \
            if (P == 0) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");   /* halves g + 2 .. g + 4: 4 + 3 + 4 */ \
            else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");          /*                        3 + 4 + 3 */ \
        } else {                                                                                     \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   /* tail: nothing younger to count */  \
        }                                                                                            \
        PS_BAR();                                                                                    \
        __builtin_amdgcn_s_setprio(1);                                                               \
        if (P == 0) PH_QUAD(0);                                                                      \
        else { PH_QUAD(1); if (hasE) PH_EXTRA16(); }                                                 \
        __builtin_amdgcn_s_setprio(0);                                                               \
        PS_BAR();                                                                                    \
    } while (0)
    // the transfers inside the loop, written out as in the full-tile kernel (PS_DMA_ASM): running bases, first used for k-tile 2
    unsigned long long curA[2] = {srcA[0] + 2048, srcA[1] + 2048}, curE = srcE + 2048;
    unsigned long long curB[2][2] = {{srcB[0][0] + 2048, srcB[0][1] + 2048}, {srcB[1][0] + 2048, srcB[1][1] + 2048}};
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw);
    const unsigned absA[2] = {lds0 + offA[0], lds0 + offA[1]}, absE = lds0 + offE;
    const unsigned absB[2][2] = {{lds0 + offB[0][0], lds0 + offB[0][1]}, {lds0 + offB[1][0], lds0 + offB[1][1]}};
#define PH_ISSUE_LOOP(half_, st_)                                                                    \
    do {                                                                                             \
        if ((half_) == 0) {                                                                          \
            PS_DMA_ASM(curA[0], absA[0], (st_) * (STAGE * 2)); PS_DMA_ASM(curA[1], absA[1], (st_) * (STAGE * 2)); \
            PS_DMA_ASM(curB[0][0], absB[0][0], (st_) * (STAGE * 2)); PS_DMA_ASM(curB[0][1], absB[0][1], (st_) * (STAGE * 2)); \
        } else {                                                                                     \
            PS_DMA_ASM(curB[1][0], absB[1][0], (st_) * (STAGE * 2)); PS_DMA_ASM(curB[1][1], absB[1][1], (st_) * (STAGE * 2)); \
            PS_DMA_ASM(curE, absE, (st_) * (STAGE * 2));                                             \
        }                                                                                            \
    } while (0)
    // prologue: k-tiles 0 and 1 (halves 0 .. 3 = 14 instructions per wave); half 0 has landed once only halves 1 .. 3
    // (3 + 4 + 3) are outstanding
    PH_ISSUE(0, 0, 0); PH_ISSUE(0, 1, 0);
    if (nk > 1) { PH_ISSUE(1, 0, 1); PH_ISSUE(1, 1, 1); asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PS_BAR();
    if (wr == 1) PS_BAR();          // the second wave row runs one barrier behind the first
    for (int t = 0; t < nk; t += 3) {
        { const int tt = t; PH_PHASE(0, 0); PH_PHASE(1, 0); }
        if (t + 1 < nk) { const int tt = t + 1; PH_PHASE(0, 1); PH_PHASE(1, 1); }
        if (t + 2 < nk) { const int tt = t + 2; PH_PHASE(0, 2); PH_PHASE(1, 2); }
    }
    if (wr == 0) PS_BAR();          // ... and the first row waits for it at the end
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    constexpr int WROW = 64, EROW0 = 128;          // tile geometry of a half tile (gemm_f16x2_ps_epilogue.inc)
    constexpr bool HALF_TILE = true;
#include "gemm_f16x2_ps_epilogue.inc"
}

// Whether the per-sample program serves this problem: packed operands, sample-structured rows with 256 < L + 15 <= 288, N
// in whole 256-column tiles, an even number of k-tiles, 16-byte-aligned row stores.  half: the half-tile program (272-row
// samples only).  Whether it PAYS is ds_gemm_f16x2_ps_choice below.
static bool ps_serves(const GemmParams& p, bool half) {
    const int L = p.rows_per_sample;
    if (!p.a_split || L <= 0 || L > PS_BM - 15 || L <= 240 || p.M % L != 0) return false;
    if (half && L != PS_BM - 16) return false;
    if (p.N % PS_BN != 0 || p.K % 64 != 0 || p.lda != p.K || p.ldw != p.K) return false;
    if (p.store == DS_STORE_ROW && !p.c_split) {
        if (((p.N | p.ldc | p.ldr) & 3) != 0 || (((uintptr_t)p.C | (uintptr_t)p.R) & 15) != 0) return false;
    } else if (p.store == DS_STORE_ROW) {
        if (p.R) return false;
    } else if (p.store == DS_STORE_ATTN) {
        if (p.row_off != 0 || p.attn_heads * 64 % PS_BN != 0) return false;
    } else {
        return false;
    }
    return true;
}

// THE grid rule of the per-sample programs (one copy: ds_launch_gemm_f16x2 dispatches on it, and the denoiser driver's
// padded-row mode -- api.hip rows_per_sample -- asks ds_gemm_f16x2_ps_taken, which also knows the forced tile).
// B samples x N columns -> 0: neither pays, 1: full tiles (B N / 256 of them), 2: half tiles (twice as many; 272-row
// samples only).  A program pays by the share of the CU-rounds it occupies that do work; half tiles move 1.46x the operand
// bytes per MFMA, hence the 0.92.  The floor: below ~0.65 the 4-wave programs (0.29-0.30 of the pipe against 0.46) catch up.
int ds_gemm_f16x2_ps_choice(long B, long N, bool half_ok) {
    const long tf = B * (N / PS_BN), th = 2 * tf;
    const double ef = (double)tf / (double)(((tf + 255) / 256) * 256);
    const double eh = half_ok ? 0.92 * (double)th / (double)(((th + 255) / 256) * 256) : 0.0;
    if (ef < 0.65 && eh < 0.65) return 0;
    return ef >= eh ? 1 : 2;
}

bool ds_gemm_f16x2_ps_applies(const GemmParams& p, bool need_full_grid) {
    if (!ps_serves(p, false)) return false;
    return !need_full_grid || ds_gemm_f16x2_ps_choice(p.M / p.rows_per_sample, p.N, false) == 1;
}
bool ds_gemm_f16x2_ph_applies(const GemmParams& p) { return ps_serves(p, true); }
// the program ds_launch_gemm_f16x2 should take for p when nothing is forced: 0 none, 1 full tiles, 2 half tiles
int ds_gemm_f16x2_ps_pick(const GemmParams& p) {
    if (!ps_serves(p, false)) return 0;
    return ds_gemm_f16x2_ps_choice(p.M / p.rows_per_sample, p.N, ps_serves(p, true));
}

template <int EPI, bool NB16>
static int launch_ps(const GemmParams& p, hipStream_t s) {
    static DsOnce attr_set;
    if (attr_set.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)ds_gemm_f16x2_ps_kernel<EPI, NB16>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, PS_LDS_BYTES);
        if (e != hipSuccess) {
            ds_set_error("gemm_f16x2_ps: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return -2;
        }
        attr_set.done();
    }
    const int tiles = (p.M / p.rows_per_sample) * (p.N / PS_BN);
    hipLaunchKernelGGL((ds_gemm_f16x2_ps_kernel<EPI, NB16>), dim3(tiles), dim3(512), PS_LDS_BYTES, s, p);
    DS_CHECK_LAUNCH();
    return 0;
}

int ds_launch_gemm_f16x2_ps(const GemmParams& p, hipStream_t s) {
    // 272-row samples (16-row aligned, 17 packed groups): the ninth block row is 16 rows on the 16x16x32 MFMA
    const bool nb16 = p.rows_per_sample == PS_BM - 16;
    if (p.store == DS_STORE_ATTN) return nb16 ? launch_ps<PS_EPI_ATTN, true>(p, s) : launch_ps<PS_EPI_ATTN, false>(p, s);
    if (p.c_split) return nb16 ? launch_ps<PS_EPI_SPLIT, true>(p, s) : launch_ps<PS_EPI_SPLIT, false>(p, s);
    return nb16 ? launch_ps<PS_EPI_ROW, true>(p, s) : launch_ps<PS_EPI_ROW, false>(p, s);
}

template <int EPI>
static int launch_ph(const GemmParams& p, hipStream_t s) {
    static DsOnce attr_set;
    if (attr_set.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)ds_gemm_f16x2_ph_kernel<EPI>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, PH_LDS_BYTES);
        if (e != hipSuccess) {
            ds_set_error("gemm_f16x2_ph: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return -2;
        }
        attr_set.done();
    }
    const int tiles = 2 * (p.M / p.rows_per_sample) * (p.N / PS_BN);
    hipLaunchKernelGGL((ds_gemm_f16x2_ph_kernel<EPI>), dim3(tiles), dim3(512), PH_LDS_BYTES, s, p);
    DS_CHECK_LAUNCH();
    return 0;
}

int ds_launch_gemm_f16x2_ph(const GemmParams& p, hipStream_t s) {
    if (p.store == DS_STORE_ATTN) return launch_ph<PS_EPI_ATTN>(p, s);
    if (p.c_split) return launch_ph<PS_EPI_SPLIT>(p, s);
    return launch_ph<PS_EPI_ROW>(p, s);
}
