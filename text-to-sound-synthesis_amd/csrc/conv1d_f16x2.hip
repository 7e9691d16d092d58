// Dilated k = 3 Conv1d with ReflectionPad1d(dil) over a channels-last fp32 signal on the fp16 matrix cores (2-way fp16 split,
// 3 MFMA passes, fp32-class): the first conv of MelGAN's 128- and 256-channel ResnetBlocks (vocoder/modules.py:72-85), LeakyReLU
// in front.  The 1-D sibling of conv3x3_f16x2.hip, same machinery: a workgroup owns 128 consecutive time positions x 128 output
// channels and walks the input channels in slabs of 32; the slab's HALO (128 + 2 dil rows, reflected at the clip's ends) is
// loaded, activated and split ONCE into LDS (80-byte rows: conflict-free ds_read_b128 with plain base + immediate addresses) and
// the three taps read their A fragments from it at row offsets 0 / dil / 2 dil.  The gather kernel (conv_f16x2.hip) it replaces
// for these layers loads, activates and splits every element once per tap and per 128-channel tile and is bound by that vector
// work (0.26 of the 3-pass ceiling).  A wave's tile is all four block rows x 32 output channels; its weight fragments (packed
// fragment-major on the host, _lib.pack_conv_weights(taps = 3)) come straight from L2 into a register ring of 6 k-steps = one
// slab, loaded by asm 5 k-steps ahead with explicit vmcnt waits (hipcc sinks visible prefetch loads to their use and waits for
// them with vmcnt(0)); the next slab's halo is held in registers for a whole slab before it is written.  One barrier per slab.
// The same kernel with two taps is MelGAN's stride-8 ConvTranspose1d (vocoder/modules.py:104-113) in polyphase form: phase p of
// out[(q + e_p) r + p - pad] = W[:, :, p] a[q + e_p] + W[:, :, p + r] a[q + e_p - 1]  (e_p = p < pad; rows outside the clip are
// zero) is a 2-tap conv over the staged rows q0 - 1 .. q0 + 128 with its own weights and an interleaving store; a workgroup
// computes ONE phase of a 128-position tile, the r phases of a tile are neighbours in the grid (the tile comes out of L2).
#include "common.h"
#include <type_traits>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

#define C1_TT 128                     // time positions per tile
#define C1_BN 128                     // output channels per tile
#define C1_PXP 40                     // halves per staged row: 32 channels + 8
#define C1_MAXDIL 27
#define C1_NF4 6                      // float4 work items per thread cover (128 + 54) rows x 8
#define C1_WSTEP 8192                 // halves of packed weights per (n-tile, slab, tap): [plane 2][wave 4][ks 2][lane 64][8]

struct Conv1Params {
    const float* x;        // [B][T][Cin]
    const _Float16* w;     // fragment-packed planes of W * 2^s: [phases][Cout/128][Cin/32][taps][C1_WSTEP]
    const float* bias;     // [Cout] or null
    float* y;              // [B][T][Cout]  (transposed conv: [B][T r][Cout])
    float out_scale;
    int B, T, Cin, Cout, dil, tiles_t, lrelu;
    int ct_r, ct_pad;      // transposed conv: stride = number of phases, padding
};

// TAPS = 3, CT = false: the dilated conv;  TAPS = 2, CT = true: one phase of the transposed conv
template <int TAPS, bool CT>
__global__ __launch_bounds__(256, 2) void ds_conv1d_f16x2_kernel(const Conv1Params p) {
    constexpr int NK = TAPS * 2;                           // k-steps per 32-channel slab = slots of the weight ring
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int R = CT ? C1_TT + 2 : C1_TT + 2 * p.dil;      // staged rows
    const int HPL = R * C1_PXP;                            // halves per plane
    _Float16* halo = (_Float16*)smem_raw;                  // [2 buffers][2 planes][R][40]
    const int tiles_n = p.Cout / C1_BN, nblk = gridDim.x;
    int bid = blockIdx.x;
    {   // each XCD works a contiguous run of tiles (the n-tiles of a position tile share its halo through that XCD's L2)
        const int xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int nt = bid % tiles_n;
    const int ph = CT ? (bid / tiles_n) % p.ct_r : 0, pt = CT ? bid / (tiles_n * p.ct_r) : bid / tiles_n;
    const int e_p = CT && ph < p.ct_pad ? 1 : 0;           // transposed conv: source index s0 = q + e_p, taps read s0 and s0 - 1
    const int b = pt / p.tiles_t, t0 = (pt - b * p.tiles_t) * C1_TT;
    const int n0 = nt * C1_BN;
    const float* xb = p.x + (size_t)b * p.T * p.Cin;

    // ---- halo staging: work item f = tid + 256 u -> row f >> 3, float4 (4 channels) f & 7 of the 32-channel slab ----
    f32x4 hv[C1_NF4];
    // (hv goes in as a parameter and the asm operands are its elements themselves: a copy "f32x4 v = hv[u]" in front of the
    //  wait lets hipcc emit a v_mov of the register BEFORE the s_waitcnt -- it read the previous contents; found by the tests)
    auto halo_load = [&](f32x4 (&hvr)[C1_NF4], int slab) {
#pragma unroll
        for (int u = 0; u < C1_NF4; ++u) {
            const int f = tid + 256 * u, r = f >> 3, c4 = f & 7;
            int t = t0 - (CT ? 1 : p.dil) + (r < R ? r : 0);
            if (!CT) {
                if (t < 0) t = -t;
                if (t >= p.T) t = 2 * (p.T - 1) - t;
            }
            if (t < 0 || t >= p.T) t = 0;                   // (transposed conv: zeroed below; conv: rows of a ragged last tile, unused)
            const unsigned off = (unsigned)(t * p.Cin + slab * 32 + c4 * 4) * 4u;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(hvr[u]) : "v"(off), "s"(xb) : "memory");
        }
    };
    auto halo_write = [&](f32x4 (&hvr)[C1_NF4], int slab, auto wait_) {
        constexpr int WAIT = decltype(wait_)::value;
        _Float16* hb = halo + (slab & 1) * (2 * HPL);
#pragma unroll
        for (int u = 0; u < C1_NF4; ++u) {
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(hvr[u]) : "n"(WAIT));
            const f32x4 v = hvr[u];
            const int f = tid + 256 * u, r = f >> 3, c4 = f & 7;
            if (u >= C1_TT / 32 && r >= R) continue;        // (items 0 .. 3 are rows 0 .. 127 < R)
            const int ts = t0 - 1 + r;
            const bool zero = CT && (ts < 0 || ts >= p.T);   // outside the clip: no contribution
            h4 s0, s1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = p.lrelu ? (v[e] > 0.f ? v[e] : 0.2f * v[e]) : v[e];
                if (zero) a = 0.f;
                s0[e] = ds_split_hi(a);
                s1[e] = ds_split_lo(a, s0[e]);
            }
            *(h4*)(hb + r * C1_PXP + c4 * 4) = s0;
            *(h4*)(hb + HPL + r * C1_PXP + c4 * 4) = s1;
        }
    };

    // ---- weight ring: k-step q = (slab * TAPS + tap) * 2 + ks, slot q % NK ----
    const int nslab = p.Cin >> 5, nq = nslab * NK;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const char* wbase = (const char*)(p.w + (size_t)(ph * tiles_n + nt) * nslab * TAPS * C1_WSTEP + wave_u * 1024);
    const unsigned voff0 = lane * 16, voff1 = lane * 16 + 8192;
    h8 bq[NK][2];
    auto w_load = [&](int q, h8 (&f)[2]) {
        const char* r = wbase + (size_t)(q >> 1) * (C1_WSTEP * 2) + (q & 1) * 1024;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(f[0]) : "v"(voff0), "s"(r) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(f[1]) : "v"(voff1), "s"(r) : "memory");
    };
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    halo_load(hv, 0);
#pragma unroll
    for (int q = 0; q < NK - 1; ++q) w_load(q, bq[q]);
    halo_write(hv, 0, std::integral_constant<int, 2 * (NK - 1)>{});
    halo_load(hv, nslab > 1 ? 1 : 0);
    __syncthreads();
    for (int slab = 0; slab < nslab; ++slab) {
        const _Float16* ab = halo + (slab & 1) * (2 * HPL) + l31 * C1_PXP + hh * 8;      // lane's row inside a block and k half
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const _Float16* at = ab + (CT ? 1 + e_p - tap : tap * p.dil) * C1_PXP;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int ql = tap * 2 + ks;
                const int qn = slab * NK + ql + NK - 1;
                w_load(qn < nq ? qn : nq - 1, bq[(ql + NK - 1) % NK]);
                h8 fa0[4], fa1[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const _Float16* ar = at + i * 32 * C1_PXP + ks * 16;
                    fa0[i] = *(const h8*)ar;
                    fa1[i] = *(const h8*)(ar + HPL);
                }
                asm volatile("s_waitcnt vmcnt(%2)" : "+v"(bq[ql][0]), "+v"(bq[ql][1]) : "n"(2 * (NK - 1)));
                const h8 b0 = bq[ql][0], b1 = bq[ql][1];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x16 c = acc[i];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa1[i], b0, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0[i], b1, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0[i], b0, c, 0, 0, 0);
                    acc[i] = c;
                }
            }
            if (tap == 0) {
                // Straight-line on purpose (no branch on the slab index: with control flow around it hipcc carried the halo
                // registers over the loop edge through v_mov copies, which read them while their loads were in flight).
                // The other buffer is free (every wave passed the barrier that ended the slab before this one); its data was
                // requested a slab ago -- at least this tap's 4 weight loads since then (slab 1's, requested in the prologue:
                // exactly 4).  Past the last slab: a harmless rewrite / reload of the last slab.
                halo_write(hv, slab + 1, std::integral_constant<int, 4>{});
                halo_load(hv, slab + 2 < nslab ? slab + 2 : nslab - 1);
            }
        }
        __syncthreads();
    }

    // The asm loads still in flight (the ring's clamped refills of the last k-steps, the redundant last halo request) must land
    // BEFORE anything reuses their registers: hipcc does not know they are outstanding.  Ring and halo registers go INTO the
    // drain as operands: a load whose result is never read is dead to the compiler, which gave all of them ONE throw-away
    // register and reused it for an A fragment while they were in flight (seen in the ISA of the first version).
#pragma unroll
    for (int q = 0; q < NK; ++q) asm volatile("s_waitcnt vmcnt(0)" : "+v"(bq[q][0]), "+v"(bq[q][1])::"memory");
#pragma unroll
    for (int u = 0; u < C1_NF4; ++u) asm volatile("s_waitcnt vmcnt(0)" : "+v"(hv[u])::"memory");

    // ---- epilogue: two passes of 64 rows (block rows 2 pass, 2 pass + 1 of every wave) staged as fp32 [64][128] in LDS ----
    const float osc = p.out_scale;
    float* Tf = (float*)smem_raw;
    const int cc = tid & 31, col = n0 + cc * 4;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias4 = *(const f32x4*)(p.bias + col);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int cl = wave * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                Tf[rl * C1_BN + cl] = acc[2 * pass + i][r] * osc;
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int rl = (tid >> 5) + 8 * it, t = t0 + 64 * pass + rl;
            if (t < p.T) {
                const f32x4 val = *(const f32x4*)(Tf + rl * C1_BN + cc * 4) + bias4;
                // transposed conv: position q = t of phase ph is output row (q + e_p) r + ph - pad (always inside [0, T r))
                const long long row = CT ? (long long)(t + e_p) * p.ct_r + ph - p.ct_pad : t;
                const long long rows = CT ? (long long)p.T * p.ct_r : p.T;
                if (row < rows) *(f32x4*)(p.y + ((size_t)b * rows + row) * p.Cout + col) = val;
            }
        }
        __syncthreads();
    }
}

template <int TAPS, bool CT>
static int c1_launch(Conv1Params& p, hipStream_t s) {
    p.tiles_t = (p.T + C1_TT - 1) / C1_TT;
    const int R = CT ? C1_TT + 2 : C1_TT + 2 * p.dil;
    size_t lds = (size_t)2 * 2 * R * C1_PXP * 2;
    if (lds < 64 * C1_BN * 4) lds = 64 * C1_BN * 4;          // the staged output half tile
    static DsOnce attr_set;
    if (attr_set.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)ds_conv1d_f16x2_kernel<TAPS, CT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           2 * 2 * (C1_TT + 2 * C1_MAXDIL) * C1_PXP * 2);
        if (e != hipSuccess) {
            ds_set_error("conv1d_f16x2: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return -2;
        }
        attr_set.done();
    }
    const long long blocks = (long long)p.B * p.tiles_t * (p.Cout / C1_BN) * (CT ? p.ct_r : 1);
    DS_CHECK_ARG(blocks < (1ll << 31), "too many tiles");
    hipLaunchKernelGGL((ds_conv1d_f16x2_kernel<TAPS, CT>), dim3((unsigned)blocks), dim3(256), lds, s, p);
    DS_CHECK_LAUNCH();
    return 0;
}

// y[b][t][n] = bias[n] + 2^-s sum_{j < 3} sum_c W2[n][j][c] act(x[b][reflect(t + (j - 1) dil)][c]),  act = LeakyReLU(0.2) if lrelu.
// x, y channels-last fp32; w2 = _lib.pack_conv_weights(split_f16x2(W [Cout][3 Cin], K ordered [tap][channel]), Cout, Cin, 3).
extern "C" int ds_conv1d_k3_f16x2(const float* x, const void* w2, long long w_halves, float out_scale, const float* bias, float* y,
                                  int B, int T, int Cin, int Cout, int dil, int lrelu, ds_stream_t stream) {
    DS_CHECK_ARG(x && w2 && y, "null pointer");
    DS_CHECK_ARG(B > 0 && T > 0 && Cin > 0 && Cin % 32 == 0 && Cout > 0 && Cout % C1_BN == 0, "Cin % 32 == 0 and Cout % 128 == 0");
    DS_CHECK_ARG(dil > 0 && dil <= C1_MAXDIL && dil < T, "0 < dil <= 27, dil < T");
    DS_CHECK_ARG(w_halves == (long long)2 * Cout * 3 * Cin && out_scale > 0.f, "packed weights: 2 * Cout * 3 * Cin halves");
    DS_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)w2 & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)bias & 15) == 0,
                 "operands must be 16-byte aligned");
    DS_CHECK_ARG((long long)T * Cin < (1ll << 30), "32-bit byte offsets inside a clip");
    Conv1Params p;
    p.x = x; p.w = (const _Float16*)w2; p.bias = bias; p.y = y; p.out_scale = out_scale;
    p.B = B; p.T = T; p.Cin = Cin; p.Cout = Cout; p.dil = dil; p.lrelu = lrelu ? 1 : 0; p.ct_r = 1; p.ct_pad = 0;
    return c1_launch<3, false>(p, (hipStream_t)stream);
}

// ConvTranspose1d(k = 2 r, stride r, padding pad) in polyphase form:  y[b][(q + e_p) r + p - pad][n] = bias[n] + 2^-s sum_c
// (W[p][n][0][c] act(x[b][q + e_p][c]) + W[p][n][1][c] act(x[b][q + e_p - 1][c])),  e_p = p < pad, source rows outside [0, T) zero.
// x [B][T][Cin], y [B][T r][Cout] channels-last fp32; w2 = the r phases' weights [Cout][2 Cin] (K ordered [tap][channel]: tap 0 =
// W[:, :, p], tap 1 = W[:, :, p + r]) each split and fragment-packed with _lib.pack_conv_weights(.., taps = 2), concatenated.
extern "C" int ds_convt1d_f16x2(const float* x, const void* w2, long long w_halves, float out_scale, const float* bias, float* y,
                                int B, int T, int Cin, int Cout, int r, int pad, int lrelu, ds_stream_t stream) {
    DS_CHECK_ARG(x && w2 && y, "null pointer");
    DS_CHECK_ARG(B > 0 && T > 1 && Cin > 0 && Cin % 32 == 0 && Cout > 0 && Cout % C1_BN == 0, "Cin % 32 == 0 and Cout % 128 == 0");
    DS_CHECK_ARG(r >= 1 && pad >= 0 && pad < r, "0 <= pad < r");
    DS_CHECK_ARG(w_halves == (long long)r * 2 * Cout * 2 * Cin && out_scale > 0.f, "packed weights: r * 2 * Cout * 2 * Cin halves");
    DS_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)w2 & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)bias & 15) == 0,
                 "operands must be 16-byte aligned");
    DS_CHECK_ARG((long long)T * Cin < (1ll << 30), "32-bit byte offsets inside a clip");
    Conv1Params p;
    p.x = x; p.w = (const _Float16*)w2; p.bias = bias; p.y = y; p.out_scale = out_scale;
    p.B = B; p.T = T; p.Cin = Cin; p.Cout = Cout; p.dil = 1; p.lrelu = lrelu ? 1 : 0; p.ct_r = r; p.ct_pad = pad;
    return c1_launch<2, true>(p, (hipStream_t)stream);
}
