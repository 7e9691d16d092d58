// 3x3 convolution (stride 1, zero padding 1) over a channels-last fp32 image on the fp16 matrix cores, 2-way fp16 split
// (3 MFMA passes, fp32-class): the hot convs of the SpecVQGAN decoder -- ResnetBlock conv1 / conv2 with their
// GroupNorm + swish in front (specvqgan/modules/diffusionmodules/model.py:92-151) and the Upsample conv (:37-52) -- at the
// resolutions that carry 94 % of the decoder's flops (20x212 and up).  Round 4: HALO-TILED form of conv_f16x2.hip's conv2d
// loader.  That kernel gathers the A operand tap by tap: every input element is loaded, normalised, swish-ed (v_exp + v_rcp)
// and split into fp16 planes NINE times per output-channel tile, and the kernel was bound by that vector work (~1500 VALU
// cycles against 768 MFMA cycles per k-tile: 214 TF-eq, 0.26 of the 3-pass ceiling).  Here a workgroup owns a 4 x 32 pixel
// output tile (M = 128) x 128 output channels and walks the input channels in slabs of 32: the slab's 6 x 34 pixel HALO is
// loaded, activated and split ONCE into LDS and all nine taps read their A fragments from it at shifted pixel offsets.
// The weights do not pass through LDS at all: they are packed once (host, _lib.pack_conv3x3_weights) so that every MFMA
// B fragment of a (n-tile, slab, tap) is one contiguous KB, and each wave loads the fragments of ITS 32 output channels
// straight from L2 into a register ring, C3_RING - 1 = 5 k-steps (60 MFMAs) ahead -- so the ONLY barrier is the halo swap, once
// per slab = per 216 MFMAs of a wave.  A wave's tile is all four tile rows x 32 output channels (no two waves load the same
// weight bytes; the A fragments, four block rows per k-step, come from LDS with immediate offsets off ONE base register).
// Round 4, second form: the wave tile had been 2 x 2 blocks with the weights one tap ahead and the GroupNorm affine loaded from
// global memory inside every halo work item -- the ISA showed s_waitcnt vmcnt(0) there, seven exposed L2 round trips per
// slab that also drained the weight prefetch (45 % of the wave time in s_waitcnt, profiles/r04m_*): 305-344 -> 335-372 TF-eq.
// Measured null on top of this form (profiles/r04ze_*): a ring of 9 k-steps, A fragments read one k-step ahead into a second
// register set -- the kernel now sits at 0.40-0.45 of the nominal ceiling where the dominant GEMM, whose matrix pipe is 98 %
// busy, reaches 0.485 under the same board power cap.
// Vector work per output element drops ~6x against the gather kernel, activation traffic out of L2 ~5x.
//   LDS: two halo slabs [2 planes][204 px][32 ch + 8] (2 x 32 KB; 80-byte pixel rows: the 16 consecutive pixels a
//        ds_read_b128 service group touches land in 16 different 16-byte slots of the 256-byte bank row at every tap offset,
//        with no swizzle in the address), then the sample's GroupNorm [scale | shift]; slab s + 1 is loaded into registers
//        and written into the other buffer in two halves under taps 0 - 5.
//   Epilogue: bias, optional residual, row-major fp32 store through LDS (16-byte accesses), and optionally the GroupNorm
//        partial sums of the OUTPUT per (tile, channel) in double -- the statistics pass of the next GroupNorm
//        (ds_gn_partial_kernel: one more read of the tensor) disappears; ds_groupnorm_finish turns them into the affine.
#include "common.h"
#include <type_traits>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define C3_TH 4                       // tile: 4 rows x 32 pixels
#define C3_TW 32
#define C3_HW (C3_TW + 2)             // halo pitch (pixels)
#define C3_HPX ((C3_TH + 2) * C3_HW)  // 204 halo pixels
#define C3_PXP 40                     // halves per halo pixel in LDS: 32 channels + 8 (80-byte rows: see "LDS" above)
#define C3_HPL (C3_HPX * C3_PXP)      // halves per halo plane
#define C3_BN 128
#define C3_WSTEP 8192                 // halves of packed weights per (n-tile, slab, tap): [plane 2][column block 4 = wave][ks 2][lane 64][8]
#define C3_NF4 ((C3_HPX * 8 + 255) / 256)   // float4 work items of a halo slab per thread (7)
#ifndef C3_RING                       // k-steps of weight fragments in flight per wave (divides 18)
#define C3_RING 6
#endif
#define C3_HALO_BYTES (2 * 2 * C3_HPL * 2)
#define C3_MAXCIN 1024               // (the LDS reservation of the prologue affine)

struct Conv3Params {
    const float* x;        // [B][Hs][Ws][Cin]
    const _Float16* w;     // W * 2^s as fp16 planes in the fragment-packed layout [Cout/128][Cin/32][9][C3_WSTEP]
    const float* bias;     // [Cout] or null
    const float* R;        // residual [B][H][W][Cout] or null
    float* y;              // [B][H][W][Cout]
    const float* pro_scale;  // [B][Cin] GroupNorm folded to a * s + o (PRO) or null
    const float* pro_shift;
    double* gn_part;       // [B][tiles][2][Cout] partial sums of the output (STATS) or null
    float out_scale;
    int B, H, W, Cin, Cout, tiles_x, tiles_y;
};

template <int PRO, int UP, bool STATS>
__global__ __launch_bounds__(256, 2) void ds_conv3x3_f16x2_kernel(const Conv3Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    _Float16* halo = (_Float16*)smem_raw;                  // [2 buffers][2 planes][204][32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int tiles_n = p.Cout / C3_BN, tiles_s = p.tiles_x * p.tiles_y, nblk = gridDim.x;
    int bid = blockIdx.x;
    {   // each XCD works a contiguous run of tiles (the n-tiles of a pixel tile share its halo through that XCD's L2)
        const int xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int nt = bid % tiles_n, pt = bid / tiles_n;
    const int b = pt / tiles_s, ts = pt - b * tiles_s;
    const int ty0 = (ts / p.tiles_x) * C3_TH, tx0 = (ts % p.tiles_x) * C3_TW;
    const int n0 = nt * C3_BN;
    const int Hs = UP ? p.H >> 1 : p.H, Ws = UP ? p.W >> 1 : p.W;
    const float* xb = p.x + (size_t)b * Hs * Ws * p.Cin;
    // GroupNorm's folded affine of this sample, [scale Cin | shift Cin], behind the halo buffers: read per slab from LDS.
    // (Loaded from global memory at its use it sat behind an s_waitcnt vmcnt(0) in every halo work item -- seven exposed L2
    // round trips per slab that also drained the weight prefetch.)
    const float* scs = (const float*)(smem_raw + C3_HALO_BYTES);
    if (PRO) {
        float* w_ = (float*)(smem_raw + C3_HALO_BYTES);
        for (int i = tid * 4; i < 2 * p.Cin; i += 1024)
            *(f32x4*)(w_ + i) = i < p.Cin ? *(const f32x4*)(p.pro_scale + (size_t)b * p.Cin + i)
                                          : *(const f32x4*)(p.pro_shift + (size_t)b * p.Cin + (i - p.Cin));
        __syncthreads();
    }

    // ---- halo staging: work item f = tid + 256 u -> halo pixel f >> 3, float4 (4 channels) f & 7 of the 32-channel slab ----
    // (source / destination offsets are re-derived per use: kept in registers they cost 14 VGPRs the main loop does not have)
    auto h_item = [&](int u, int& src, int& dst) {
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));     // opaque: hipcc must not hoist these (slab-invariant) offsets out of the slab loop
        const int f = tid_o + 256 * u, px = f >> 3, c4 = f & 7;
        const int hy = px / C3_HW, hx = px - hy * C3_HW;
        int yy = ty0 + hy - 1, xx = tx0 + hx - 1;
        const bool ok = px < C3_HPX && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
        if (UP) { yy >>= 1; xx >>= 1; }
        src = ok ? (yy * Ws + xx) * p.Cin + c4 * 4 : -1;      // element offset of the source pixel's channel quad (-1: zeros)
        dst = px < C3_HPX ? px * C3_PXP + c4 * 4 : -1;       // halves, inside a plane
    };
    // (in two halves, work items [0, 4) and [4, 7): 16 instead of 28 registers of halo data in flight next to the 64
    //  accumulators and the two sets of weight fragments)
    f32x4 hv[C3_NF4];
    // (hv goes in as a parameter and the asm operands are its elements themselves: no register copy between a load and its wait)
    auto halo_load = [&](f32x4 (&hvr)[C3_NF4], int slab, auto u0_, auto u1_) {
        constexpr int U0 = decltype(u0_)::value, U1 = decltype(u1_)::value;
#pragma unroll
        for (int u = U0; u < U1; ++u) {
            int src, dst;
            h_item(u, src, dst);
            // asm like the weight loads below (the compiler's own s_waitcnt for a load it knows would be vmcnt(0): it cannot see the
            // asm loads around it, and would drain the weight ring); a pixel outside the image loads pixel 0 and is zeroed below
            const unsigned off = (unsigned)((src < 0 ? 0 : src) + slab * 32) * 4u;
            const float* base = xb;      // (a generic lambda's asm operand cannot name the captured variable itself)
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(hvr[u]) : "v"(off), "s"(base) : "memory");
        }
    };
    // wait_: the loads issued since the matching halo_load (they may stay in flight)
    auto halo_write = [&](f32x4 (&hvr)[C3_NF4], int slab, auto u0_, auto u1_, auto wait_) {
        constexpr int U0 = decltype(u0_)::value, U1 = decltype(u1_)::value, WAIT = decltype(wait_)::value;
        _Float16* hb = halo + (slab & 1) * (2 * C3_HPL);
#pragma unroll
        for (int u = U0; u < U1; ++u) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(hvr[u]) : "n"(WAIT));
        f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};     // the thread's channel quad is the same for every work item
        if (PRO) {
            sc = *(const f32x4*)(scs + slab * 32 + (tid & 7) * 4);
            sh = *(const f32x4*)(scs + p.Cin + slab * 32 + (tid & 7) * 4);
        }
#pragma unroll
        for (int u = U0; u < U1; ++u) {
            int src, dst;
            h_item(u, src, dst);
            if (dst < 0) continue;
            f32x4 v = hvr[u];
            if (PRO) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = v[e] * sc[e] + sh[e];
                    v[e] = t * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504f * t));   // swish (as conv_f16x2.hip)
                }
            }
            const bool ok = src >= 0;               // zero padding lives in the activated domain
            h4 s0, s1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = ok ? v[e] : 0.f;
                s0[e] = ds_split_hi(a);
                s1[e] = ds_split_lo(a, s0[e]);
            }
            *(h4*)(hb + dst) = s0;
            *(h4*)(hb + C3_HPL + dst) = s1;
        }
    };
    // ---- wave tile: all four tile rows (4 MFMA block rows) x output channels wave * 32 .. + 31 --------------------------------
    // Its B fragments of k-step q = (slab * 9 + tap) * 2 + ks are [plane][64 lanes][8]: two 16-byte loads per lane straight from
    // L2, no two waves loading the same bytes.  C3_RING k-steps are held in a register ring: the fragments of k-step q + RING - 1
    // are requested at the start of k-step q into the slot k-step q - 1 has just consumed, i.e. (RING - 1) * 12 MFMAs ahead.
    const int nslab = p.Cin >> 5;
    // The loads are written as asm: hipcc sinks compiler-visible prefetch loads towards their first use (the whole ring then
    // refills in one burst right before it is needed); asm volatile keeps them where they are written, in order, and the
    // matching wait is explicit -- s_waitcnt vmcnt(2 (RING - 1)) leaves exactly the younger k-steps in flight (loads the compiler
    // issues in between, the halo's, only make that wait conservative).
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const char* wbase = (const char*)(p.w + (size_t)nt * nslab * 9 * C3_WSTEP + wave_u * 1024);
    const unsigned voff0 = lane * 16, voff1 = lane * 16 + 8192;          // plane 1 sits 4096 halves behind plane 0
    const int nq = nslab * 18;
    h8 bq[C3_RING][2];
    auto w_load = [&](int q, h8 (&f)[2]) {
        const char* r = wbase + (size_t)(q >> 1) * (C3_WSTEP * 2) + (q & 1) * 1024;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(f[0]) : "v"(voff0), "s"(r) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(f[1]) : "v"(voff1), "s"(r) : "memory");
    };
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const std::integral_constant<int, 0> U_A{};
    const std::integral_constant<int, 4> U_B{};
    const std::integral_constant<int, C3_NF4> U_C{};
    const std::integral_constant<int, 12> W12{};
    halo_load(hv, 0, U_A, U_C);
#pragma unroll
    for (int q = 0; q < C3_RING - 1; ++q) w_load(q, bq[q]);         // (nq >= 18 > RING - 1)
    halo_write(hv, 0, U_A, U_C, std::integral_constant<int, 2 * (C3_RING - 1)>{});
    __syncthreads();
    for (int slab = 0; slab < nslab; ++slab) {
        const bool more_slabs = slab + 1 < nslab;
        const _Float16* ab = halo + (slab & 1) * (2 * C3_HPL) + l31 * C3_PXP + hh * 8;    // lane's pixel column and k half
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap == 0 && more_slabs) halo_load(hv, slab + 1, U_A, U_B);      // first half of the next slab: lands under taps 0, 1
            if (tap == 3 && more_slabs) halo_load(hv, slab + 1, U_B, U_C);      // second half: under taps 3, 4
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int ql = tap * 2 + ks;                                 // k-step inside the slab; 18 % C3_RING == 0
                const int qn = slab * 18 + ql + C3_RING - 1;
                w_load(qn < nq ? qn : nq - 1, bq[(ql + C3_RING - 1) % C3_RING]);     // (past the end: a harmless reload, no branch)
                h8 fa0[4], fa1[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const _Float16* ar = ab + ((i + ky) * C3_HW + kx) * C3_PXP + ks * 16;     // an immediate offset
                    fa0[i] = *(const h8*)ar;
                    fa1[i] = *(const h8*)(ar + C3_HPL);
                }
                asm volatile("s_waitcnt vmcnt(%2)" : "+v"(bq[ql % C3_RING][0]), "+v"(bq[ql % C3_RING][1]) : "n"(2 * (C3_RING - 1)));
                const h8 b0 = bq[ql % C3_RING][0], b1 = bq[ql % C3_RING][1];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x16 c = acc[i];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa1[i], b0, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0[i], b1, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0[i], b0, c, 0, 0, 0);
                    acc[i] = c;
                }
            }
            // the other halo buffer is free: every wave passed the barrier that ended the slab before this one
            if (tap == 2 && more_slabs) halo_write(hv, slab + 1, U_A, U_B, W12);      // 3 taps x 2 k-steps x 2 loads since its halo_load
            if (tap == 5 && more_slabs) halo_write(hv, slab + 1, U_B, U_C, W12);
        }
        __syncthreads();            // slab s is read, slab s + 1 is written
    }

    // The asm loads still in flight (the ring's clamped refills of the last k-steps) must land BEFORE anything reuses their
    // registers: hipcc does not know they are outstanding.  The ring goes INTO the drain as operands: a load whose result is
    // never read is dead to the compiler, which then gives every such load the same throw-away register and reuses it at once.
#pragma unroll
    for (int q = 0; q < C3_RING; ++q) asm volatile("s_waitcnt vmcnt(0)" : "+v"(bq[q][0]), "+v"(bq[q][1])::"memory");

    // ---- epilogue: two passes of 64 tile rows (block rows 2 pass, 2 pass + 1 of every wave) staged as fp32 [64][128] in LDS ----
    const float osc = p.out_scale;
    float* Tf = (float*)smem_raw;
    const int cc = tid & 31, col = n0 + cc * 4;
    double s_acc[4] = {0.0, 0.0, 0.0, 0.0}, q_acc[4] = {0.0, 0.0, 0.0, 0.0};
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias4 = *(const f32x4*)(p.bias + col);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {                     // every wave: its block rows 2 pass, 2 pass + 1, columns wave * 32 ..
            const int cl = wave * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                Tf[rl * C3_BN + cl] = acc[2 * pass + i][r] * osc;
            }
        }
        __syncthreads();
        // staged row rl (0..63) = tile row 2 pass + (rl >> 5), pixel rl & 31
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int rl = (tid >> 5) + 8 * it;
            const int yy = ty0 + 2 * pass + (rl >> 5), xx = tx0 + (rl & 31);
            if (yy < p.H && xx < p.W) {
                f32x4 val = *(const f32x4*)(Tf + rl * C3_BN + cc * 4) + bias4;
                const size_t o = (((size_t)b * p.H + yy) * p.W + xx) * p.Cout + col;
                if (p.R) val += *(const f32x4*)(p.R + o);
                *(f32x4*)(p.y + o) = val;
                if (STATS) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const double d = (double)val[e];
                        s_acc[e] += d;
                        q_acc[e] += d * d;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (STATS) {
        // the 8 threads that share a column group (tid & 31) add up through LDS; thread (tid < 32) writes 4 channels
        double* Td = (double*)smem_raw;           // [8][32][8]
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            Td[((tid >> 5) * 32 + cc) * 8 + e] = s_acc[e];
            Td[((tid >> 5) * 32 + cc) * 8 + 4 + e] = q_acc[e];
        }
        __syncthreads();
        if (tid < 32) {
            double* o = p.gn_part + (((size_t)b * tiles_s + ts) * 2) * p.Cout;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double s = 0.0, q = 0.0;
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    s += Td[(g * 32 + cc) * 8 + e];
                    q += Td[(g * 32 + cc) * 8 + 4 + e];
                }
                o[col + e] = s;
                o[p.Cout + col + e] = q;
            }
        }
    }
}

template <int PRO, int UP, bool STATS>
static int conv3_launch(const Conv3Params& p, hipStream_t s) {
    // two halo slabs (52 224 bytes) + the sample's [scale | shift] of the prologue
    const size_t lds = (size_t)C3_HALO_BYTES + (PRO ? (size_t)2 * p.Cin * sizeof(float) : 0);
    static_assert((2 * 2 * C3_HPL) * 2 >= 64 * C3_BN * 4, "the staged output half tile fits");
    static DsOnce attr_set;
    if (attr_set.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)ds_conv3x3_f16x2_kernel<PRO, UP, STATS>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C3_HALO_BYTES + 2 * C3_MAXCIN * 4);
        if (e != hipSuccess) {
            ds_set_error("conv3x3_f16x2: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return -2;
        }
        attr_set.done();
    }
    const long long blocks = (long long)p.B * p.tiles_x * p.tiles_y * (p.Cout / C3_BN);
    hipLaunchKernelGGL((ds_conv3x3_f16x2_kernel<PRO, UP, STATS>), dim3((unsigned)blocks), dim3(256), lds, s, p);
    DS_CHECK_LAUNCH();
    return 0;
}

extern "C" int ds_conv3x3_tiles(int H, int W) {
    return H > 0 && W > 0 ? ((H + C3_TH - 1) / C3_TH) * ((W + C3_TW - 1) / C3_TW) : -1;
}

extern "C" int ds_conv3x3_f16x2(const float* x, const void* w2, long long w_halves, float out_scale, const float* bias,
                                const float* residual, float* y, int B, int H, int W, int Cin, int Cout, int up,
                                const float* pro_scale, const float* pro_shift, double* gn_part, ds_stream_t stream) {
    DS_CHECK_ARG(x && w2 && y, "null pointer");
    DS_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cin % 32 == 0 && Cin <= C3_MAXCIN && Cout > 0 && Cout % C3_BN == 0,
                 "Cin % 32 == 0, Cin <= 1024 and Cout % 128 == 0");
    DS_CHECK_ARG(up == 0 || (up == 1 && H % 2 == 0 && W % 2 == 0), "up: 0, or 1 (source is H/2 x W/2, nearest-upsampled)");
    DS_CHECK_ARG((pro_scale == nullptr) == (pro_shift == nullptr), "prologue: both of scale / shift or neither");
    DS_CHECK_ARG(!(up == 1 && pro_scale), "no prologue on the upsampling conv");
    DS_CHECK_ARG(w_halves == (long long)2 * Cout * 9 * Cin && out_scale > 0.f, "packed weights: 2 * Cout * 9 * Cin halves");
    DS_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)w2 & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)residual & 15) == 0 &&
                     ((uintptr_t)bias & 15) == 0 && ((uintptr_t)pro_scale & 15) == 0 && ((uintptr_t)pro_shift & 15) == 0,
                 "operands must be 16-byte aligned");
    DS_CHECK_ARG((long long)B * H * W * (Cin > Cout ? Cin : Cout) < (1ll << 31), "32-bit element offsets");
    // the halo loads address a sample's source image with a 32-bit BYTE offset ((src + slab * 32) * 4u)
    DS_CHECK_ARG((long long)(up ? H / 2 : H) * (up ? W / 2 : W) * Cin < (1ll << 30), "32-bit byte offsets inside a sample");
    Conv3Params p;
    p.x = x; p.w = (const _Float16*)w2; p.bias = bias; p.R = residual; p.y = y;
    p.pro_scale = pro_scale; p.pro_shift = pro_shift; p.gn_part = gn_part; p.out_scale = out_scale;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.tiles_x = (W + C3_TW - 1) / C3_TW; p.tiles_y = (H + C3_TH - 1) / C3_TH;
    hipStream_t s = (hipStream_t)stream;
    const bool st = gn_part != nullptr;
    if (up == 1) return st ? conv3_launch<0, 1, true>(p, s) : conv3_launch<0, 1, false>(p, s);
    if (pro_scale) return st ? conv3_launch<1, 0, true>(p, s) : conv3_launch<1, 0, false>(p, s);
    return st ? conv3_launch<0, 0, true>(p, s) : conv3_launch<0, 0, false>(p, s);
}
