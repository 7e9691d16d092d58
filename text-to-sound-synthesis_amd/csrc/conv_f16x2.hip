// Convolutions over channels-last tensors as implicit GEMMs on the fp16 matrix cores with the 2-way fp16 split of
// gemm_f16x2.hip (fp32-class results, 3 MFMA passes): the gather-GEMM loaders of gemm_f32.hip (same loader semantics,
// same prologues, same epilogues) in front of the split arithmetic, instead of the 16x slower fp32 MFMA:
//   DS_LOAD_CONV2D   3x3 convs of the SpecVQGAN decoder / encoder (specvqgan/modules/diffusionmodules/model.py:37-77,92-151)
//   DS_LOAD_CONV1D   MelGAN's k7 / dilated k3 Conv1d with ReflectionPad1d (vocoder/modules.py:72-85,95-127)
//   DS_LOAD_CONVT1D  MelGAN's ConvTranspose1d(k = 2r, s = r) as r polyphase GEMMs (blockIdx.y = phase), DS_STORE_CONVT
//   DS_LOAD_DENSE    1x1 convs (ResnetBlock shortcut / second conv), optional LeakyReLU(0.2) prologue
//
//   C[m][n] = sum_{tap, c} pro(X)[pixel(m) + tap][c] * W[n][tap][c] * 2^s ... * 2^-s + bias[n] (+ R[m][n])
//
// A is gathered and split while it is staged (zero padding, nearest-2x upsample or stride-2 as index math; the
// GroupNorm affine + swish prologue is applied before the split); W arrives as two row-major fp16 planes of
// W * 2^s (_lib.split_f16x2).  256 threads = 4 waves (2x2), block tile BM x BN x 32, the LDS layout and fragment
// reads of gemm_f16x2.hip (64-byte rows, 16-byte chunks XOR-swizzled by (row>>2)&3), register-staged double
// buffering: the raw loads of tile t+1 are issued before the MFMAs of tile t, prologue + split + LDS write after.
#include "common.h"
#include <string.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CLD 32  // halves per LDS row

template <int BM, int BN, int LOADER, int PRO>
__global__ __launch_bounds__(256, 2) void ds_conv2d_f16x2_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int SA = BM / 64, SB = BN / 64;            // 8-element staging chunks per thread
    constexpr int APL = BM * CLD, BPL = BN * CLD;        // plane strides (halves)
    constexpr int STAGE = 2 * (APL + BPL);
    _Float16* smem = (_Float16*)smem_raw;                // [2 stages]{ A[2][BM][32], B[2][BN][32] }

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + BN - 1) / BN, nblk = gridDim.x;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
    const int g = blockIdx.y;                            // group = ConvTranspose1d phase (1 group otherwise)
    const float* Ag = p.A + (size_t)g * p.a_gstride;

    // staging chunk c = tid + 256 i: row c>>2, the 8 consecutive k at (c&3)*8 of the 32-wide k-tile
    int a_b[SA], a_y[SA], a_x[SA], a_dst[SA];
    const int ck8 = (tid & 3) * 8;
#pragma unroll
    for (int i = 0; i < SA; ++i) {
        const int row = (tid + 256 * i) >> 2;
        int m = m0 + row;
        if (m >= p.M) m = p.M - 1;
        if constexpr (LOADER == DS_LOAD_CONV2D) {
            const int hw = p.H * p.W_;
            a_b[i] = m / hw;
            const int rem = m - a_b[i] * hw;
            a_y[i] = rem / p.W_;
            a_x[i] = rem - a_y[i] * p.W_;
        } else if constexpr (LOADER == DS_LOAD_CONV1D) {
            a_b[i] = m / p.W_;
            a_x[i] = m - a_b[i] * p.W_;
            a_y[i] = 0;
        } else if constexpr (LOADER == DS_LOAD_CONVT1D) {   // rows of phase g are (b, q'); source index s0 = q' + (g < p)
            a_b[i] = m / p.ct_tin;
            a_x[i] = m - a_b[i] * p.ct_tin + (g < p.ct_p ? 1 : 0);
            a_y[i] = 0;
        } else {                                             // dense rows
            a_b[i] = 0;
            a_x[i] = m;
            a_y[i] = 0;
        }
        a_dst[i] = row * CLD + (((tid & 3) ^ ((row >> 2) & 3)) * 8);
    }
    const unsigned short* w2 = (const unsigned short*)p.W + (size_t)g * p.w_gstride;
    const unsigned short* b_base[SB];
    int b_dst[SB];
#pragma unroll
    for (int j = 0; j < SB; ++j) {
        const int row = (tid + 256 * j) >> 2;
        int n = n0 + row;
        if (n >= p.N) n = p.N - 1;
        b_base[j] = w2 + (size_t)n * p.ldw + ck8;
        b_dst[j] = row * CLD + (((tid & 3) ^ ((row >> 2) & 3)) * 8);
    }
    const size_t pl1 = (size_t)p.w3_plane;

    unsigned okmask = 0;
    f32x4 ra[2 * SA];
    u32x4 rb0[SB], rb1[SB];
    // raw loads of k-tile k0: address math + two 16-byte loads per slot, no branch on loaded data
#define C_LOAD(k0_)                                                                                 \
    do {                                                                                            \
        const int tap = LOADER == DS_LOAD_DENSE ? 0 : (k0_) / p.Cin;                                \
        const int c0 = (k0_) - tap * p.Cin + ck8;                                                   \
        _Pragma("unroll") for (int i = 0; i < SA; ++i) {                                            \
            const float* src;                                                                       \
            if constexpr (LOADER == DS_LOAD_CONV2D) {                                               \
                const int ky = tap / 3, kx = tap - ky * 3;                                          \
                int sy = a_y[i] + ky - 1, sx = a_x[i] + kx - 1;                                     \
                bool ok = sy >= 0 && sy < p.H && sx >= 0 && sx < p.W_;                              \
                sy = ok ? sy : a_y[i];                                                              \
                sx = ok ? sx : a_x[i];                                                              \
                int hs = p.H, ws = p.W_;                                                            \
                if (p.up == 1) { sy >>= 1; sx >>= 1; hs >>= 1; ws >>= 1; }                          \
                if (p.up == 2) {                                                                    \
                    hs = 2 * p.H; ws = 2 * p.W_;                                                    \
                    sy = 2 * a_y[i] + ky; sx = 2 * a_x[i] + kx;                                     \
                    ok = sy < hs && sx < ws;                                                        \
                    sy = ok ? sy : 2 * a_y[i];                                                      \
                    sx = ok ? sx : 2 * a_x[i];                                                      \
                }                                                                                   \
                okmask = ok ? (okmask | (1u << i)) : (okmask & ~(1u << i));                         \
                src = Ag + ((size_t)(a_b[i] * hs + sy) * ws + sx) * p.Cin + c0;                     \
            } else if constexpr (LOADER == DS_LOAD_CONV1D) {                                        \
                int ts = a_x[i] + (tap - (p.taps - 1) / 2) * p.dil;   /* ReflectionPad1d */         \
                if (ts < 0) ts = -ts;                                                               \
                if (ts >= p.W_) ts = 2 * (p.W_ - 1) - ts;                                           \
                okmask |= 1u << i;                                                                  \
                src = Ag + ((size_t)a_b[i] * p.W_ + ts) * p.Cin + c0;                               \
            } else if constexpr (LOADER == DS_LOAD_CONVT1D) {                                       \
                int s_ = a_x[i] - tap;   /* tap 0: x[s0] * W[:,:,phase]; 1: x[s0-1] * W[:,:,phase+r] */ \
                const bool ok = s_ >= 0 && s_ < p.ct_tin;                                           \
                s_ = ok ? s_ : 0;                                                                   \
                okmask = ok ? (okmask | (1u << i)) : (okmask & ~(1u << i));                         \
                src = Ag + ((size_t)a_b[i] * p.ct_tin + s_) * p.Cin + c0;                           \
            } else {                                                                                \
                okmask |= 1u << i;                                                                  \
                src = (p.A2 && (k0_) >= p.k_split) ? p.A2 + (size_t)a_x[i] * p.lda2 + (c0 - p.k_split) /* second row source */ \
                                                   : Ag + (size_t)a_x[i] * p.lda + c0;              \
            }                                                                                       \
            ra[2 * i] = *(const f32x4*)src;                                                         \
            ra[2 * i + 1] = *(const f32x4*)(src + 4);                                               \
        }                                                                                           \
        _Pragma("unroll") for (int j = 0; j < SB; ++j) {                                            \
            rb0[j] = *(const u32x4*)(b_base[j] + (k0_));                                            \
            rb1[j] = *(const u32x4*)(b_base[j] + pl1 + (k0_));                                      \
        }                                                                                           \
    } while (0)
    // prologue (GroupNorm affine + swish on the raw activations), zero padding, split, LDS write
#define C_WRITE(k0_, stage_)                                                                        \
    do {                                                                                            \
        _Float16* As_ = smem + (stage_) * STAGE;                                                    \
        _Float16* Bs_ = As_ + 2 * APL;                                                              \
        const int ch0 = LOADER == DS_LOAD_DENSE ? (k0_) + ck8 : (k0_) - ((k0_) / p.Cin) * p.Cin + ck8; \
        _Pragma("unroll") for (int i = 0; i < SA; ++i) {                                            \
            h8 s0, s1;                                                                              \
            const bool ok = (okmask >> i) & 1u;                                                     \
            _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                         \
                f32x4 v = ra[2 * i + q];                                                            \
                if constexpr (PRO == DS_PRO_AFFINE_SWISH) {                                         \
                    const f32x4 sc = *(const f32x4*)(p.pro_scale + (size_t)a_b[i] * p.Cin + ch0 + 4 * q); \
                    const f32x4 sh = *(const f32x4*)(p.pro_shift + (size_t)a_b[i] * p.Cin + ch0 + 4 * q); \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                 \
                        const float y = v[e] * sc[e] + sh[e];                                       \
                        v[e] = y * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504f * y)); /* swish by v_exp / v_rcp: the prologue runs once per tap on every element and bounded the kernel with expf + IEEE division */ \
                    }                                                                               \
                }                                                                                   \
                if constexpr (PRO == DS_PRO_LRELU) {                                                \
                    if (!(LOADER == DS_LOAD_DENSE && p.A2 && (k0_) >= p.k_split)) {   /* (the second row source is taken as it is) */ \
                        _Pragma("unroll") for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e]; \
                    }                                                                               \
                }                                                                                   \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                     \
                    const float a = ok ? v[e] : 0.f;   /* zero padding lives in the activated domain */ \
                    s0[4 * q + e] = ds_split_hi(a);                                                 \
                    s1[4 * q + e] = ds_split_lo(a, s0[4 * q + e]);                                  \
                }                                                                                   \
            }                                                                                       \
            *(h8*)(As_ + a_dst[i]) = s0;                                                            \
            *(h8*)(As_ + APL + a_dst[i]) = s1;                                                      \
        }                                                                                           \
        _Pragma("unroll") for (int j = 0; j < SB; ++j) {                                            \
            *(u32x4*)(Bs_ + b_dst[j]) = rb0[j];                                                     \
            *(u32x4*)(Bs_ + BPL + b_dst[j]) = rb1[j];                                               \
        }                                                                                           \
    } while (0)

    const int swz[2] = {((0 + hh) ^ ((l31 >> 2) & 3)) * 8, ((2 + hh) ^ ((l31 >> 2) & 3)) * 8};
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K / 32;
    C_LOAD(0);
    C_WRITE(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) C_LOAD((kt + 1) * 32);
        {
            const _Float16* Ac = smem + cur * STAGE + (wm * TM * 32 + l31) * CLD;
            const _Float16* Bc = smem + cur * STAGE + 2 * APL + (wn * TN * 32 + l31) * CLD;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                h8 fa0[TM], fa1[TM], fb0[TN], fb1[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    fa0[i] = *(const h8*)(Ac + i * 32 * CLD + swz[ks]);
                    fa1[i] = *(const h8*)(Ac + APL + i * 32 * CLD + swz[ks]);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    fb0[j] = *(const h8*)(Bc + j * 32 * CLD + swz[ks]);
                    fb1[j] = *(const h8*)(Bc + BPL + j * 32 * CLD + swz[ks]);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        f32x16 c = acc[i][j];
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa1[i], fb0[j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0[i], fb1[j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0[i], fb0[j], c, 0, 0, 0);
                        acc[i][j] = c;
                    }
            }
        }
        if (more) C_WRITE((kt + 1) * 32, cur ^ 1);
        __syncthreads();
    }

    // epilogue: bias, residual, row-major store, staged through LDS for 16-byte accesses (as gemm_f16x2.hip)
    const float osc = p.out_scale;
    float* Tf = (float*)smem_raw;     // [BM][BN] floats: the operand stages are free after the last barrier
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int cl = (wn * TN + j) * 32 + l31;
            const float bv = (p.bias && n0 + cl < p.N) ? p.bias[n0 + cl] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                Tf[rl * BN + cl] = acc[i][j][r] * osc + bv;
            }
        }
    __syncthreads();
    constexpr int CPR = BN / 4;
    float* Cg = p.C + (size_t)g * p.c_gstride;
    const float* Rg = p.R ? p.R + (size_t)g * p.c_gstride : nullptr;
    for (int c = tid; c < BM * CPR; c += 256) {
        const int cc = c % CPR, rl = c / CPR;
        const int row = m0 + rl, col = n0 + cc * 4;
        if (row < p.M && col < p.N) {
            f32x4 val = *(const f32x4*)(Tf + rl * BN + cc * 4);
            if (Rg) val += *(const f32x4*)(Rg + (size_t)row * p.ldr + col);
            size_t orow = row;
            if (p.store == DS_STORE_CONVT) {            // row (b, q') of phase g -> output time t = q r + g - p
                const int b = row / p.ct_tin, qq = row - b * p.ct_tin + (g < p.ct_p ? 1 : 0);
                orow = (size_t)b * p.ct_tin * p.ct_r + (qq * p.ct_r + g - p.ct_p);
            }
            *(f32x4*)(Cg + orow * p.ldc + col) = val;
        }
    }
}

template <int BM, int BN, int LOADER, int PRO>
static int conv_launch(const GemmParams& p, hipStream_t s) {
    const size_t stage = (size_t)2 * 2 * (BM + BN) * CLD * sizeof(unsigned short);
    const size_t tile = (size_t)BM * BN * sizeof(float);
    const size_t lds = stage > tile ? stage : tile;
    static DsOnce attr_set;
    if (attr_set.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)ds_conv2d_f16x2_kernel<BM, BN, LOADER, PRO>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            ds_set_error("conv_f16x2: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return -2;
        }
        attr_set.done();
    }
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    hipLaunchKernelGGL((ds_conv2d_f16x2_kernel<BM, BN, LOADER, PRO>), dim3(tiles, p.groups > 0 ? p.groups : 1), dim3(256),
                       lds, s, p);
    DS_CHECK_LAUNCH();
    return 0;
}

template <int LOADER, int PRO>
static int conv_tile(const GemmParams& p, hipStream_t s) {
    const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * (p.groups > 0 ? p.groups : 1);
    if (t128 >= 256 && p.N > 64) return conv_launch<128, 128, LOADER, PRO>(p, s);
    return conv_launch<64, 64, LOADER, PRO>(p, s);
}

// p.A: channels-last fp32 tensor; p.W: 2 row-major fp16 planes [groups][N][ldw] of W * 2^s (K ordered [tap][c]),
// w3_plane halves apart; p.out_scale = 2^-s.  loader: DS_LOAD_CONV2D (K = 9 Cin), DS_LOAD_CONV1D (K = taps Cin, reflect
// padding), DS_LOAD_CONVT1D (K = 2 Cin, groups = r phases, DS_STORE_CONVT) or DS_LOAD_DENSE (K = lda-contiguous rows).
int ds_launch_conv2d_f16x2(const GemmParams& p, hipStream_t stream, int loader) {
    DS_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0 && p.K % 32 == 0, "K must be a positive multiple of 32");
    DS_CHECK_ARG(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.W & 15) == 0 && ((uintptr_t)p.C & 15) == 0 &&
                     ((uintptr_t)p.R & 15) == 0,
                 "operands must be 16-byte aligned");
    DS_CHECK_ARG(p.ldw >= p.K && p.ldw % 8 == 0 && p.w3_plane % 8 == 0 &&
                     p.w3_plane >= (long long)(p.groups > 0 ? p.groups : 1) * p.N * p.ldw && p.w_gstride % 8 == 0,
                 "split-weight strides must be multiples of 8");
    DS_CHECK_ARG(p.N % 4 == 0 && p.ldc % 4 == 0 && (!p.R || p.ldr % 4 == 0), "N, ldc, ldr must be multiples of 4");
    DS_CHECK_ARG(p.act == DS_ACT_NONE, "no activation");
    DS_CHECK_ARG(p.out_scale > 0.f, "out_scale must be set (2^-s of the weight pre-scale)");
    switch (loader) {
        case DS_LOAD_CONV2D:
            DS_CHECK_ARG(p.Cin > 0 && p.Cin % 32 == 0 && p.K == 9 * p.Cin, "conv2d: K = 9*Cin, Cin % 32 == 0");
            DS_CHECK_ARG(p.H > 0 && p.W_ > 0 && p.M % (p.H * p.W_) == 0, "conv2d: M = samples * H * W");
            DS_CHECK_ARG(p.store == DS_STORE_ROW && p.groups <= 1, "conv2d: row store, no groups");
            DS_CHECK_ARG(p.pro == DS_PRO_NONE || (p.pro == DS_PRO_AFFINE_SWISH && p.pro_scale && p.pro_shift),
                         "conv2d prologue: none or GroupNorm affine + swish");
            return p.pro == DS_PRO_NONE ? conv_tile<DS_LOAD_CONV2D, DS_PRO_NONE>(p, stream)
                                        : conv_tile<DS_LOAD_CONV2D, DS_PRO_AFFINE_SWISH>(p, stream);
        case DS_LOAD_CONV1D:
            DS_CHECK_ARG(p.Cin > 0 && p.Cin % 32 == 0 && p.taps > 0 && p.K == p.taps * p.Cin && p.dil > 0,
                         "conv1d: K = taps*Cin, Cin % 32 == 0");
            DS_CHECK_ARG(p.W_ > 0 && p.M % p.W_ == 0 && (p.taps - 1) / 2 * p.dil < p.W_, "conv1d: M = samples * T");
            DS_CHECK_ARG(p.store == DS_STORE_ROW && p.groups <= 1, "conv1d: row store, no groups");
            DS_CHECK_ARG(p.pro == DS_PRO_NONE || p.pro == DS_PRO_LRELU, "conv1d prologue: none or LeakyReLU(0.2)");
            return p.pro == DS_PRO_NONE ? conv_tile<DS_LOAD_CONV1D, DS_PRO_NONE>(p, stream)
                                        : conv_tile<DS_LOAD_CONV1D, DS_PRO_LRELU>(p, stream);
        case DS_LOAD_CONVT1D:
            DS_CHECK_ARG(p.Cin > 0 && p.Cin % 32 == 0 && p.K == 2 * p.Cin, "convT1d: K = 2*Cin, Cin % 32 == 0");
            DS_CHECK_ARG(p.ct_r > 0 && p.groups == p.ct_r && p.ct_tin > 0 && p.M % p.ct_tin == 0 && p.ct_p >= 0 &&
                             p.ct_p < p.ct_r && p.store == DS_STORE_CONVT && !p.R,
                         "convT1d: groups = stride phases, M = samples * T_in, CONVT store, no residual");
            DS_CHECK_ARG(p.pro == DS_PRO_NONE || p.pro == DS_PRO_LRELU, "convT1d prologue: none or LeakyReLU(0.2)");
            return p.pro == DS_PRO_NONE ? conv_tile<DS_LOAD_CONVT1D, DS_PRO_NONE>(p, stream)
                                        : conv_tile<DS_LOAD_CONVT1D, DS_PRO_LRELU>(p, stream);
        case DS_LOAD_DENSE:
            DS_CHECK_ARG((p.A2 ? p.lda >= p.k_split && p.lda2 >= p.K - p.k_split && p.lda2 % 4 == 0 && p.k_split % 32 == 0 &&
                                     p.k_split > 0 && p.k_split < p.K && ((uintptr_t)p.A2 & 15) == 0
                               : p.lda >= p.K) && p.lda % 4 == 0 && p.store == DS_STORE_ROW && p.groups <= 1,
                         "dense: lda >= K (two sources: lda >= k_split, lda2 >= K - k_split, k_split % 32 == 0), row store, no groups");
            DS_CHECK_ARG(p.pro == DS_PRO_NONE || p.pro == DS_PRO_LRELU, "dense prologue: none or LeakyReLU(0.2)");
            return p.pro == DS_PRO_NONE ? conv_tile<DS_LOAD_DENSE, DS_PRO_NONE>(p, stream)
                                        : conv_tile<DS_LOAD_DENSE, DS_PRO_LRELU>(p, stream);
        default:
            DS_CHECK_ARG(false, "unknown loader");
    }
}

// MelGAN ResnetBlock tail in ONE contraction (vocoder/modules.py:72-85):  y = W2 LReLU(h) + Ws x + (b2 + bs)  -- the block's 1x1
// conv on the activated k3 output and its 1x1 shortcut, K = 2 C over the two row sources [LReLU(h) | x].  Against three
// launches (k3 conv | shortcut | 1x1 + residual) the shortcut tensor is never written and read back: two of the block's
// five tensor passes disappear on layers that are HBM-bound.  w = the fp16 planes of [W2 | Ws] * 2^s ([C][2 C], w_plane
// halves apart), bias = b2 + bs.
extern "C" int ds_melgan_resblock_tail(const float* h, const float* x, const void* w, long long w_plane, float out_scale,
                                       const float* bias, float* y, int M, int C, ds_stream_t stream) {
    DS_CHECK_ARG(h && x && w && y, "null pointer");
    DS_CHECK_ARG(M > 0 && C > 0 && C % 32 == 0, "C % 32 == 0");
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = h; p.A2 = x; p.W = (const float*)w; p.bias = bias; p.C = y;
    p.M = M; p.N = C; p.K = 2 * C; p.k_split = C;
    p.lda = C; p.lda2 = C; p.ldw = 2 * C; p.ldc = C; p.ldr = C; p.groups = 1;
    p.pro = DS_PRO_LRELU; p.act = DS_ACT_NONE; p.store = DS_STORE_ROW;
    p.w3_plane = w_plane; p.out_scale = out_scale;
    return ds_launch_conv2d_f16x2(p, (hipStream_t)stream, DS_LOAD_DENSE);
}

// The whole MelGAN ResnetBlock (vocoder/modules.py:72-85) behind one entry:
//   y = shortcut(x) + conv1x1(LReLU(conv_k3_dil(reflect_pad_dil(LReLU(x)))))
// x, y: [B][T][C] channels-last fp32; w3 = fp16 planes of W1 * 2^s1 ([C][3 C], K ordered [tap][channel], w3_plane halves
// apart), b3 [C]; wt / bt = the tail's [W2 | Ws] planes and b2 + bs.  C % 32 == 0, dil < T.
//   h != NULL: two launches -- the dilated k3 conv into the scratch tensor h [B][T][C], then the one-GEMM tail above;
//   h == NULL: the single-pass kernel (melgan_fused.hip: x read once, y written once, LReLU(h) never leaves registers) --
//              only where ds_melgan_resblock_fused_ok(T, C, dil) says so, an error otherwise.
int ds_launch_melgan_rb(const float* x, const void* w3, long long w3_plane, float w3_scale, const float* b3, const void* wt,
                        long long wt_plane, float wt_scale, const float* bt, float* y, int B, int T, int C, int dil, hipStream_t s);

extern "C" int ds_melgan_resblock(const float* x, const void* w3, long long w3_plane, float w3_scale, const float* b3,
                                  const void* wt, long long wt_plane, float wt_scale, const float* bt, float* h, float* y,
                                  int B, int T, int C, int dil, ds_stream_t stream) {
    DS_CHECK_ARG(x && w3 && wt && y && b3 && bt, "null pointer");
    DS_CHECK_ARG(B > 0 && T > 0 && C > 0 && C % 32 == 0 && dil > 0 && dil < T, "C % 32 == 0, 0 < dil < T");
    if (!h) {
        DS_CHECK_ARG(ds_melgan_resblock_fused_ok(T, C, dil), "h == NULL asks for the single-pass kernel: not built for this (T, C, dil)");
        DS_CHECK_ARG(w3_scale > 0.f && wt_scale > 0.f, "out scales must be set");
        return ds_launch_melgan_rb(x, w3, w3_plane, w3_scale, b3, wt, wt_plane, wt_scale, bt, y, B, T, C, dil, (hipStream_t)stream);
    }
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = x; p.W = (const float*)w3; p.bias = b3; p.C = h;
    p.M = B * T; p.N = C; p.K = 3 * C;
    p.lda = C; p.ldw = 3 * C; p.ldc = C; p.ldr = C; p.groups = 1;
    p.pro = DS_PRO_LRELU; p.act = DS_ACT_NONE; p.store = DS_STORE_ROW;
    p.Cin = C; p.W_ = T; p.taps = 3; p.dil = dil;
    p.w3_plane = w3_plane; p.out_scale = w3_scale;
    const int rc = ds_launch_conv2d_f16x2(p, (hipStream_t)stream, DS_LOAD_CONV1D);
    if (rc) return rc;
    return ds_melgan_resblock_tail(h, x, wt, wt_plane, wt_scale, bt, y, B * T, C, stream);
}
