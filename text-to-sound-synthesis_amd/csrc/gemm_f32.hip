// fp32 gather-GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 FMA chain).
//
//   C[m][n] = store( act( sum_k A(m,k) * W[n][k] + bias[n] ) + R[m][n] )
//
// One kernel serves every contraction on the Diffsound path (reference ops it replaces:
// nn.Linear in transformer_utils.py:31-36,75-82,248-253,345-348; Conv2d 3x3/1x1 in
// specvqgan/modules/diffusionmodules/model.py:92-226,570-671; Conv1d / ConvTranspose1d in
// vocoder/modules.py:72-127).  The A operand is gathered on the fly (implicit GEMM) by a
// loader: dense rows, 3x3 conv over a channels-last image (optionally nearest-2x upsampled,
// model.py:48-52), dilated reflect-padded conv1d, or polyphase transposed conv1d.  An
// elementwise prologue (GroupNorm affine [+swish], LeakyReLU) is applied while staging A, so
// normalised/activated tensors are never written to HBM.
//
// Tiling: 256 threads = 4 waves (2x2); block tile BM x BN x 32; each wave owns TMxTN 32x32
// accumulators.  LDS rows are padded to 36 floats so that the ds_read_b128 fragment reads are
// bank-conflict free (16 consecutive rows start at distinct multiples of 4 banks).  One
// ds_read_b128 per fragment feeds 4 MFMAs: lane half h supplies k = 8c+4h+j for MFMA j, which
// both operands agree on, so the contraction index set is covered exactly once.
// Register-staged double buffering: the raw global loads of tile t+1 are all issued (no
// dependent branch between them) before the MFMAs of tile t; the prologue transform and the LDS
// write happen after the MFMAs, so HBM/L2 latency hides under 64-cycle MFMAs; one barrier per
// k-tile.  Loader and prologue are template parameters: the hot dense/no-prologue instance has a
// branch-free staging path.
#include "common.h"

#define BK 32
#define LDT 36  // padded LDS row (floats)

template <int PRO>
__device__ __forceinline__ f32x4 ds_pro(const GemmParams& p, f32x4 v, int sample, int ch) {
    if constexpr (PRO == DS_PRO_LRELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.2f * v[j];
    } else if constexpr (PRO != DS_PRO_NONE) {
        const f32x4 s = *(const f32x4*)(p.pro_scale + (size_t)sample * p.Cin + ch);
        const f32x4 o = *(const f32x4*)(p.pro_shift + (size_t)sample * p.Cin + ch);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float y = v[j] * s[j] + o[j];
            if constexpr (PRO == DS_PRO_AFFINE_SWISH) y = y / (1.f + expf(-y));
            v[j] = y;
        }
    }
    return v;
}

template <int BM, int BN, int LOADER, int PRO>
__global__ __launch_bounds__(256) void ds_gemm_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TM = BM / 64, TN = BN / 64;  // 32x32 fragments per wave (2x2 wave grid)
    constexpr int SA = BM / 32, SB = BN / 32;  // float4 staging slots per thread
    float* As = smem;                 // [2][BM][LDT]
    float* Bs = smem + 2 * BM * LDT;  // [2][BN][LDT]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int g = blockIdx.y;

    // XCD-aware tile order: consecutive block ids land on different XCDs; give each XCD a
    // contiguous run of tiles (bijective for any grid size).
    const int tiles_n = (p.N + BN - 1) / BN;
    const int nblk = gridDim.x;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, idx = bid >> 3;
        const int q = nblk >> 3, r = nblk & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (bid / tiles_n) * BM;
    const int n0 = (bid % tiles_n) * BN;

    const float* Ag = p.A + (size_t)g * p.a_gstride;
    const float* Wg = p.W + (size_t)g * p.w_gstride;

    // ---- per-slot row decode (fixed across k) ----
    const int srow = tid >> 3;  // + 32*i
    const int kq = (tid & 7) * 4;
    const float* a_base[SA];
    int a_b[SA], a_y[SA], a_x[SA];
#pragma unroll
    for (int i = 0; i < SA; ++i) {
        int m = m0 + srow + 32 * i;
        if (m >= p.M) m = p.M - 1;
        if constexpr (LOADER == DS_LOAD_DENSE) {
            a_base[i] = Ag + (size_t)m * p.lda + kq;
            a_b[i] = (PRO == DS_PRO_AFFINE || PRO == DS_PRO_AFFINE_SWISH) ? m / p.rows_per_sample : 0;
            a_y[i] = a_x[i] = 0;
        } else if constexpr (LOADER == DS_LOAD_CONV2D) {
            const int hw = p.H * p.W_;
            a_b[i] = m / hw;
            const int rem = m - a_b[i] * hw;
            a_y[i] = rem / p.W_;
            a_x[i] = rem - a_y[i] * p.W_;
            a_base[i] = Ag;
        } else if constexpr (LOADER == DS_LOAD_CONV1D) {
            a_b[i] = m / p.W_;
            a_x[i] = m - a_b[i] * p.W_;
            a_y[i] = 0;
            a_base[i] = Ag + (size_t)a_b[i] * p.W_ * p.Cin;
        } else {  // CONVT1D: rows of phase g are (b, q'); source index s0 = q' + (g < p)
            a_b[i] = m / p.ct_tin;
            a_x[i] = m - a_b[i] * p.ct_tin + (g < p.ct_p ? 1 : 0);
            a_y[i] = 0;
            a_base[i] = Ag + (size_t)a_b[i] * p.ct_tin * p.Cin;
        }
    }
    const float* w_base[SB];
#pragma unroll
    for (int i = 0; i < SB; ++i) {
        int n = n0 + srow + 32 * i;
        if (n >= p.N) n = p.N - 1;
        w_base[i] = Wg + (size_t)n * p.ldw + kq;
    }

    // Raw load of one staging slot: address math + one global_load_dwordx4, no dependent branch
    // on the loaded value.  Out-of-image taps read a valid dummy address and are zeroed in
    // stage_finish (zero padding lives in the activated domain).
    unsigned okmask = 0;  // bit i: slot i of the tile being staged is inside the image
    auto stage_load = [&](int i, int k0) -> f32x4 {
        if constexpr (LOADER == DS_LOAD_DENSE) {
            return *(const f32x4*)(a_base[i] + k0);
        } else if constexpr (LOADER == DS_LOAD_CONV2D) {
            const int tap = k0 / p.Cin;
            const int c0 = k0 - tap * p.Cin + kq;
            const int ky = tap / 3, kx = tap - ky * 3;
            int sy = a_y[i] + ky - 1, sx = a_x[i] + kx - 1;
            bool ok = sy >= 0 && sy < p.H && sx >= 0 && sx < p.W_;
            sy = ok ? sy : a_y[i];
            sx = ok ? sx : a_x[i];
            int hs = p.H, ws = p.W_;
            if (p.up == 1) { sy >>= 1; sx >>= 1; hs >>= 1; ws >>= 1; }   // nearest-2x upsample of the source
            if (p.up == 2) {   // stride-2 conv over a (2H x 2W) source zero-padded on the right/bottom only
                hs = 2 * p.H; ws = 2 * p.W_;        // (Downsample, diffusionmodules/model.py:60-77)
                sy = 2 * a_y[i] + ky; sx = 2 * a_x[i] + kx;
                ok = sy < hs && sx < ws;
                sy = ok ? sy : 2 * a_y[i];
                sx = ok ? sx : 2 * a_x[i];
            }
            okmask = ok ? (okmask | (1u << i)) : (okmask & ~(1u << i));
            return *(const f32x4*)(a_base[i] + ((size_t)(a_b[i] * hs + sy) * ws + sx) * p.Cin + c0);
        } else if constexpr (LOADER == DS_LOAD_CONV1D) {
            const int tap = k0 / p.Cin;
            const int c0 = k0 - tap * p.Cin + kq;
            int ts = a_x[i] + (tap - (p.taps - 1) / 2) * p.dil;  // ReflectionPad1d
            if (ts < 0) ts = -ts;
            if (ts >= p.W_) ts = 2 * (p.W_ - 1) - ts;
            return *(const f32x4*)(a_base[i] + (size_t)ts * p.Cin + c0);
        } else {
            const int tap = k0 / p.Cin;  // 0: x[s0] * W[:,:,phase]; 1: x[s0-1] * W[:,:,phase+r]
            const int c0 = k0 - tap * p.Cin + kq;
            int s = a_x[i] - tap;
            const bool ok = s >= 0 && s < p.ct_tin;
            s = ok ? s : 0;
            okmask = ok ? (okmask | (1u << i)) : (okmask & ~(1u << i));
            return *(const f32x4*)(a_base[i] + (size_t)s * p.Cin + c0);
        }
    };
    auto stage_finish = [&](int i, int k0, f32x4 v) -> f32x4 {
        if constexpr (PRO != DS_PRO_NONE) {
            int ch = k0 + kq;
            if constexpr (LOADER != DS_LOAD_DENSE) ch = k0 - (k0 / p.Cin) * p.Cin + kq;
            v = ds_pro<PRO>(p, v, a_b[i], ch);
        }
        if constexpr (LOADER == DS_LOAD_CONV2D || LOADER == DS_LOAD_CONVT1D) {
            if (!((okmask >> i) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        return v;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[SA], rb[SB];
    const int nk = p.K / BK;

    // prologue: tile 0 -> LDS[0]
#pragma unroll
    for (int i = 0; i < SA; ++i) ra[i] = stage_load(i, 0);
#pragma unroll
    for (int i = 0; i < SB; ++i) rb[i] = *(const f32x4*)(w_base[i]);
#pragma unroll
    for (int i = 0; i < SA; ++i) *(f32x4*)(As + (srow + 32 * i) * LDT + kq) = stage_finish(i, 0, ra[i]);
#pragma unroll
    for (int i = 0; i < SB; ++i) *(f32x4*)(Bs + (srow + 32 * i) * LDT + kq) = rb[i];
    __syncthreads();

    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        const int k0n = (kt + 1) * BK;
        if (more) {
#pragma unroll
            for (int i = 0; i < SA; ++i) ra[i] = stage_load(i, k0n);
#pragma unroll
            for (int i = 0; i < SB; ++i) rb[i] = *(const f32x4*)(w_base[i] + k0n);
        }
        const float* Ac = As + cur * BM * LDT + (wm * TM * 32 + l31) * LDT + 4 * hh;
        const float* Bc = Bs + cur * BN * LDT + (wn * TN * 32 + l31) * LDT + 4 * hh;
#pragma unroll
        for (int ch = 0; ch < BK / 8; ++ch) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *(const f32x4*)(Ac + i * 32 * LDT + ch * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *(const f32x4*)(Bc + j * 32 * LDT + ch * 8);
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][c], fb[j][c], acc[i][j], 0, 0, 0);
        }
        if (more) {
            float* An = As + (cur ^ 1) * BM * LDT;
            float* Bn = Bs + (cur ^ 1) * BN * LDT;
#pragma unroll
            for (int i = 0; i < SA; ++i) *(f32x4*)(An + (srow + 32 * i) * LDT + kq) = stage_finish(i, k0n, ra[i]);
#pragma unroll
            for (int i = 0; i < SB; ++i) *(f32x4*)(Bn + (srow + 32 * i) * LDT + kq) = rb[i];
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* Cg = p.C + (size_t)g * p.c_gstride;
    const float* Rg = p.R ? p.R + (size_t)g * p.c_gstride : nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + l31;
            if (col >= p.N) continue;
            const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (row >= p.M) continue;
                float v = acc[i][j][r] + bv;
                if (p.f16_round) {  // fp16 Linear -> QuickGELU chain, each op rounded like the reference
                    v = ds_r16(v);
                    if (p.act == DS_ACT_GELU2) {
                        const float sg = ds_r16(1.f / (1.f + expf(-ds_r16(1.702f * v))));
                        v = ds_r16(v * sg);
                    }
                } else if (p.act == DS_ACT_GELU2) v = v / (1.f + expf(-1.702f * v));
                else if (p.act == DS_ACT_TANH) v = tanhf(v);
                size_t off;
                if (p.store == DS_STORE_ROW) {
                    off = (size_t)row * p.ldc + col;
                } else if (p.store == DS_STORE_BATCH_T) {
                    const int b = row / p.rows_per_sample, pp = row - b * p.rows_per_sample;
                    off = ((size_t)b * p.N + col) * p.ldc + pp;  // [b][n][ldc >= rows_per_sample]
                } else {
                    const int b = row / p.ct_tin, qq = row - b * p.ct_tin + (g < p.ct_p ? 1 : 0);
                    const int t = qq * p.ct_r + g - p.ct_p;
                    off = ((size_t)b * p.ct_tin * p.ct_r + t) * p.ldc + col;
                }
                if (Rg) {
                    v += Rg[(size_t)row * p.ldr + col];
                    if (p.f16_round) v = ds_r16(v);
                }
                Cg[off] = v;
            }
        }
    }
}

// ---- host side: tile selection + launch ----------------------------------------------------
template <int BM, int BN, int LOADER, int PRO>
static int launch_cfg(const GemmParams& p, hipStream_t s) {
    const size_t lds = (size_t)2 * (BM + BN) * LDT * sizeof(float);
    static DsOnce attr_set;
    if (attr_set.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)ds_gemm_kernel<BM, BN, LOADER, PRO>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            ds_set_error("gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return -2;
        }
        attr_set.done();
    }
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    dim3 grid(tiles, p.groups > 0 ? p.groups : 1);
    hipLaunchKernelGGL((ds_gemm_kernel<BM, BN, LOADER, PRO>), grid, dim3(256), lds, s, p);
    DS_CHECK_LAUNCH();
    return 0;
}

int g_last_tile = 0;           // tile config of the most recent launch (read by the profiler)
static int g_force_tile = -1;  // test hook: 0..2 forces a tile config, -1 = auto
extern "C" void ds_gemm_force_tile(int t) { g_force_tile = t; }

template <int LOADER, int PRO>
static int launch_tile(const GemmParams& p, hipStream_t s) {
    // Pick the tile whose block count quantises best over the 256 CUs: blocks co-reside
    // (LDS-limited) 2 / 3 / 4 per CU for 128x128 / 128x64 / 64x64; MFMA time per block ~ BM*BN.
    struct Cfg { int bm, bn, per_cu; double pen; };
    static const Cfg cfgs[3] = {{128, 128, 2, 1.00}, {128, 64, 3, 1.03}, {64, 64, 4, 1.06}};
    int best = 0;
    if (g_force_tile >= 0) {
        best = g_force_tile;
    } else {
        double bc = 1e300;
        const int groups = p.groups > 0 ? p.groups : 1;
        for (int c = 0; c < 3; ++c) {
            const long tiles = (long)((p.M + cfgs[c].bm - 1) / cfgs[c].bm) * ((p.N + cfgs[c].bn - 1) / cfgs[c].bn) * groups;
            // a CU works through ceil(tiles / 256) blocks, per_cu at a time sharing its 4 SIMDs
            const long per_cu_blocks = (tiles + 255) / 256;
            const double cost = (double)per_cu_blocks * cfgs[c].bm * cfgs[c].bn * cfgs[c].pen;
            if (cost < bc) { bc = cost; best = c; }
        }
    }
    g_last_tile = best;
    switch (best) {
        case 0: return launch_cfg<128, 128, LOADER, PRO>(p, s);
        case 1: return launch_cfg<128, 64, LOADER, PRO>(p, s);
        default: return launch_cfg<64, 64, LOADER, PRO>(p, s);
    }
}

int ds_launch_gemm(const GemmParams& p, hipStream_t stream, int loader) {
    DS_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "empty problem");
    DS_CHECK_ARG(p.K % BK == 0, "K must be a multiple of 32");
    DS_CHECK_ARG(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.W & 15) == 0, "A/W must be 16-byte aligned");
    DS_CHECK_ARG(p.ldw >= p.K && p.ldw % 4 == 0, "ldw must be >= K and a multiple of 4");
    const bool affine = p.pro == DS_PRO_AFFINE || p.pro == DS_PRO_AFFINE_SWISH;
    DS_CHECK_ARG(!affine || (p.pro_scale && p.pro_shift && p.Cin > 0), "affine prologue needs scale/shift/Cin");
    switch (loader) {
        case DS_LOAD_DENSE:
            DS_CHECK_ARG(p.lda % 4 == 0, "lda must be a multiple of 4");
            DS_CHECK_ARG(!affine || p.rows_per_sample > 0, "dense affine prologue needs rows_per_sample");
            switch (p.pro) {
                case DS_PRO_NONE: return launch_tile<DS_LOAD_DENSE, DS_PRO_NONE>(p, stream);
                case DS_PRO_AFFINE: return launch_tile<DS_LOAD_DENSE, DS_PRO_AFFINE>(p, stream);
                case DS_PRO_AFFINE_SWISH: return launch_tile<DS_LOAD_DENSE, DS_PRO_AFFINE_SWISH>(p, stream);
                case DS_PRO_LRELU: return launch_tile<DS_LOAD_DENSE, DS_PRO_LRELU>(p, stream);
            }
            break;
        case DS_LOAD_CONV2D:
            DS_CHECK_ARG(p.Cin % BK == 0 && p.K == 9 * p.Cin, "conv2d: K = 9*Cin, Cin % 32 == 0");
            switch (p.pro) {
                case DS_PRO_NONE: return launch_tile<DS_LOAD_CONV2D, DS_PRO_NONE>(p, stream);
                case DS_PRO_AFFINE_SWISH: return launch_tile<DS_LOAD_CONV2D, DS_PRO_AFFINE_SWISH>(p, stream);
            }
            break;
        case DS_LOAD_CONV1D:
            DS_CHECK_ARG(p.Cin % BK == 0 && p.K == p.taps * p.Cin, "conv1d: K = taps*Cin, Cin % 32 == 0");
            switch (p.pro) {
                case DS_PRO_NONE: return launch_tile<DS_LOAD_CONV1D, DS_PRO_NONE>(p, stream);
                case DS_PRO_LRELU: return launch_tile<DS_LOAD_CONV1D, DS_PRO_LRELU>(p, stream);
            }
            break;
        case DS_LOAD_CONVT1D:
            DS_CHECK_ARG(p.Cin % BK == 0 && p.K == 2 * p.Cin, "convT1d: K = 2*Cin, Cin % 32 == 0");
            switch (p.pro) {
                case DS_PRO_NONE: return launch_tile<DS_LOAD_CONVT1D, DS_PRO_NONE>(p, stream);
                case DS_PRO_LRELU: return launch_tile<DS_LOAD_CONVT1D, DS_PRO_LRELU>(p, stream);
            }
            break;
    }
    ds_set_error("gemm: unsupported loader/prologue combination %d/%d", loader, p.pro);
    return -1;
}
