// Fused multi-head attention (head dim 64) on the fp16 matrix cores with the 2-way fp16 split of
// gemm_f16x2.hip: fp32-class results at 3 fp16-MFMA passes per contraction instead of the fp32 MFMA's
// 16x slower rate.  Same contract as ds_attention (attention.hip); replaces FullAttention /
// CrossAttention cores (transformer_utils.py:43-58, :91-109) when the denoiser runs in f16x2 mode.
//
//   pass 1  S^T[key][q] = sum_d K[key][d] Q[q][d]:  k = k0 + k1, q = q0 + q1 (fp16), MFMAs k1q0 + k0q1 + k0q0.
//           A operand = K rows from LDS (two fp16 planes, 128-byte rows, 16-byte chunks XOR-swizzled by
//           (key>>1)&7 -> conflict-free ds_read_b128), B operand = the wave's own Q rows in registers.
//           Transposed scores: lane&31 is the query, the keys of a tile are spread over the 16 accumulator
//           registers, so the softmax is in-lane + one cross-half shuffle (as in attention.hip).
//           MFMA row i of a 32-key tile is fed with key pi(i) (bits 2 and 3 of i swapped, ds_attn_pi), so that
//           accumulator register r of lane half h holds key (r&3) + 4((r>>2)&1) + 8h + 16(r>>3): the 8 registers
//           of a k-step are 8 CONSECUTIVE keys.
//   pass 2  O[q][d] = sum_key P[q][key] V[key][d]:  the P registers are split (p0 + p1) and packed straight
//           into the MFMA A operand: k-step s of a 32-key tile takes registers 8s..8s+7 = keys 16s + 8*half + e.
//           V is staged TRANSPOSED in natural key order, VT[plane][d][key], so the B operand is one ds_read_b128
//           per plane (16-byte chunks swizzled by (d>>2)&3).  MFMAs p1v0 + p0v1 + p0v0.
// K and V^T share one 72 KB LDS buffer (K first), so two workgroups fit per CU.
//
// READY variant (the denoiser's path): Q arrives as two fp16 planes and K / V^T as ready-made LDS images
// (common.h "attention-ready operands", written by the QKV / cross-Q GEMM epilogues and ds_attn_pack_kv), so
// staging is 1 KB LDS-DMA transfers with no conversion work, and the V^T transfer runs under the softmax.
// The generic variant below converts fp32 Q / K / V itself and was latency-bound on that staging (per workgroup
// ~12 us of load -> convert -> ds_write chains against ~4 us of MFMA work).
#include "common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

#define AH_WAVES 3

__device__ __forceinline__ _Float16 ah_hi(float a) { return ds_split_hi(a); }
__device__ __forceinline__ _Float16 ah_lo(float a, _Float16 h) { return ds_split_lo(a, h); }

typedef __attribute__((address_space(1))) const void* ah_gptr;
typedef __attribute__((address_space(3))) void* ah_lptr;

template <int NKT, bool READY>
__global__ __launch_bounds__(AH_WAVES * 64, 2) void ds_attn_f16x2_kernel(const float* __restrict__ Q, int ldq,
                                                                        const float* __restrict__ Kp, int ldk,
                                                                        const float* __restrict__ Vp, int ldv,
                                                                        float* __restrict__ O, int ldo, int Lq, int Lk,
                                                                        int heads, float scale, long long o_plane,
                                                                        long long q_plane) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NKEY = NKT * 32;
    constexpr int KPL = NKEY * 64;   // halves per K plane   ([key][64 d])
    constexpr int VPL = 64 * NKEY;   // halves per V^T plane ([d][NKEY keys, permuted])
    _Float16* buf = (_Float16*)smem_raw;   // K: [2][NKEY][64]   then   V^T: [2][64][NKEY]   (same size)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int head = blockIdx.x % heads;
    const int grp = blockIdx.x / heads;
    const int b = blockIdx.y;
    const int q0 = (grp * AH_WAVES + wave) * 32;
    const bool active = q0 < Lq;   // wave-uniform

    const float* kb = Kp + (size_t)b * Lk * ldk + head * 64;
    const float* vb = Vp + (size_t)b * Lk * ldv + head * 64;
#ifdef AH_TIMING   // probe build only (tools/attn_timing.py): per-workgroup s_memrealtime stamps through the unused V pointer
    unsigned long long ah_ts[6];
#define AH_STAMP(i_) do { ah_ts[i_] = __builtin_amdgcn_s_memrealtime(); } while (0)
    AH_STAMP(0);
#else
#define AH_STAMP(i_) do { } while (0)
#endif

    // READY: this (sample, head)'s image  K hi | K lo | V^T hi | V^T lo  (bytes), NKEY/4 KB per operand
    const unsigned char* img = (const unsigned char*)Kp + ((size_t)b * heads + head) * (size_t)(8 * KPL);
    constexpr int NDMA = NKEY / 4 / AH_WAVES;   // 1 KB transfers per wave per operand: 24 (self) / 8 (cross)
    static_assert(NKEY / 4 % AH_WAVES == 0, "whole transfers per wave");
    if constexpr (READY) {
#pragma unroll
        for (int j = 0; j < NDMA; ++j) {
            const int piece = j * AH_WAVES + wave;
            __builtin_amdgcn_global_load_lds((ah_gptr)(img + piece * 1024 + lane * 16),
                                             (ah_lptr)(smem_raw + piece * 1024), 16, 0, 0);
        }
    } else {
        // ---- stage K: fp32 -> two fp16 planes, 8-byte half-chunks, chunk swizzle (key>>1)&7 ----
        // (loads are issued 8 deep before the first dependent conversion: a load -> convert -> ds_write chain
        //  per iteration would serialise 24 L2/HBM round trips per thread)
        constexpr int NIT = NKEY * 16 / (AH_WAVES * 64);   // 24 (self) / 8 (cross), exact
        static_assert(NIT % 8 == 0, "staging loop is unrolled 8 deep");
        for (int it0 = 0; it0 < NIT; it0 += 8) {
            f32x4 v[8];
    #pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int f = tid + (it0 + u) * (AH_WAVES * 64);
                const int row = f >> 4, c4 = f & 15;          // 4 consecutive d at c4*4
                v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (row < Lk) v[u] = *(const f32x4*)(kb + (size_t)row * ldk + c4 * 4);
            }
    #pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int f = tid + (it0 + u) * (AH_WAVES * 64);
                const int row = f >> 4, c4 = f & 15;
                h4 s0, s1;
    #pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s0[e] = ah_hi(v[u][e]);
                    s1[e] = ah_lo(v[u][e], s0[e]);
                }
                const int chunk = (c4 >> 1) ^ ((row >> 1) & 7);
                _Float16* dst = buf + row * 64 + chunk * 8 + (c4 & 1) * 4;
                *(h4*)dst = s0;
                *(h4*)(dst + KPL) = s1;
            }
        }
    }
    // ---- Q operand: lane (q = l31, half hh) keeps Q[q][16 ks + 8 hh + 0..7], ks = 0..3, both planes ----
    h8 q_hi[4], q_lo[4];
    if constexpr (READY) {
        int qr = q0 + l31;
        if (qr >= Lq) qr = Lq - 1;
        const _Float16* qp = (const _Float16*)Q + (((size_t)b * heads + head) * Lq + qr) * 64 + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            q_hi[ks] = *(const h8*)(qp + 16 * ks);
            q_lo[ks] = *(const h8*)(qp + q_plane + 16 * ks);
        }
    } else {
        int qr = q0 + l31;
        if (qr >= Lq) qr = Lq - 1;
        const float* qp = Q + ((size_t)b * Lq + qr) * ldq + head * 64 + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const f32x4 a = *(const f32x4*)(qp + 16 * ks), c = *(const f32x4*)(qp + 16 * ks + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                q_hi[ks][e] = ah_hi(a[e]);
                q_lo[ks][e] = ah_lo(a[e], q_hi[ks][e]);
                q_hi[ks][4 + e] = ah_hi(c[e]);
                q_lo[ks][4 + e] = ah_lo(c[e], q_hi[ks][4 + e]);
            }
        }
    }
    // (explicit: hipcc does not reliably add the vmcnt(0) an in-flight LDS-DMA needs before a barrier, see gemm_f16x2.hip)
    if constexpr (READY) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // K is in LDS
    AH_STAMP(1);

    f32x16 s[NKT];
    if (active) {
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
            const int key = kt * 32 + ds_attn_pi(l31);
            const _Float16* kr = buf + key * 64;
            const int sw = (key >> 1) & 7;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int off = ((2 * ks + hh) ^ sw) * 8;
                const h8 k0 = *(const h8*)(kr + off), k1 = *(const h8*)(kr + KPL + off);
                f32x16 c = s[kt];
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, q_hi[ks], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, q_lo[ks], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, q_hi[ks], c, 0, 0, 0);
                s[kt] = c;
            }
        }
    }
    AH_STAMP(2);
    __syncthreads();  // everyone is done reading K

    if constexpr (READY) {   // the V^T image replaces K in the same buffer; the transfer runs under the softmax
#pragma unroll 4   // (the score registers are live here: a fully unrolled loop's 24 address pairs would spill)
        for (int j = 0; j < NDMA; ++j) {
            const int piece = j * AH_WAVES + wave;
            __builtin_amdgcn_global_load_lds((ah_gptr)(img + 4 * KPL + piece * 1024 + lane * 16),
                                             (ah_lptr)(smem_raw + piece * 1024), 16, 0, 0);
        }
    } else {
        // ---- stage V transposed + key-permuted: VT[plane][d][kt*32 + s*16 + half*8 + e], chunk swizzle (d>>2)&3 ----
        // Work item = (group of 4 consecutive keys, 4 consecutive d): the 4 keys are e&3 = 0..3 of one (tile, s, half,
        // e>>2) slot, i.e. 4 contiguous halves of a V^T row -> one ds_write_b64 per (d, plane) instead of four b16 writes.
        constexpr int NITV = NKEY / 4 * 16 / (AH_WAVES * 64);   // 6 (self) / 2 (cross)
    #pragma unroll
        for (int it = 0; it < NITV; ++it) {
            const int f = tid + it * (AH_WAVES * 64);
            const int kg = f >> 4, c4 = f & 15;                 // keys 4kg .. 4kg+3, d = 4 c4 .. 4 c4 + 3
            f32x4 v[4];
    #pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                const int key = 4 * kg + kx;
                v[kx] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (key < Lk) v[kx] = *(const f32x4*)(vb + (size_t)key * ldv + c4 * 4);
            }
            const int e0 = (4 * kg) & 7;                        // e = e0 + kx inside the 8-key chunk
            const int chunk = (4 * kg) >> 3;
    #pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = c4 * 4 + j;
                h4 s0, s1;
    #pragma unroll
                for (int kx = 0; kx < 4; ++kx) {
                    s0[kx] = ah_hi(v[kx][j]);
                    s1[kx] = ah_lo(v[kx][j], s0[kx]);
                }
                _Float16* dst = buf + d * NKEY + ((chunk ^ ((d >> 2) & 3)) * 8) + e0;
                *(h4*)dst = s0;
                *(h4*)(dst + VPL) = s1;
            }
        }
    }

    // ---- softmax over keys (fp32, in registers) ----
    // The longest VALU stretch of a wave (measured in-kernel: 5.9 of 23 us), so it is kept to 4 operations per score:
    // the row maximum is taken on the RAW scores (scale > 0), keys past Lk are masked only in the tiles that contain
    // them, each exponential is one v_fma (scale into the log2 domain and subtract the maximum) + one v_exp, and the
    // division by the row sum is applied to the 32 output values after P V instead of to the 144 probabilities (the
    // unnormalised e <= 1 splits into fp16 planes exactly as well).
    float inv = 1.f;
    if (active) {
        const float sl = scale * 1.4426950408889634f;   // exp(x) = 2^(x log2 e)
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (kt * 32 + 32 > Lk) {                    // wave-uniform: this tile holds keys >= Lk
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * 32 + (r & 3) + 4 * ((r >> 2) & 1) + 8 * hh + 16 * (r >> 3);   // pi(MFMA row)
                    s[kt][r] = key < Lk ? s[kt][r] : -INFINITY;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float msl = mx * sl;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], sl, -msl));   // masked: 2^-inf = 0
                s[kt][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 32);
        inv = 1.f / sum;
    }
    AH_STAMP(3);
    if constexpr (READY) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // V^T is in LDS
    AH_STAMP(4);

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    if (active) {
        const _Float16* v0r = buf + l31 * NKEY;          // d = l31
        const _Float16* v1r = buf + (32 + l31) * NKEY;   // d = 32 + l31   ((d>>2)&3 is the same for both)
        const int sw = (l31 >> 2) & 3;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                h8 p0, p1;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float pv = s[kt][8 * st + e];
                    p0[e] = (_Float16)pv;                 // p in [0, 1]
                    p1[e] = (_Float16)(pv - (float)p0[e]);
                }
                const int off = ((kt * 4 + st * 2 + hh) ^ sw) * 8;
                const h8 va0 = *(const h8*)(v0r + off), va1 = *(const h8*)(v0r + VPL + off);
                const h8 vb0 = *(const h8*)(v1r + off), vb1 = *(const h8*)(v1r + VPL + off);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(p1, va0, o0, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(p0, va1, o0, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(p0, va0, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(p1, vb0, o1, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(p0, vb1, o1, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(p0, vb0, o1, 0, 0, 0);
            }
    }
    if (active) {   // normalise: output register r holds query (r & 3) + 8 (r >> 2) + 4 hh, whose 1 / sum sits in that lane
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float iq = __shfl(inv, (r & 3) + 8 * (r >> 2) + 4 * hh);
            o0[r] *= iq;
            o1[r] *= iq;
        }
    }
    AH_STAMP(5);
    if (o_plane > 0) {
        // packed split planes for the f16x2 projection GEMM (K = ldo): each wave stages its 32 x 64 tile (hi, lo) in
        // the now free LDS and stores 16-byte chunks (8 d of one row) instead of 2-byte pieces
        __syncthreads();                                   // every wave is done reading V^T
        if (active) {
            _Float16* T = buf + wave * (2 * 32 * 64);      // [plane][32 rows][64 d]
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (r & 3) + 8 * (r >> 2) + 4 * hh;
                const _Float16 a0 = ds_split_hi(o0[r]), a1 = ds_split_hi(o1[r]);
                T[rl * 64 + l31] = a0;
                T[rl * 64 + 32 + l31] = a1;
                T[2048 + rl * 64 + l31] = ds_split_lo(o0[r], a0);
                T[2048 + rl * 64 + 32 + l31] = ds_split_lo(o1[r], a1);
            }
            // (LDS operations of one wave complete in order: no barrier between its own writes and reads)
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int c = lane + 64 * it, pl = c >> 8, rl = (c >> 3) & 31, ch = c & 7;
                const int qr = q0 + rl;
                if (qr < Lq) {
                    const h8 val = *(const h8*)(T + pl * 2048 + rl * 64 + ch * 8);
                    *(h8*)((_Float16*)O + (size_t)pl * o_plane + ds_packed_off(b * Lq + qr, head * 64 + ch * 8, ldo >> 5)) = val;
                }
            }
        }
    } else if (active) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qr = q0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (qr < Lq) {
                const size_t off = ((size_t)b * Lq + qr) * ldo + head * 64 + l31;
                O[off] = o0[r];
                O[off + 32] = o1[r];
            }
        }
    }
#ifdef AH_TIMING
    if (READY && Vp && lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* o = (unsigned long long*)Vp + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * AH_WAVES + wave) * 8;
        for (int i = 0; i < 6; ++i) o[i] = ah_ts[i];
        o[6] = __builtin_amdgcn_s_memrealtime();
        o[7] = active;
    }
#endif
}

template <bool READY>
static int attn_f16x2_launch(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o,
                             int ldo, int B, int heads, int Lq, int Lk, float scale, long long o_plane,
                             long long q_plane, hipStream_t stream) {
    DS_CHECK_ARG(q && k && (READY || v) && o, "null pointer");
    DS_CHECK_ARG(B > 0 && heads > 0 && Lq > 0 && Lk > 0, "bad shape");
    // the softmax takes the row maximum on the RAW scores and folds `scale` into the exp2 FMA: correct for scale > 0 only
    DS_CHECK_ARG(scale > 0.f, "scale must be positive");
    DS_CHECK_ARG(READY || (ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0), "leading dims must be multiples of 4");
    const int qtiles = (Lq + 31) / 32;
    const int groups = (qtiles + AH_WAVES - 1) / AH_WAVES;
    dim3 grid(groups * heads, B), block(AH_WAVES * 64);
    static bool attr9 = false, attr3 = false;
    if (Lk <= 96) {
        const size_t lds = (size_t)2 * 96 * 64 * sizeof(unsigned short);
        if (!attr3) {
            (void)hipFuncSetAttribute((const void*)ds_attn_f16x2_kernel<3, READY>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr3 = true;
        }
        hipLaunchKernelGGL((ds_attn_f16x2_kernel<3, READY>), grid, block, lds, stream, q, ldq, k, ldk, v, ldv, o, ldo, Lq,
                           Lk, heads, scale, o_plane, q_plane);
    } else {
        DS_CHECK_ARG(Lk <= 288, "at most 288 keys are supported");
        const size_t lds = (size_t)2 * 288 * 64 * sizeof(unsigned short);
        if (!attr9) {
            hipError_t e = hipFuncSetAttribute((const void*)ds_attn_f16x2_kernel<9, READY>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) {
                ds_set_error("attention_f16x2: hipFuncSetAttribute: %s", hipGetErrorString(e));
                return -2;
            }
            attr9 = true;
        }
        hipLaunchKernelGGL((ds_attn_f16x2_kernel<9, READY>), grid, block, lds, stream, q, ldq, k, ldk, v, ldv, o, ldo, Lq,
                           Lk, heads, scale, o_plane, q_plane);
    }
    DS_CHECK_LAUNCH();
    return 0;
}

extern "C" int ds_attention_f16x2(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o,
                                  int ldo, int B, int heads, int Lq, int Lk, float scale, ds_stream_t stream) {
    return attn_f16x2_launch<false>(q, ldq, k, ldk, v, ldv, o, ldo, B, heads, Lq, Lk, scale, 0, 0, (hipStream_t)stream);
}

// output written as packed split planes (2 planes of ceil16(B*Lq) * ldo halves), ldo % 32 == 0
extern "C" int ds_attention_f16x2_split(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                                        void* oh, int ldo, int B, int heads, int Lq, int Lk, float scale,
                                        ds_stream_t stream) {
    DS_CHECK_ARG(ldo % 32 == 0 && ldo >= heads * 64, "packed output needs ldo % 32 == 0");
    return attn_f16x2_launch<false>(q, ldq, k, ldk, v, ldv, (float*)oh, ldo, B, heads, Lq, Lk, scale,
                                    (long long)((B * Lq + 15) & ~15) * ldo, 0, (hipStream_t)stream);
}

// nkey of the K / V^T images for Lk keys: the kernel is built for 96 (cross) and 288 (self) key slots
extern "C" int ds_attn_nkey(int Lk) { return Lk <= 96 ? 96 : (Lk <= 288 ? 288 : -1); }

// Attention on attention-ready operands (common.h): qh = Q planes [2][B][heads][Lq][64] (q_plane halves apart),
// kv_img = [B][heads][4][nkey*64] halves (K hi | K lo | V^T hi | V^T lo, rows of keys >= Lk zero), output as in
// ds_attention_f16x2_split.  Bit-identical to ds_attention_f16x2_split on the same values.
#ifdef AH_TIMING
static const float* g_ah_timing_buf = nullptr;       // probe build: 8 x u64 per wave (ds_attn_timing_buffer)
extern "C" void ds_attn_timing_buffer(void* p) { g_ah_timing_buf = (const float*)p; }
#define AH_TIMING_V g_ah_timing_buf
#else
#define AH_TIMING_V nullptr
#endif
extern "C" int ds_attention_f16x2_ready(const void* qh, long long q_plane, const void* kv_img, void* oh, int ldo, int B,
                                        int heads, int Lq, int Lk, float scale, ds_stream_t stream) {
    DS_CHECK_ARG(ldo % 32 == 0 && ldo >= heads * 64, "packed output needs ldo % 32 == 0");
    DS_CHECK_ARG(q_plane >= (long long)B * heads * Lq * 64 && q_plane % 8 == 0, "Q plane stride");
    DS_CHECK_ARG(((uintptr_t)qh & 15) == 0 && ((uintptr_t)kv_img & 15) == 0, "operands must be 16-byte aligned");
    return attn_f16x2_launch<true>((const float*)qh, 0, (const float*)kv_img, 0, AH_TIMING_V, 0, (float*)oh, ldo, B, heads,
                                   Lq, Lk, scale, (long long)((B * Lq + 15) & ~15) * ldo, q_plane, (hipStream_t)stream);
}

// fp32 K | V rows (kv [B*Lk][ld], K of head h at column h*64, V at v_col + h*64) -> K / V^T images, zero rows
// beyond Lk.  One thread per (sample, head, key, 4 consecutive d).  Used once per batch for the caption K/V.
__global__ __launch_bounds__(256) void ds_attn_pack_kv_kernel(const float* __restrict__ kv, int ld, int v_col,
                                                              _Float16* __restrict__ img, int B, int heads, int Lk,
                                                              int nkey) {
    const long long f = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * heads * nkey * 16;
    if (f >= total) return;
    const int c4 = (int)(f & 15);
    const int key = (int)((f >> 4) % nkey);
    const long long bh = (f >> 4) / nkey;
    const int head = (int)(bh % heads), b = (int)(bh / heads);
    f32x4 kx = {0.f, 0.f, 0.f, 0.f}, vx = {0.f, 0.f, 0.f, 0.f};
    if (key < Lk) {
        const float* r = kv + ((size_t)b * Lk + key) * ld + head * 64 + c4 * 4;
        kx = *(const f32x4*)r;
        vx = *(const f32x4*)(r + v_col);
    }
    const int pl = nkey * 64;
    _Float16* im = img + (size_t)bh * 4 * pl;
    h4 k0, k1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        k0[e] = ds_split_hi(kx[e]);
        k1[e] = ds_split_lo(kx[e], k0[e]);
    }
    const int ko = ds_attn_k_off(key, c4 * 4);      // 4 consecutive d stay inside one 8-half chunk
    *(h4*)(im + ko) = k0;
    *(h4*)(im + pl + ko) = k1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int vo = 2 * pl + ds_attn_vt_off(key, c4 * 4 + e, nkey);
        const _Float16 hi = ds_split_hi(vx[e]);
        im[vo] = hi;
        im[vo + pl] = ds_split_lo(vx[e], hi);
    }
}

extern "C" int ds_attn_pack_kv(const float* kv, int ld, int v_col, void* img, int B, int heads, int Lk,
                               ds_stream_t stream) {
    DS_CHECK_ARG(kv && img && B > 0 && heads > 0 && Lk > 0 && ld % 4 == 0 && v_col % 4 == 0, "bad arguments");
    const int nkey = ds_attn_nkey(Lk);
    DS_CHECK_ARG(nkey > 0, "at most 288 keys are supported");
    const long long total = (long long)B * heads * nkey * 16;
    hipLaunchKernelGGL(ds_attn_pack_kv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, kv,
                       ld, v_col, (_Float16*)img, B, heads, Lk, nkey);
    DS_CHECK_LAUNCH();
    return 0;
}
