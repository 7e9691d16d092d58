// Backward of the fused multi-head attention (head dim 64, exact-fp32 MFMA) for the training step (scope row 8f-3):
// what loss.backward() computes for FullAttention / CrossAttention (transformer_utils.py:43-58, :91-109)
//     P = softmax(scale Q K^T),  O = P V
//     dV = P^T dO,   dP = dO V^T,   dS = scale P (dP - delta),  delta_q = sum_d dO[q][d] O[q][d],
//     dQ = dS K,     dK = dS^T Q
// by TILE-WISE RECOMPUTATION: neither P nor dS ever exists in HBM (the composed version kept P [B*16][288][288] per
// layer -- 106 MB -- and moved it, dP and two transposes of them through HBM for every layer).
//
// Same formulation as attention.hip: a wave owns a 32-row tile of one index and holds the OTHER index in registers
// (transposed score tiles, 144 accumulator registers), all rows of the other operand sit in LDS (68-float rows).  The
// MFMA contracts over the register index for free, never over the lane index -- hence two kernels:
//   * ds_attn_bwd_q_kernel   wave = 32 queries, keys in registers:   S^T = K Q^T -> softmax (as the forward) -> row
//                            statistics L_q = max + log(sum), delta_q -> dP^T = V dO^T -> dS -> dQ = dS K
//   * ds_attn_bwd_kv_kernel  wave = 32 keys, queries in registers:   S = Q K^T -> P = exp(scale S - L_q) -> dV = P^T dO
//                            -> dP = dO V^T -> dS = scale P (dP - delta_q) -> dK = dS^T Q
// (7 tile products instead of the 5 of a formulation with cross-lane transposes; the exact-fp32 MFMA makes them the
// cost of the kernels: 2 Lq Lk 64 flops each at 157 TFLOP/s peak.)  One LDS buffer per workgroup, re-staged between
// the phases (K, V, K  /  Q, dO, Q), two workgroups per CU.
//
// SPLIT = true (round 5, ds_attention_bwd_f16x2: the "f16x2" training backend): the same two kernels and phases, but every
// tile product runs on v_mfma_f32_32x32x16_f16 with both operands as fp16 hi + lo (a0b0 + a0b1 + a1b0, fp32 accumulate -- the
// arithmetic of gemm_f16x2.hip): 12 MFMAs of 8 passes per 64-deep tile product instead of 32 of 16 passes, i.e. 5.3x less
// matrix-pipe time.  The LDS operand is split ONCE when it is staged (two swizzled fp16 planes, ab_stage_split), register
// fragments when they are loaded, the score / dS tiles right at the MFMA.  Only the k-index mapping of the fragments changes (a lane holds 8 consecutive d / 8 of its own
// accumulator rows per MFMA instead of 1).  P in [0, 1] is multiplied by 2^10 before it is split (its lo plane would sit in
// fp16's subnormal range otherwise) and dV by 2^-10 when it is stored; dO carries the step's loss scale and is folded into
// the step's saturation monitor (modeling/train.py) by the q kernel itself (ds_attention_bwd_f16x2_mon, round 6: max |dO| over
// the fragment every lane holds anyway, one wave reduction, an atomicMax that is skipped when the monitor already holds more).
// dS = scale P (dP - delta) exists only in registers and has NO fixed relation to dO: it can exceed it by |V| x 64 / 8, and
// where the probabilities are near-uniform it is the small difference of two nearly equal numbers -- against the reference at
// 19 layers / B = 20 (tests/test_hip_train_batch.py) the cross-attention dS sat 2^-20 under dO, its fp16 planes in the
// subnormal range, and dQ / dK came out with 1e-2 .. 6e-5 relative error depending on the loss scale.  So since round 6 dS is
// NORMALISED PER WAVE before it is split: the wave's max |dS| (all 9 x 16 accumulator values of its tiles, one reduction) goes to
// 2^8 .. 2^9 by an exact power of two that the dQ / dK store takes out again (ab_norm_scale) -- neither saturation nor the
// subnormal range can be reached, whatever the loss scale.
#include "common.h"

typedef _Float16 ab_h8 __attribute__((ext_vector_type(8)));

#define AB_LD 68
#define AB_WAVES 3
#define AB_NT (AB_WAVES * 64)
#define AB_FENCE()                                                     \
    do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

// rows [0, valid) of a [rows][64] operand (row stride ld) -> lds[NT * 32][AB_LD]; rows past `valid` are zero
template <int NT, int INFLIGHT>
__device__ __forceinline__ void ab_stage(float* __restrict__ lds, const float* __restrict__ src, int ld, int valid, int tid) {
    static_assert((NT * 32 * 16) % (AB_NT * INFLIGHT) == 0, "staging trip count");
    // an opaque copy of the thread id: the addresses of one staging pass are re-derived in the next one instead of being
    // kept live across the MFMA phases in between (they were spilled to scratch: the kernels run at the 256-register cap)
    asm volatile("" : "+v"(tid));
    for (int it0 = 0; it0 < NT * 32 * 16 / AB_NT; it0 += INFLIGHT) {
        f32x4 t8[INFLIGHT];
#pragma unroll
        for (int u = 0; u < INFLIGHT; ++u) {
            const int f = tid + (it0 + u) * AB_NT;
            const int row = f >> 4, c4 = (f & 15) * 4;
            t8[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < valid) t8[u] = *(const f32x4*)(src + (size_t)row * ld + c4);
        }
#pragma unroll
        for (int u = 0; u < INFLIGHT; ++u) {
            const int f = tid + (it0 + u) * AB_NT;
            const int row = f >> 4, c4 = (f & 15) * 4;
            *(f32x4*)(lds + row * AB_LD + c4) = t8[u];
        }
    }
}

// tile product with the register index on the ROWS of the result: acc[i = lds row (tile * 32 + lane & 31)][j = the
// lane's own row of the register operand]:  acc += X_lds[tile rows][d] * Y_reg[d]   (attention.hip pass 1)
struct AbFragF;
__device__ __forceinline__ void ab_rows_times_reg(f32x16& acc, const float* __restrict__ lds, int tile, int l31, int hh,
                                                  const f32x4 (&yf)[8]) {
    const float* xr = lds + (tile * 32 + l31) * AB_LD + 4 * hh;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const f32x4 xf = *(const f32x4*)(xr + 8 * c);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[j], yf[c][j], acc, 0, 0, 0);
    }
}

// out[i = lane's own index][d] += sum over the 32 register rows of `tile`:  W[i][row] * Z_lds[row][d]  (attention.hip
// pass 2); o0: d = lane & 31, o1: d = 32 + lane & 31
__device__ __forceinline__ void ab_reg_times_rows(f32x16& o0, f32x16& o1, const f32x16& w, const float* __restrict__ lds, int tile,
                                                  int l31, int hh) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const float* zr = lds + row * AB_LD + l31;
        o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[r], zr[0], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[r], zr[32], o1, 0, 0, 0);
    }
}

// lane (row = lane & 31, half hh) keeps X[row][8c + 4hh + j], c = 0..7, j = 0..3
struct AbFragF {
    f32x4 v[8];
};
__device__ __forceinline__ void ab_load_frag(AbFragF& f, const float* __restrict__ rowp, int hh, float pre = 1.f) {   // (pre: SPLIT only)
#pragma unroll
    for (int c = 0; c < 8; ++c) f.v[c] = *(const f32x4*)(rowp + 4 * hh + 8 * c);
}

// ---- SPLIT forms: the register fragment of a 64-deep row as fp16 hi / lo, k-block c (16 k) = X[row][16c + 8hh + 0..7] ----
struct AbFragH {
    ab_h8 hi[4], lo[4];
};
__device__ __forceinline__ void ab_split8(const f32x4& a, const f32x4& b, ab_h8& hi, ab_h8& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hi[e] = ds_split_hi(a[e]);
        lo[e] = ds_split_lo(a[e], hi[e]);
        hi[4 + e] = ds_split_hi(b[e]);
        lo[4 + e] = ds_split_lo(b[e], hi[4 + e]);
    }
}
__device__ __forceinline__ void ab_load_frag(AbFragH& f, const float* __restrict__ rowp, int hh, float pre = 1.f) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const f32x4 a = *(const f32x4*)(rowp + 8 * hh + 16 * c) * pre, b = *(const f32x4*)(rowp + 8 * hh + 16 * c + 4) * pre;
        ab_split8(a, b, f.hi[c], f.lo[c]);
    }
}
// m: wave-uniform max |x| of a register tile set.  up = the power of two that puts it in [2^8, 2^9), dn = 1 / up; both 1 when
// m is 0, subnormal-small, huge or not finite (exponent field outside [16, 240])
__device__ __forceinline__ void ab_norm_scale(float m, float& up, float& dn) {
    // (readfirstlane: the value is wave-uniform after the reduction -- the two factors live in SGPRs, not in the VGPR budget)
    const unsigned eb = ((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(m)) >> 23) & 0xffu;
    const bool ok = eb >= 16u && eb <= 240u;
    up = ok ? __uint_as_float((262u - eb) << 23) : 1.f;
    dn = ok ? __uint_as_float((eb - 8u) << 23) : 1.f;
}
__device__ __forceinline__ float ab_wave_max(float m) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));      // (fmaxf drops a NaN: it never wins)
    return m;
}
// max |x| over the fragment a lane holds (monitor: the hi plane is x to 2^-11 relative)
__device__ __forceinline__ float ab_frag_absmax(const AbFragH& f) {
    float m = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf((float)f.hi[c][e]));
    return m;
}
// delta_q = sum_d dO[q][d] O[q][d] needs the fp32 dO values of this lane's k-slots next to the split fragment
__device__ __forceinline__ float ab_frag_dot(const float* __restrict__ ap, const float* __restrict__ bp, int hh, bool split) {
    float acc = 0.f;
    if (split) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 a = *(const f32x4*)(ap + 8 * hh + 16 * c + 4 * h), b = *(const f32x4*)(bp + 8 * hh + 16 * c + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc += a[j] * b[j];
            }
    } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const f32x4 a = *(const f32x4*)(ap + 4 * hh + 8 * c), b = *(const f32x4*)(bp + 4 * hh + 8 * c);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += a[j] * b[j];
        }
    }
    return acc;
}
// SPLIT staging: the operand rows go to LDS ALREADY SPLIT -- two fp16 planes [rows][64 halves] (hi, then lo `AB_HPLANE(NT)`
// halves further), 128-byte rows without padding, the 16-byte chunk c of row r at chunk position c ^ (r & 7) (the 32 lanes
// of a fragment read cover all banks; 73.7 KB per workgroup against 78.3 KB of padded fp32 rows).  An element is converted
// ONCE per workgroup here instead of once per wave and use in the products (the first form of this kernel -- fp32 rows,
// conversion at the MFMA -- was VALU-bound: 104 / 115 us per launch against 143 / 160 on the fp32 MFMA).
#define AB_HPLANE(NT_) ((NT_) * 32 * 64)
__device__ __forceinline__ int ab_hoff(int row, int d) { return row * 64 + ((((d >> 3) ^ (row & 7))) << 3) + (d & 7); }
typedef _Float16 ab_h4 __attribute__((ext_vector_type(4)));
template <int NT, int INFLIGHT>
__device__ __forceinline__ void ab_stage_split(_Float16* __restrict__ lds, const float* __restrict__ src, int ld, int valid, int tid,
                                               float pre = 1.f) {
    static_assert((NT * 32 * 16) % (AB_NT * INFLIGHT) == 0, "staging trip count");
    asm volatile("" : "+v"(tid));
    for (int it0 = 0; it0 < NT * 32 * 16 / AB_NT; it0 += INFLIGHT) {
        f32x4 t8[INFLIGHT];
#pragma unroll
        for (int u = 0; u < INFLIGHT; ++u) {
            const int f = tid + (it0 + u) * AB_NT;
            const int row = f >> 4, c4 = (f & 15) * 4;
            t8[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < valid) t8[u] = *(const f32x4*)(src + (size_t)row * ld + c4);
        }
#pragma unroll
        for (int u = 0; u < INFLIGHT; ++u) {
            const int f = tid + (it0 + u) * AB_NT;
            const int row = f >> 4, c4 = (f & 15) * 4;
            ab_h4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = t8[u][e] * pre;                // (a power of two: exact)
                hi[e] = ds_split_hi(v);
                lo[e] = ds_split_lo(v, hi[e]);
            }
            _Float16* d = lds + ab_hoff(row, c4);          // 4 consecutive d stay inside one 8-half chunk
            *(ab_h4*)d = hi;
            *(ab_h4*)(d + AB_HPLANE(NT)) = lo;
        }
    }
}
// acc[i = lds row][j = the lane's own register-operand row] += X_lds[tile rows][d] * Y_reg[d], three split passes per k-block
template <int NT>
__device__ __forceinline__ void ab_rows_times_reg_h(f32x16& acc, const _Float16* __restrict__ lds, int tile, int l31, int hh,
                                                    const AbFragH& y) {
    const int row = tile * 32 + l31;
    const _Float16* xr = lds + row * 64;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int off = ((2 * c + hh) ^ (row & 7)) << 3;                       // k-block c: d = 16c + 8hh + 0..7
        const ab_h8 xh = *(const ab_h8*)(xr + off), xl = *(const ab_h8*)(xr + AB_HPLANE(NT) + off);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, y.hi[c], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, y.lo[c], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, y.hi[c], acc, 0, 0, 0);
    }
}
// out[i = lane's own index][d] += sum over the 32 register rows of `tile`: (wscale W[i][row]) * Z_lds[row][d].  MFMA m (0, 1)
// contracts the 16 rows held in accumulator registers 8m .. 8m+7 of the two lane halves: k-slot (hh, e) <-> row
// (e & 3) + 8 (2m + (e >> 2)) + 4 hh -- the A operand is the lane's own registers; the B operand is one COLUMN d of 4 + 4
// consecutive LDS rows per lane: two transposing reads (ds_read_b64_tr_b16: the 16 lanes of a group address a [4 rows][16 d]
// block, lane 4 j + r the four halves at (row0 + j, d0 + 4 r), and lane l receives column d0 + l of the four rows; lane map
// measured in tools/probe/tr/).  Rounds 5-6 read it with 8 ds_read_u16 + byte permutes per fragment: 1152 two-byte LDS reads
// and ~600 v_perm of the kv kernel's 11 000 instructions.
typedef short ab_v4s __attribute__((__vector_size__(4 * sizeof(short))));
__device__ __forceinline__ ab_h8 ab_tr8(const _Float16* p0, const _Float16* p1) {
    union { struct { ab_v4s a, b; } q; ab_h8 v; } u;
    u.q.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ab_v4s*)p0);
    u.q.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ab_v4s*)p1);
    return u.v;
}
template <int NT>
__device__ __forceinline__ void ab_reg_times_rows_split(f32x16& o0, f32x16& o1, const f32x16& w, const _Float16* __restrict__ lds,
                                                        int tile, int l31, int hh, float wscale) {
    // this lane's read address inside a [4 rows][16 d] block: row 4 hh + (l >> 2), d = 16 (l31 >> 4) + 4 (l & 3) (+ 32 for o1)
    const int rl = 4 * hh + ((l31 >> 2) & 3), cq = l31 & 3, ch = 2 * (l31 >> 4) + (cq >> 1);       // (rl < 8 = its swizzle key)
    const _Float16* zb = lds + (tile * 32 + rl) * 64 + (cq & 1) * 4;
    const _Float16* z0 = zb + ((ch ^ rl) << 3);
    const _Float16* z1 = zb + (((4 + ch) ^ rl) << 3);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        ab_h8 wh, wl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma clang fp contract(off)
            // (no saturation clamps here, unlike ds_split_hi / _lo: w * wscale is a probability x 2^10 or a dS tile the wave has
            //  just normalised to [2^8, 2^9) -- two v_med3 per value, 768 of the kv kernel's instructions, guarded nothing)
            const float wv = w[8 * m + e] * wscale;
            wh[e] = (_Float16)wv;
            wl[e] = (_Float16)(wv - (float)wh[e]);
        }
        // rows 16 m + 4 hh + 0..3 (k-slots e = 0..3) and + 8 (e = 4..7): 1024 and 512 halves apart
        const ab_h8 z0h = ab_tr8(z0 + m * 1024, z0 + m * 1024 + 512), z0l = ab_tr8(z0 + AB_HPLANE(NT) + m * 1024, z0 + AB_HPLANE(NT) + m * 1024 + 512);
        const ab_h8 z1h = ab_tr8(z1 + m * 1024, z1 + m * 1024 + 512), z1l = ab_tr8(z1 + AB_HPLANE(NT) + m * 1024, z1 + AB_HPLANE(NT) + m * 1024 + 512);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, z0h, o0, 0, 0, 0);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, z0l, o0, 0, 0, 0);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, z0h, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, z1h, o1, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, z1l, o1, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, z1h, o1, 0, 0, 0);
    }
}

// result tile [32 rows over registers][lane & 31 (+32) = d] -> dst rows row0 + .., for rows < limit (oscale: a power of two)
__device__ __forceinline__ void ab_store_tile(float* __restrict__ dst, int ld, int row0, int limit, const f32x16& o0, const f32x16& o1,
                                              int l31, int hh, float oscale = 1.f) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (row < limit) {
            float* p = dst + (size_t)row * ld + l31;
            p[0] = o0[r] * oscale;
            p[32] = o1[r] * oscale;
        }
    }
}

template <bool SPLIT> struct AbSel { typedef AbFragF Frag; };
template <> struct AbSel<true> { typedef AbFragH Frag; };
__device__ __forceinline__ void ab_rows_times_reg(f32x16& acc, const float* __restrict__ lds, int tile, int l31, int hh,
                                                  const AbFragF& y) {
    ab_rows_times_reg(acc, lds, tile, l31, hh, y.v);
}
// the same LDS buffer is fp32 rows (exact form) or fp16 planes (SPLIT): the *_any forms dispatch at compile time
template <bool SPLIT, int NT>
__device__ __forceinline__ void ab_reg_times_rows_any(f32x16& o0, f32x16& o1, const f32x16& w, const float* __restrict__ lds, int tile,
                                                      int l31, int hh, float wscale = 1.f) {
    if constexpr (SPLIT) ab_reg_times_rows_split<NT>(o0, o1, w, (const _Float16*)lds, tile, l31, hh, wscale);
    else ab_reg_times_rows(o0, o1, w, lds, tile, l31, hh);       // (wscale is 1 in the exact-fp32 form)
}
template <bool SPLIT, int NT, typename Frag>
__device__ __forceinline__ void ab_rows_times_reg_any(f32x16& acc, const float* __restrict__ lds, int tile, int l31, int hh,
                                                      const Frag& y) {
    if constexpr (SPLIT) ab_rows_times_reg_h<NT>(acc, (const _Float16*)lds, tile, l31, hh, y);
    else ab_rows_times_reg(acc, lds, tile, l31, hh, y);
}
template <bool SPLIT, int NT, int INFLIGHT>
__device__ __forceinline__ void ab_stage_any(float* __restrict__ lds, const float* __restrict__ src, int ld, int valid, int tid,
                                             float pre = 1.f) {
    if constexpr (SPLIT) ab_stage_split<NT, INFLIGHT>((_Float16*)lds, src, ld, valid, tid, pre);
    else ab_stage<NT, INFLIGHT>(lds, src, ld, valid, tid);       // (pre is 1 in the exact-fp32 form)
}
// floats of the operand buffer for NT row tiles
template <bool SPLIT> __host__ __device__ constexpr int ab_buf_floats(int nt) { return SPLIT ? nt * 32 * 64 : nt * 32 * AB_LD; }
#define AB_PSCALE 1024.f      // SPLIT: P * 2^10 is what gets split (its lo plane stays in fp16's normal range), dV * 2^-10 stored

template <int NKT, bool SPLIT>
__global__ __launch_bounds__(AB_NT, 2) void ds_attn_bwd_q_kernel(const float* __restrict__ Q, int ldq, const float* __restrict__ Kp,
                                                                int ldk, const float* __restrict__ Vp, int ldv,
                                                                const float* __restrict__ O, int ldo, const float* __restrict__ dO,
                                                                int lddo, float* __restrict__ dQ, int lddq, float* __restrict__ stats,
                                                                int Lq, int Lk, int heads, float scale, unsigned* __restrict__ amax,
                                                                float do_scale, float do_inv) {
    extern __shared__ __attribute__((aligned(16))) float kv[];  // [NKT*32][AB_LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int head = blockIdx.x % heads, grp = blockIdx.x / heads, b = blockIdx.y;
    const int q0 = (grp * AB_WAVES + wave) * 32;
    const bool active = q0 < Lq;             // wave-uniform
    const float* kb = Kp + (size_t)b * Lk * ldk + head * 64;
    const float* vb = Vp + (size_t)b * Lk * ldv + head * 64;
    int qr = q0 + l31;
    if (qr >= Lq) qr = Lq - 1;
    const size_t qrow = (size_t)b * Lq + qr;

    ab_stage_any<SPLIT, NKT, 4>(kv, kb, ldk, Lk, tid);
    f32x16 s[NKT];
    {
        typename AbSel<SPLIT>::Frag qf;
        ab_load_frag(qf, Q + qrow * ldq + head * 64, hh);
        __syncthreads();
        if (active) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
                ab_rows_times_reg_any<SPLIT, NKT>(s[kt], kv, kt, l31, hh, qf);          // S^T tile: rows = keys, lane = query
            }
        }
    }
    __syncthreads();                          // everyone is done reading K
    ab_stage_any<SPLIT, NKT, 2>(kv, vb, ldv, Lk, tid);  // V into the same buffer (its latency runs under the softmax)

    // softmax over the keys of query l31, exactly as the forward (attention.hip)
    float Lrow = 0.f;
    if (active) {
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const float v = key < Lk ? s[kt][r] * scale : -INFINITY;
                s[kt][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // SPLIT: the forward's own exponential (attention_f16x2.hip: one v_exp_f32 on a fused multiply-add); expf's
                // range reduction and overflow guards were 9 instructions per score, 1 300 of the kernel's 8 400
                const float e = SPLIT ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], 1.44269504f, -mx * 1.44269504f)) : expf(s[kt][r] - mx);
                s[kt][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.f / sum;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] *= inv;
        Lrow = mx + logf(sum);
    }
    // dO fragment and delta_q = sum_d dO O  (loaded only now: with the Q fragment still live the kernel would not fit
    // its 256 registers; the fence keeps the scheduler from hoisting the loads over the softmax)
    AB_FENCE();
    typename AbSel<SPLIT>::Frag dof;
    ab_load_frag(dof, dO + qrow * lddo + head * 64, hh, do_scale);      // SPLIT: dO 2^e is what is split (the call's own scale)
    float delta = ab_frag_dot(dO + qrow * lddo + head * 64, O + qrow * ldo + head * 64, hh, SPLIT);   // (fp32 values, L1-hot)
    delta += __shfl_xor(delta, 32);
    if (active && hh == 0 && q0 + l31 < Lq) {
        const int lqs = ((Lq + 31) >> 5) << 5;
        float* st = stats + ((size_t)b * heads + head) * lqs + q0 + l31;
        st[0] = Lrow;
        st[(size_t)gridDim.y * heads * lqs] = delta;      // (in dO's own units: the kv kernel applies do_scale itself)
    }
    if constexpr (SPLIT) delta *= do_scale;
    __syncthreads();                          // V is in LDS
    float ds_max = 0.f;                       // SPLIT: max |dS| of this lane's values, taken as they are produced
    if (active) {
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            f32x16 dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) dp[r] = 0.f;
            ab_rows_times_reg_any<SPLIT, NKT>(dp, kv, kt, l31, hh, dof);                // dP^T tile = V dO^T
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[kt][r] = scale * s[kt][r] * (dp[r] - delta);                           // dS^T (0 for masked keys: P = 0)
                if constexpr (SPLIT) ds_max = fmaxf(ds_max, fabsf(s[kt][r]));            // (fmaxf drops a NaN)
            }
            AB_FENCE();                       // one dP tile live at a time
        }
        if constexpr (SPLIT) {
            if (amax != nullptr) {            // saturation monitor: the operand that carries the loss scale into the split
                const float m = ab_wave_max(ab_frag_absmax(dof));
                // non-negative floats order like their bit patterns; the plain load may be stale, but only ever too SMALL
                if (lane == 0 && __float_as_uint(m) > *(volatile unsigned*)amax) atomicMax(amax, __float_as_uint(m));
            }
        }
    }
    // SPLIT: dS is normalised per wave before its split (header).  The two factors are computed in STRAIGHT-LINE code from a
    // readfirstlane, so that they are scalar registers: a vector register across the staging and the dQ loop below is one more
    // than this kernel has (it runs at the 256-register cap; `active` is wave-uniform but the compiler cannot know).
    float ds_dn = 1.f;
    if constexpr (SPLIT) {
        float ds_up;
        ab_norm_scale(ab_wave_max(ds_max), ds_up, ds_dn);
        if (active) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kt][r] *= ds_up;                 // in place: no second copy of the tiles
        }
        AB_FENCE();
    }
    __syncthreads();
    ab_stage_any<SPLIT, NKT, 2>(kv, kb, ldk, Lk, tid);  // K again
    __syncthreads();
    if (active) {
        f32x16 o0, o1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) ab_reg_times_rows_any<SPLIT, NKT>(o0, o1, s[kt], kv, kt, l31, hh);   // dQ = dS K
        ab_store_tile(dQ + (size_t)b * Lq * lddq + head * 64, lddq, q0, Lq, o0, o1, l31, hh, ds_dn * do_inv);
    }
}

template <int NQT, bool SPLIT>
__global__ __launch_bounds__(AB_NT, 2) void ds_attn_bwd_kv_kernel(const float* __restrict__ Q, int ldq, const float* __restrict__ Kp,
                                                                 int ldk, const float* __restrict__ Vp, int ldv,
                                                                 const float* __restrict__ dO, int lddo, float* __restrict__ dK, int lddk,
                                                                 float* __restrict__ dV, int lddv, const float* __restrict__ stats, int Lq,
                                                                 int Lk, int heads, float scale, float do_scale, float do_inv) {
    extern __shared__ __attribute__((aligned(16))) float qs[];  // [NQT*32][AB_LD] operand rows, then L[NQT*32], delta[NQT*32]
    float* Ls = qs + ab_buf_floats<SPLIT>(NQT);
    float* Ds = Ls + NQT * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int head = blockIdx.x % heads, grp = blockIdx.x / heads, b = blockIdx.y;
    const int k0 = (grp * AB_WAVES + wave) * 32;
    const bool active = k0 < Lk;             // wave-uniform
    const float* qb = Q + (size_t)b * Lq * ldq + head * 64;
    const float* dob = dO + (size_t)b * Lq * lddo + head * 64;
    int kr = k0 + l31;
    const bool key_ok = kr < Lk;
    if (kr >= Lk) kr = Lk - 1;
    const size_t krow = (size_t)b * Lk + kr;

    ab_stage_any<SPLIT, NQT, 4>(qs, qb, ldq, Lq, tid);
    {
        const int lqs = ((Lq + 31) >> 5) << 5;
        const float* st = stats + ((size_t)b * heads + head) * lqs;
        for (int i = tid; i < NQT * 32; i += AB_NT) {
            Ls[i] = i < Lq ? (SPLIT ? st[i] * 1.44269504f : st[i]) : 0.f;
            Ds[i] = i < Lq ? st[(size_t)gridDim.y * heads * lqs + i] * do_scale : 0.f;       // delta in the scaled dO's units
        }
    }
    f32x16 t[NQT];
    {
        typename AbSel<SPLIT>::Frag kf;
        ab_load_frag(kf, Kp + krow * ldk + head * 64, hh);
        __syncthreads();
        if (active) {
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) t[qt][r] = 0.f;
                ab_rows_times_reg_any<SPLIT, NQT>(t[qt], qs, qt, l31, hh, kf);          // S tile: rows = queries, lane = key
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    // P[q][key]  (SPLIT: Ls holds L_q log2(e), see its load above)
                    const float pe = SPLIT ? __builtin_amdgcn_exp2f(__builtin_fmaf(t[qt][r], scale * 1.44269504f, -Ls[q]))
                                           : expf(t[qt][r] * scale - Ls[q]);
                    t[qt][r] = (q < Lq && key_ok) ? pe : 0.f;
                }
                AB_FENCE();      // (one query tile at a time: without expf's branches the scheduler overlaps tiles and spills)
            }
        }
    }
    __syncthreads();                          // everyone is done reading Q
    ab_stage_any<SPLIT, NQT, 2>(qs, dob, lddo, Lq, tid, do_scale);       // SPLIT: dO 2^e is what is split
    __syncthreads();
    if (active) {
        f32x16 o0, o1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt)
            ab_reg_times_rows_any<SPLIT, NQT>(o0, o1, t[qt], qs, qt, l31, hh, SPLIT ? AB_PSCALE : 1.f);   // dV = P^T dO
        ab_store_tile(dV + (size_t)b * Lk * lddv + head * 64, lddv, k0, Lk, o0, o1, l31, hh, (SPLIT ? 1.f / AB_PSCALE : 1.f) * do_inv);
        typename AbSel<SPLIT>::Frag vf;
        ab_load_frag(vf, Vp + krow * ldv + head * 64, hh);
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
            f32x16 dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) dp[r] = 0.f;
            ab_rows_times_reg_any<SPLIT, NQT>(dp, qs, qt, l31, hh, vf);                 // dP tile = dO V^T
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                t[qt][r] = scale * t[qt][r] * (dp[r] - Ds[q]);          // dS (0 where P = 0)
            }
            AB_FENCE();
        }
    }
    float ds_dn = 1.f;                        // SPLIT: dS of ALL query tiles feeds this wave's dK: one normalisation for them
    if constexpr (SPLIT) {                    // (straight-line code, scalar factors: see the q kernel)
        float m = 0.f;
        if (active) {
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(t[qt][r]));
        }
        float ds_up;
        ab_norm_scale(ab_wave_max(m), ds_up, ds_dn);
        if (active) {
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) t[qt][r] *= ds_up;                 // in place: no second copy of the tiles
        }
        AB_FENCE();
    }
    __syncthreads();
    ab_stage_any<SPLIT, NQT, 2>(qs, qb, ldq, Lq, tid);   // Q again
    __syncthreads();
    if (active) {
        f32x16 o0, o1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) ab_reg_times_rows_any<SPLIT, NQT>(o0, o1, t[qt], qs, qt, l31, hh);   // dK = dS^T Q
        ab_store_tile(dK + (size_t)b * Lk * lddk + head * 64, lddk, k0, Lk, o0, o1, l31, hh, ds_dn * do_inv);
    }
}

template <typename KernelT>
static int ab_set_lds(KernelT kernel, size_t lds, DsOnce& done) {
    if (!done.need()) return 0;
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
        ds_set_error("attention backward: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return -2;
    }
    done.done();
    return 0;
}

// Q / O / dO / dQ: [B*Lq][ld] (head h at columns h*64..), K / V / dK / dV: [B*Lk][ld]; any row strides (column ranges of
// fused projections are addressed in place).  stats: 2 * B * heads * ceil32(Lq) floats of workspace.
template <bool SPLIT>
static int ab_launch(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o, int ldo,
                     const float* d_o, int lddo, float* dq, int lddq, float* dk, int lddk, float* dv, int lddv, float* stats, int B,
                     int heads, int Lq, int Lk, float scale, hipStream_t stream, float* amax = nullptr, float do_scale = 1.f) {
    DS_CHECK_ARG(do_scale > 0.f && (SPLIT || do_scale == 1.f), "do_scale: a positive power of two (f16x2 form only)");
    const float do_inv = 1.f / do_scale;
    DS_CHECK_ARG(q && k && v && o && d_o && dq && dk && dv && stats, "null pointer");
    DS_CHECK_ARG(B > 0 && heads > 0 && Lq > 0 && Lk > 0 && Lq <= 288 && Lk <= 288, "at most 288 queries / keys are supported");
    DS_CHECK_ARG(((ldq | ldk | ldv | ldo | lddo) & 3) == 0, "leading dims of the inputs must be multiples of 4");
    DS_CHECK_ARG((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)d_o) & 15) == 0, "16-byte aligned inputs");
    const int qgroups = ((Lq + 31) / 32 + AB_WAVES - 1) / AB_WAVES, kgroups = ((Lk + 31) / 32 + AB_WAVES - 1) / AB_WAVES;
    static DsOnce a3, a9, akv;
    if (Lk <= 96) {
        const size_t lds = ab_buf_floats<SPLIT>(3) * sizeof(float);
        if (ab_set_lds(ds_attn_bwd_q_kernel<3, SPLIT>, lds, a3)) return -2;
        hipLaunchKernelGGL((ds_attn_bwd_q_kernel<3, SPLIT>), dim3(qgroups * heads, B), dim3(AB_NT), lds, stream, q, ldq, k, ldk, v, ldv, o,
                           ldo, d_o, lddo, dq, lddq, stats, Lq, Lk, heads, scale, (unsigned*)amax, do_scale, do_inv);
    } else {
        const size_t lds = ab_buf_floats<SPLIT>(9) * sizeof(float);
        if (ab_set_lds(ds_attn_bwd_q_kernel<9, SPLIT>, lds, a9)) return -2;
        hipLaunchKernelGGL((ds_attn_bwd_q_kernel<9, SPLIT>), dim3(qgroups * heads, B), dim3(AB_NT), lds, stream, q, ldq, k, ldk, v, ldv, o,
                           ldo, d_o, lddo, dq, lddq, stats, Lq, Lk, heads, scale, (unsigned*)amax, do_scale, do_inv);
    }
    DS_CHECK_LAUNCH();
    {
        const size_t lds = (ab_buf_floats<SPLIT>(9) + 2 * 9 * 32) * sizeof(float);
        if (ab_set_lds(ds_attn_bwd_kv_kernel<9, SPLIT>, lds, akv)) return -2;
        hipLaunchKernelGGL((ds_attn_bwd_kv_kernel<9, SPLIT>), dim3(kgroups * heads, B), dim3(AB_NT), lds, stream, q, ldq, k, ldk, v, ldv,
                           d_o, lddo, dk, lddk, dv, lddv, stats, Lq, Lk, heads, scale, do_scale, do_inv);
    }
    DS_CHECK_LAUNCH();
    return 0;
}

extern "C" int ds_attention_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o, int ldo,
                                const float* d_o, int lddo, float* dq, int lddq, float* dk, int lddk, float* dv, int lddv,
                                float* stats, int B, int heads, int Lq, int Lk, float scale, ds_stream_t stream_) {
    return ab_launch<false>(q, ldq, k, ldk, v, ldv, o, ldo, d_o, lddo, dq, lddq, dk, lddk, dv, lddv, stats, B, heads, Lq, Lk, scale,
                            (hipStream_t)stream_);
}

// the same backward with every tile product on the fp16 matrix cores (3-pass split, fp32-class; |dO| must stay below 65504:
// the training step's loss scale sees to it)
extern "C" int ds_attention_bwd_f16x2(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o,
                                      int ldo, const float* d_o, int lddo, float* dq, int lddq, float* dk, int lddk, float* dv,
                                      int lddv, float* stats, int B, int heads, int Lq, int Lk, float scale, ds_stream_t stream_) {
    return ab_launch<true>(q, ldq, k, ldk, v, ldv, o, ldo, d_o, lddo, dq, lddq, dk, lddk, dv, lddv, stats, B, heads, Lq, Lk, scale,
                           (hipStream_t)stream_);
}

// ... and with the saturation monitor of the training step folded in: *amax (a float the caller zeroed at some point; it is
// only ever raised) takes max |dO * do_scale|: the operand that carries the step's loss scale into the kernels' fp16 splits --
// no separate ds_amax pass over dO.  do_scale: a power of two, this CALL's own scale on top of the loss scale (the training
// step calibrates one per attention, like one per linear): dO * do_scale is what is split, 1 / do_scale goes into the stores.
extern "C" int ds_attention_bwd_f16x2_mon(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o,
                                          int ldo, const float* d_o, int lddo, float* dq, int lddq, float* dk, int lddk, float* dv,
                                          int lddv, float* stats, int B, int heads, int Lq, int Lk, float scale, float do_scale,
                                          float* amax, ds_stream_t stream_) {
    DS_CHECK_ARG(amax, "null monitor pointer");
    return ab_launch<true>(q, ldq, k, ldk, v, ldv, o, ldo, d_o, lddo, dq, lddq, dk, lddk, dv, lddv, stats, B, heads, Lq, Lk, scale,
                           (hipStream_t)stream_, amax, do_scale);
}
