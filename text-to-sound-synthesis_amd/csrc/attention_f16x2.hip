// Fused multi-head attention (head dim 64) on the fp16 matrix cores with the 2-way fp16 split of
// gemm_f16x2.hip: fp32-class results at 3 fp16-MFMA passes per contraction instead of the fp32 MFMA's
// 16x slower rate.  Same contract as ds_attention (attention.hip); replaces FullAttention /
// CrossAttention cores (transformer_utils.py:43-58, :91-109) when the denoiser runs in f16x2 mode.
//
// Round 4: STREAMED form.  Rounds 1-3 kept all 288 key slots of a head resident (72 KB of LDS per workgroup, K then V^T in
// the same buffer) and the whole 32 x 288 score tile of a wave in registers (196 VGPRs): 1.5 waves per SIMD, phases of
// ~100 MFMAs / ~600 VALU instructions that nothing on the SIMD could overlap with (profiles/r02g_attn_kernel_timeline.txt).
// Now a workgroup (3 waves = 3 query tiles of 32) walks the keys in CHUNKS of 96 (three 32-key tiles) with a running row
// maximum / row sum (the online-softmax recurrence): 48 score registers instead of 144, 24 KB of K + 24 KB of V^T per
// workgroup instead of 72 KB -- three workgroups per CU at <= 168 registers, whose short MFMA and VALU phases interleave.
//
//   scores  S^T[key][q] = sum_d K[key][d] Q[q][d]:  k = k0 + k1, q = q0 + q1 (fp16), MFMAs k1q0 + k0q1 + k0q0.
//           A operand = K rows from LDS (two fp16 planes, 128-byte rows, 16-byte chunks XOR-swizzled by
//           (key>>1)&7 -> conflict-free ds_read_b128), B operand = the wave's own Q rows in registers.
//           Transposed scores: lane&31 is the query, the keys of a tile are spread over the 16 accumulator
//           registers, so the softmax statistics are in-lane + one cross-half shuffle.
//           MFMA row i of a 32-key tile is fed with key pi(i) (bits 2 and 3 of i swapped, ds_attn_pi), so that
//           accumulator register r of lane half h holds key (r&3) + 4((r>>2)&1) + 8h + 16(r>>3): the 8 registers
//           of a k-step are 8 CONSECUTIVE keys.
//   output  O^T[d][q] = sum_key V^T[d][key] P[q][key]  (TRANSPOSED as well since round 4: A operand = V^T rows from LDS,
//           one ds_read_b128 per plane; B operand = the P registers, split p0 + p1 and packed: k-step s of a 32-key
//           tile takes registers 8s..8s+7 = keys 16s + 8*half + e).  MFMAs v0p1 + v1p0 + v0p0.  lane&31 is the query in
//           the scores AND in the output, so the rescale by exp(m_old - m_new) when the running maximum moves and the
//           final division by the row sum are in-lane multiplies (the round-3 form needed 16 cross-lane reads for the
//           normalisation alone), and a lane owns 4 consecutive d per register quad: 8-byte staging writes.
// LDS per workgroup: K chunk [2 planes][96 keys][64 d] + V^T chunk [2 planes][4 d-blocks][3 key tiles][16 d][32 keys]
// (1 KB tiles = what ONE LDS-DMA instruction moves: lane -> (d = lane>>2, 16-byte piece = lane&3)), 48 KB.
//
// READY variant (the denoiser's path): Q arrives as two fp16 planes and K / V^T as ready-made images
// (common.h "attention-ready operands", written by the QKV / cross-Q GEMM epilogues and ds_attn_pack_kv), so
// staging is 1 KB LDS-DMA transfers with no conversion work: chunk c + 1 of K lands under the softmax and the P V of
// chunk c, chunk c + 1 of V^T under the scores of chunk c + 1; two barriers per chunk.
// The generic variant converts fp32 Q / K / V itself (same arithmetic, same bits).
#include "common.h"
#include <type_traits>
// (the LDS-DMA is written out with M0 as its LDS address register; hipcc warns about M0 on a clobber list)
#pragma clang diagnostic ignored "-Winline-asm"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

#define AH_WAVES 3
// The adopted configuration (DESIGN.md section 3 "Attention"; the alternatives -- other chunk sizes and read-ahead depths,
// separate K / V^T buffers, a persistent grid that walks several items per workgroup: all measured, profiles/r04g_*, r04r_* --
// and the in-kernel timing stamps are in tools/probe/attn_probe.patch, applied by the probe builds only).
#define AH_TPC 3                       // 32-key tiles per chunk, many-chunk kernel (self-attention: 288 key slots = 3 chunks of 96)
#define AH_TPC1 3                      // ... of the kernel for <= 96 keys (cross-attention): 3 = one chunk
#define AH_KAHEAD 2                    // k-steps a K fragment is read ahead of its MFMAs; V^T fragments are read one k-step ahead
#define AH_XSHARE 1                    // READY: K and V^T chunks share ONE LDS buffer (4 / 5 workgroups per CU; every next K chunk is waited for)
#define AH_TS 72                       // halves per staged output row (64 + 8: 16-byte aligned, 2-way banks at most)

__device__ __forceinline__ _Float16 ah_hi(float a) { return ds_split_hi(a); }
__device__ __forceinline__ _Float16 ah_lo(float a, _Float16 h) { return ds_split_lo(a, h); }

typedef __attribute__((address_space(1))) const void* ah_gptr;
typedef __attribute__((address_space(3))) void* ah_lptr;
typedef float ah_f2 __attribute__((ext_vector_type(2)));
typedef _Float16 ah_h2 __attribute__((ext_vector_type(2)));
typedef unsigned ah_u4 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int ah_tpc(int nkt) { return nkt <= 3 ? AH_TPC1 : AH_TPC; }

// halves offset of V^T element (d, local key kl) inside one plane of the V^T chunk buffer (TPC key tiles per chunk)
template <int TPC>
__device__ __forceinline__ int ah_vt_off(int d, int kl) {
    return (((d >> 4) * TPC + (kl >> 5)) << 9) + ((d & 15) << 5) + ((((kl >> 3) & 3) ^ ((d >> 2) & 3)) << 3) + (kl & 7);
}

// fp16 split of two probabilities e0, e1 in [0, 1] as packed pairs: p0 = RNE(e) by v_cvt_pk_f16_f32, p1 = RNE(e - p0) by
// v_fma_mix{lo,hi}_f16 (an fp32 fma p0 * -1 + e, exact, rounded once to fp16) -- the same bits as
// (_Float16)(e - (float)(_Float16)e) in 1.5 instead of 2.5 instructions per value (hipcc: cvt_pk + 2 cvt_f32 + pk_add + cvt_pk)
__device__ __forceinline__ void ah_split2(float e0, float e1, unsigned& p0, unsigned& p1) {
    const ah_f2 pr = {e0, e1};
    p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, ah_h2));
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p0), "v"(e0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(p0), "v"(e1));
    p1 = r;
}

// One ITEM = (sample b, head, query group grp): three query tiles of 32 rows (one per wave) against all keys of the head.
// Items are numbered b * (groups * heads) + grp * heads + head; workgroup w works items w, w + gridDim.x, ... (one item per
// workgroup unless the READY kernel is launched persistent).
template <int NKT, bool READY>
__global__ __launch_bounds__(AH_WAVES * 64, READY ? 3 : 2) void ds_attn_f16x2_kernel(const float* __restrict__ Q, int ldq,
                                                                        const float* __restrict__ Kp, int ldk,
                                                                        const float* __restrict__ Vp, int ldv,
                                                                        float* __restrict__ O, int ldo, int Lq, int Lk,
                                                                        int heads, float scale, long long o_plane,
                                                                        long long q_plane, int groups, int n_items) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int TPC = ah_tpc(NKT);
    constexpr int CK = 32 * TPC;                 // keys per chunk
    constexpr int PL = CK * 64;                  // halves per plane of a K or V^T chunk
    constexpr int PP = 4 * TPC;                  // 1 KB LDS-DMA pieces per plane of a chunk
    constexpr int NCH = (NKT + TPC - 1) / TPC;   // chunks the images hold (self: 288 key slots, cross: 96)
    constexpr int NKEY = NKT * 32;
    constexpr int KPL = NKEY * 64;   // halves per K plane of the image ([key][64 d]) = per V^T plane ([d][NKEY])
    constexpr bool XS = READY && AH_XSHARE;
    _Float16* Kb = (_Float16*)smem_raw;            // [2][CK][64]
    _Float16* Vb = Kb + (XS ? 0 : 2 * PL);         // [2][4 d-blocks][TPC][16][32]
    constexpr int VB_BYTES = XS ? 0 : 4 * PL;      // byte offset of the V^T chunk buffer

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int nch = (Lk + CK - 1) / CK;          // chunks that hold keys (<= NCH by the launch rule)
    const int per_b = groups * heads;

    // READY: a (sample, head)'s image  K hi | K lo | V^T hi | V^T lo  (bytes), NKEY/4 KB per operand
    const unsigned vlane = (lane >> 2) * (NKEY * 2) + (lane & 3) * 16;   // V^T gather: lane -> (d row, 16-byte piece) of a tile
    const unsigned klane = lane * 16;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw);
    // wave-uniform image base (SGPRs) of (b, head)
    auto image_of = [&](int b_, int head_) {
        const unsigned long long a_ = (unsigned long long)((const unsigned char*)Kp + ((size_t)b_ * heads + head_) * (size_t)(8 * KPL));
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a_ >> 32)) << 32) |
               (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a_);
    };
    // 2 * PP one-KB transfers per operand and chunk, dealt round-robin to the waves: transfer p = plane (p / PP), piece
    // (p % PP); the pieces of key tiles the image does not have (last chunk, NKT % TPC != 0) are skipped.
    // Written out as  SGPR base + ONE per-lane byte offset  (the form of gemm_f16x2_ps.hip): through the builtin hipcc
    // keeps a 64-bit per-lane pointer per transfer alive across the chunk loop -- 32 VGPRs this kernel does not have.
    constexpr int NXF = (2 * PP + AH_WAVES - 1) / AH_WAVES;
#define AH_DMA(vofs_, base_, lds_)                                                                   \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(vofs_), "s"(base_), "s"(lds_) : "memory", "m0")
    auto issue_k = [&](unsigned long long img_s, int c) {
#pragma unroll
        for (int j = 0; j < NXF; ++j) {
            const int p = j * AH_WAVES + wave, pl = p >= PP ? 1 : 0, pp = p - PP * pl;   // pp / 4 = key tile
            if (p < 2 * PP && c * TPC + (pp >> 2) < NKT) {
                const unsigned long long src = img_s + (unsigned long long)(pl * (2 * KPL) + c * (CK * 128) + pp * 1024);
                const unsigned dst = lds0 + pl * (2 * PL) + pp * 1024;
                AH_DMA(klane, src, dst);
            }
        }
    };
    auto issue_v = [&](unsigned long long img_s, int c) {
#pragma unroll
        for (int j = 0; j < NXF; ++j) {
            const int p = j * AH_WAVES + wave, pl = p >= PP ? 1 : 0, pp = p - PP * pl;   // pp = d-block * TPC + key tile
            const int db = pp / TPC, kt = pp - TPC * db;
            if (p < 2 * PP && c * TPC + kt < NKT) {
                const unsigned long long src = img_s + (unsigned long long)((2 + pl) * (2 * KPL) + db * (16 * NKEY * 2) + (c * PP + kt * 4) * 16);
                const unsigned dst = lds0 + VB_BYTES + pl * (2 * PL) + pp * 1024;
                AH_DMA(vlane, src, dst);
            }
        }
    };
    // generic variant: fp32 rows -> fp16 planes in the same LDS layouts (keys >= Lk are zero rows)
    auto stage_k = [&](const float* kb, int c) {
        constexpr int NIT = (CK * 16 + AH_WAVES * 64 - 1) / (AH_WAVES * 64);   // CK keys x 16 float4 over 192 threads
        f32x4 v[NIT];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {      // loads first (NIT round trips in flight)
            const int f = tid + u * (AH_WAVES * 64), row = f >> 4, c4 = f & 15;
            v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < CK && c * CK + row < Lk) v[u] = *(const f32x4*)(kb + (size_t)(c * CK + row) * ldk + c4 * 4);
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int f = tid + u * (AH_WAVES * 64), row = f >> 4, c4 = f & 15;
            if (row >= CK) break;
            h4 s0, s1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s0[e] = ah_hi(v[u][e]);
                s1[e] = ah_lo(v[u][e], s0[e]);
            }
            _Float16* dst = Kb + row * 64 + ((c4 >> 1) ^ ((row >> 1) & 7)) * 8 + (c4 & 1) * 4;
            *(h4*)dst = s0;
            *(h4*)(dst + PL) = s1;
        }
    };
    auto stage_v = [&](const float* vb, int c) {
        // work item = (4 consecutive keys, 4 consecutive d): CK / 4 x 16 items
        for (int f = tid; f < CK * 4; f += AH_WAVES * 64) {
            const int kg = f >> 4, c4 = f & 15;
            f32x4 v[4];
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                const int key = c * CK + 4 * kg + kx;
                v[kx] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (key < Lk) v[kx] = *(const f32x4*)(vb + (size_t)key * ldv + c4 * 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h4 s0, s1;
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) {
                    s0[kx] = ah_hi(v[kx][j]);
                    s1[kx] = ah_lo(v[kx][j], s0[kx]);
                }
                _Float16* dst = Vb + ah_vt_off<TPC>(c4 * 4 + j, 4 * kg);
                *(h4*)dst = s0;
                *(h4*)(dst + PL) = s1;
            }
        }
    };
    // ---- Q operand: lane (q = l31, half hh) keeps Q[q][16 ks + 8 hh + 0..7], ks = 0..3, both planes ----
    h8 q_hi[4], q_lo[4];
    auto load_q = [&](int b_, int head_, int q0_) {
        int qr = q0_ + l31;
        if (qr >= Lq) qr = Lq - 1;
        if constexpr (READY) {
            const _Float16* qp = (const _Float16*)Q + (((size_t)b_ * heads + head_) * Lq + qr) * 64 + 8 * hh;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                q_hi[ks] = *(const h8*)(qp + 16 * ks);
                q_lo[ks] = *(const h8*)(qp + q_plane + 16 * ks);
            }
        } else {
            const float* qp = Q + ((size_t)b_ * Lq + qr) * ldq + head_ * 64 + 8 * hh;
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
    };

    const float sl = scale * 1.4426950408889634f;   // exp(x) = 2^(x log2 e)
    float m_run, l_run;                             // running row maximum (raw scores) / row sum of this lane's keys
    f32x16 o[2];                                    // O^T: register r = d (r&3) + 8 (r>>2) + 4 hh (+ 32 for o[1]), lane&31 = query
    f32x16 s[TPC];                                  // S^T of the current chunk, then its exponentials

    // scores of chunk c.  FULL: all CK keys of the chunk are < Lk -- straight-line code, no masks, no tests; the K
    // fragments are read two k-steps ahead of their MFMAs and every k-step is its own scheduling region (left to itself
    // hipcc hoists every fragment read of the chunk to the top: 40+ registers of spills at the 168 this kernel may use)
    auto kfrag = [&](int it, h8& k0, h8& k1) {       // it = key tile * 4 + k-step
        const int kl = (it >> 2) * 32 + ds_attn_pi(l31);
        const _Float16* kr = Kb + kl * 64 + ((((2 * (it & 3) + hh) ^ ((kl >> 1) & 7))) << 3);
        k0 = *(const h8*)kr;
        k1 = *(const h8*)(kr + PL);
    };
    auto scores = [&](auto full_tag, int c) {
        constexpr bool FULL = decltype(full_tag)::value;
        if constexpr (FULL) {
            constexpr int NIT = 4 * TPC;
            constexpr int KA = AH_KAHEAD;
            h8 kf[KA + 1][2];
#pragma unroll
            for (int it = 0; it < KA; ++it) kfrag(it, kf[it][0], kf[it][1]);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                if (it + KA < NIT) kfrag(it + KA, kf[(it + KA) % (KA + 1)][0], kf[(it + KA) % (KA + 1)][1]);
                const int kt = it >> 2, ks = it & 3;
                f32x16 a;
                if (ks == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) a[r] = 0.f;
                } else {
                    a = s[kt];
                }
                a = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[it % (KA + 1)][1], q_hi[ks], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[it % (KA + 1)][0], q_lo[ks], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[it % (KA + 1)][0], q_hi[ks], a, 0, 0, 0);
                s[kt] = a;
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int kt = 0; kt < TPC; ++kt) {
                const int kbase = c * CK + kt * 32;
                if (kbase < Lk) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        h8 k0, k1;
                        kfrag(kt * 4 + ks, k0, k1);
                        f32x16 a = s[kt];
                        a = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, q_hi[ks], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, q_lo[ks], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, q_hi[ks], a, 0, 0, 0);
                        s[kt] = a;
                    }
                    if (kbase + 32 > Lk) {   // wave-uniform: this tile holds keys >= Lk
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int key = kbase + (r & 3) + 4 * ((r >> 2) & 1) + 8 * hh + 16 * (r >> 3);   // pi(MFMA row)
                            s[kt][r] = key < Lk ? s[kt][r] : -INFINITY;
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[kt][r] = -INFINITY;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // online softmax: new running maximum, rescale what is accumulated, exponentials of this chunk (first chunk: m_run =
    // -inf makes the rescale factor 2^-inf = 0 on accumulators that are still 0)
    auto softmax = [&]() {
        float mx = m_run;
#pragma unroll
        for (int kt = 0; kt < TPC; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float msl = mx * sl;
        const float alpha = __builtin_amdgcn_exp2f(__builtin_fmaf(m_run, sl, -msl));
        l_run *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
        m_run = mx;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < TPC; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], sl, -msl));   // masked: 2^-inf = 0
                s[kt][r] = e;
                sum += e;
            }
        l_run += sum;
    };
    // O^T += V^T P^T over the chunk's 16-key k-steps (FULL: all of them, the V^T fragments read one k-step ahead; else
    // those that hold keys < Lk).  One scheduling region per k-step, as in the scores.
    const _Float16* vr = Vb + ((l31 >> 4) * TPC << 9) + ((l31 & 15) << 5);   // d = l31; d = 32 + l31 is 2 d-blocks on
    auto vfrag = [&](int step, h8 (&v)[4]) {           // step = key tile * 2 + half tile
        const int off = ((step >> 1) << 9) + (((((step & 1) * 2 + hh) ^ ((l31 >> 2) & 3))) << 3);
        v[0] = *(const h8*)(vr + off);
        v[1] = *(const h8*)(vr + PL + off);
        v[2] = *(const h8*)(vr + 2 * TPC * 512 + off);
        v[3] = *(const h8*)(vr + 2 * TPC * 512 + PL + off);
    };
    auto pv_step = [&](int step, const h8 (&v)[4]) {
        const int kt = step >> 1, st = step & 1;
        ah_u4 p0, p1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned a0, a1;
            ah_split2(s[kt][8 * st + 2 * e], s[kt][8 * st + 2 * e + 1], a0, a1);
            p0[e] = a0;
            p1[e] = a1;
        }
        const h8 ph = __builtin_bit_cast(h8, p0), pl = __builtin_bit_cast(h8, p1);
        o[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v[0], pl, o[0], 0, 0, 0);
        o[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v[2], pl, o[1], 0, 0, 0);
        o[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v[1], ph, o[0], 0, 0, 0);
        o[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v[3], ph, o[1], 0, 0, 0);
        o[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v[0], ph, o[0], 0, 0, 0);
        o[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v[2], ph, o[1], 0, 0, 0);
    };
    auto pv = [&](auto full_tag, int c) {
        constexpr bool FULL = decltype(full_tag)::value;
        constexpr int NS = 2 * TPC;
        if constexpr (FULL) {
            h8 vf[2][4];
            vfrag(0, vf[0]);
#pragma unroll
            for (int step = 0; step < NS; ++step) {
                if (step + 1 < NS) vfrag(step + 1, vf[(step + 1) & 1]);
                pv_step(step, vf[step & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int step = 0; step < NS; ++step) {
                if (c * CK + step * 16 < Lk) {   // wave-uniform: k-steps past the last key are skipped
                    h8 v[4];
                    vfrag(step, v);
                    pv_step(step, v);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    const std::true_type yes{};
    const std::false_type no{};

    // (one (sample, head, query group) item per workgroup)
    const int item = blockIdx.x;
    if (item < n_items) {
        const int b = item / per_b, rem = item - b * per_b;
        const int grp = rem / heads, head = rem - grp * heads;
        const int q0 = (grp * AH_WAVES + wave) * 32;
        const bool active = q0 < Lq;   // wave-uniform
        const unsigned long long img_s = image_of(b, head);
        const float* kb = Kp + (size_t)b * Lk * ldk + head * 64;
        const float* vb = Vp + (size_t)b * Lk * ldv + head * 64;

        {
            // ---- first chunk of K (and V^T), the wave's Q rows ----
            if constexpr (READY) {
                issue_k(img_s, 0);
                if constexpr (!XS) issue_v(img_s, 0);
            }
            load_q(b, head, q0);
            if constexpr (READY) {
                // (explicit: hipcc does not reliably add the vmcnt(0) an in-flight LDS-DMA needs before a barrier, see gemm_f16x2.hip)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                stage_k(kb, 0);
                stage_v(vb, 0);
            }
            __syncthreads();   // chunk 0 of K (and V^T) is in LDS
        }
        m_run = -INFINITY;
        l_run = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }

        // chunks 0 .. nch-2 hold CK keys < Lk each (FULL); ONE copy of that code in a rolled loop, then the last chunk
        int c = 0;
        if constexpr (NCH > 1)
#pragma nounroll
        for (; c + 1 < nch; ++c) {
            if (active) scores(yes, c);
            if constexpr (XS) {
                __syncthreads();            // every wave is done with the K chunk: V^T replaces it, landing under the softmax
                issue_v(img_s, c);
            } else if constexpr (READY) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this chunk's V^T (issued one barrier ago)
                __syncthreads();            // every wave is done with the K chunk; the V^T chunk is visible
                issue_k(img_s, c + 1);
            }
            if (active) softmax();
            if constexpr (XS) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();            // V^T is in LDS
            }
            if (active) pv(yes, c);
            if constexpr (XS) {
                __syncthreads();            // every wave is done with the V^T chunk: the next K chunk replaces it
                issue_k(img_s, c + 1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            } else if constexpr (READY) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next K chunk (issued under the softmax)
                __syncthreads();            // every wave is done with the V^T chunk; the next K chunk is visible
                issue_v(img_s, c + 1);
            } else {
                __syncthreads();
                stage_k(kb, c + 1);
                stage_v(vb, c + 1);
                __syncthreads();
            }
        }
        {   // last chunk: keys c * CK .. Lk - 1
            if (active) scores(no, c);
            if constexpr (XS) {
                __syncthreads();
                issue_v(img_s, c);
            } else if constexpr (READY) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();            // every wave is done with the K chunk (and with its Q rows); V^T is visible
            }
            if (active) softmax();
            if constexpr (XS) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            if (active) pv(no, c);
        }
        // ---- normalise (in-lane: the row sum of query l31 is this lane's + the other half's) ----
        if (active) {
            const float inv = 1.f / (l_run + __shfl_xor(l_run, 32));
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] *= inv; o[1][r] *= inv; }
        }
        if (o_plane > 0) {
            // packed split planes for the f16x2 projection GEMM (K = ldo): each wave stages its 32 x 64 tile, one plane after
            // the other, in the now free V^T buffer (8-byte writes: 4 consecutive d of a query) and stores 16-byte chunks (8 d
            // of one row).
            __syncthreads();                                   // every wave is done reading K / V^T
            if (active) {
                _Float16* T = (_Float16*)(smem_raw + VB_BYTES) + wave * (32 * AH_TS);      // [32 rows][AH_TS]
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            h4 a;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const _Float16 hi_ = ds_split_hi(o[dh][4 * g + e]);
                                a[e] = pl == 0 ? hi_ : ds_split_lo(o[dh][4 * g + e], hi_);
                            }
                            *(h4*)(T + l31 * AH_TS + dh * 32 + 8 * g + 4 * hh) = a;
                        }
                    // (LDS operations of one wave complete in order: no barrier between its own writes and reads)
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int cc = lane + 64 * it, rl = cc >> 3, ch = cc & 7;
                        const int qr = q0 + rl;
                        if (qr < Lq) {
                            const h8 val = *(const h8*)(T + rl * AH_TS + ch * 8);
                            *(h8*)((_Float16*)O + (size_t)pl * o_plane + ds_packed_off(b * Lq + qr, head * 64 + ch * 8, ldo >> 5)) = val;
                        }
                    }
                }
            }
        } else {
            if (active) {
                const int qr = q0 + l31;
                if (qr < Lq) {
                    float* orow = O + ((size_t)b * Lq + qr) * ldo + head * 64 + 4 * hh;
#pragma unroll
                    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            *(f32x4*)(orow + dh * 32 + 8 * g) = f32x4{o[dh][4 * g], o[dh][4 * g + 1], o[dh][4 * g + 2], o[dh][4 * g + 3]};
                }
            }
        }
    }
}

template <int NKT, bool READY>
static int attn_f16x2_launch_n(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                               int B, int heads, int Lq, int Lk, float scale, long long o_plane, long long q_plane,
                               hipStream_t stream) {
    constexpr int TPC = ah_tpc(NKT);
    constexpr bool XS = READY && AH_XSHARE;
    const int qtiles = (Lq + 31) / 32;
    const int groups = (qtiles + AH_WAVES - 1) / AH_WAVES;
    const long long items = (long long)groups * heads * B;
    DS_CHECK_ARG(items < (1ll << 30), "too many (sample, head, query group) items");
    // K chunk + V^T chunk (one shared buffer in the XS build); the staged output plane (14 KB) lives in the V^T buffer
    constexpr size_t stage_bytes = (size_t)AH_WAVES * 32 * AH_TS * sizeof(unsigned short);
    constexpr size_t pl_bytes = (size_t)2 * TPC * 32 * 64 * sizeof(unsigned short);       // one operand chunk, both planes
    constexpr size_t lds = XS ? (pl_bytes > stage_bytes ? pl_bytes : stage_bytes) : 2 * pl_bytes;
    static_assert(pl_bytes >= stage_bytes, "the staged output plane fits in the V^T chunk buffer");
    static DsOnce cap_once;
    if (cap_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)ds_attn_f16x2_kernel<NKT, READY>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            ds_set_error("attention_f16x2: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return -2;
        }
        cap_once.done();
    }
    const int grid = (int)items;
    hipLaunchKernelGGL((ds_attn_f16x2_kernel<NKT, READY>), dim3(grid), dim3(AH_WAVES * 64), lds, stream, q, ldq, k, ldk, v, ldv,
                       o, ldo, Lq, Lk, heads, scale, o_plane, q_plane, groups, (int)items);
    DS_CHECK_LAUNCH();
    return 0;
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
    DS_CHECK_ARG(o_plane > 0 || ldo % 4 == 0, "ldo must be a multiple of 4");
    if (Lk <= 96) return attn_f16x2_launch_n<3, READY>(q, ldq, k, ldk, v, ldv, o, ldo, B, heads, Lq, Lk, scale, o_plane, q_plane, stream);
    DS_CHECK_ARG(Lk <= 288, "at most 288 keys are supported");
    return attn_f16x2_launch_n<9, READY>(q, ldq, k, ldk, v, ldv, o, ldo, B, heads, Lq, Lk, scale, o_plane, q_plane, stream);
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
extern "C" int ds_attention_f16x2_ready(const void* qh, long long q_plane, const void* kv_img, void* oh, int ldo, int B,
                                        int heads, int Lq, int Lk, float scale, ds_stream_t stream) {
    DS_CHECK_ARG(ldo % 32 == 0 && ldo >= heads * 64, "packed output needs ldo % 32 == 0");
    DS_CHECK_ARG(q_plane >= (long long)B * heads * Lq * 64 && q_plane % 8 == 0, "Q plane stride");
    DS_CHECK_ARG(((uintptr_t)qh & 15) == 0 && ((uintptr_t)kv_img & 15) == 0, "operands must be 16-byte aligned");
    return attn_f16x2_launch<true>((const float*)qh, 0, (const float*)kv_img, 0, nullptr, 0, (float*)oh, ldo, B, heads,
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
