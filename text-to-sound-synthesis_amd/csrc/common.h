// Shared device/host declarations for the Diffsound gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "diffsound_hip.h"  // public enums + C ABI prototypes (include/)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DS_WAVE 64

// round-trip through IEEE half (RNE): emulates the reference's fp16 CLIP activations
// (sound_synthesis/modeling/modules/clip/model.py:432 convert_weights) on fp32 storage
__device__ __forceinline__ float ds_r16(float x) { return (float)(_Float16)x; }

// the f16x2 split  a = hi + lo  (gemm_f16x2.hip): identical code in every producer and consumer, so a value
// split by a producing kernel is bit-identical to the split the GEMM loader would have computed itself
__device__ __forceinline__ _Float16 ds_split_hi(float a) {
    return (_Float16)__builtin_amdgcn_fmed3f(a, -65504.f, 65504.f);  // saturate instead of overflowing to inf
}
__device__ __forceinline__ _Float16 ds_split_lo(float a, _Float16 hi) {
    // contraction off: when `a` is a bare product of the caller (GELU2's v * sigmoid(..) under a compile-time-true branch),
    // hipcc's default -ffp-contract=fast would fuse it into fma(v, s, -hi) -- the residual of the UNROUNDED product, a
    // different lo plane than every other producer writes (found by the bit-identity tests, round 3)
#pragma clang fp contract(off)
    return (_Float16)__builtin_amdgcn_fmed3f(a - (float)hi, -65504.f, 65504.f);  // a - hi is exact in fp32
}

// hi | lo << 16 of the same split in one 32-bit word (the staging format of the per-sample GEMM epilogues): the second
// conversion and the packing are one v_cvt_pk_f16_f32 (its low half re-derives hi from the same clamped value)
__device__ __forceinline__ unsigned ds_split_pack(float a) {
#pragma clang fp contract(off)
    typedef float ds_f2 __attribute__((ext_vector_type(2)));
    typedef _Float16 ds_h2 __attribute__((ext_vector_type(2)));
    const float ac = __builtin_amdgcn_fmed3f(a, -65504.f, 65504.f);
    const _Float16 hi = (_Float16)ac;
    const ds_f2 pr = {ac, __builtin_amdgcn_fmed3f(a - (float)hi, -65504.f, 65504.f)};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(pr, ds_h2));
}

// GELU2 (x * sigmoid(1.702 x), transformer_utils.py:111-115) for the f16x2 GEMM epilogues: v_exp_f32 + v_rcp_f32
// (~1 ulp each) instead of expf + IEEE division -- a tenth of the instructions, which matters where the epilogue is
// not hidden behind another workgroup's MFMAs (gemm_f16x2_ps.hip).  ONE definition for every f16x2 program, so their
// outputs stay bit-identical.  exp2 overflow (v << 0) gives 1 / inf = 0 -> -0, the limit value.
__device__ __forceinline__ float ds_gelu2_fast(float v) {
    return v * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.45546696f * v));   // 1.702 * log2(e)
}

// "Packed split planes": the HBM layout of every pre-split f16x2 GEMM operand (activations written by the
// ds_*_split producers and the GEMM's c_split epilogue; weights packed once by _lib.split_f16x2(packed=True)).
// For X[R][K], K % 32 == 0, each of the two fp16 planes is  [ceil(R/16)][K/32][16 rows][4 chunks][8 halves]:
// a 16-row x 32-k tile is one contiguous KB that already is the GEMM's LDS image (16-byte chunk c of row r sits
// at chunk position c ^ ((r >> 2) & 3), the bank swizzle of the fragment reads), so one global_load_lds_dwordx4
// per wave moves it verbatim: full-cacheline requests, no address math, no staging registers.
__device__ __forceinline__ size_t ds_packed_off(int row, int col, int ktiles) {
    return ((size_t)(row >> 4) * ktiles + (col >> 5)) * 512 + (row & 15) * 32 +
           ((((col >> 3) & 3) ^ ((row >> 2) & 3)) << 3) + (col & 7);
}

// "Attention-ready" operands (attention_f16x2.hip): what the QKV / cross-Q GEMM epilogue and ds_attn_pack_kv write
// so that the attention kernel stages K and V^T by LDS-DMA and reads Q fragments with plain 16-byte loads.
//   Q     : two fp16 planes [B][heads][Lq][64]
//   K,V^T : per (sample, head) one image  K hi | K lo | V^T hi | V^T lo,  each nkey*64 halves, = the kernel's LDS
//           layout: K[key][64] with 16-byte chunks swizzled by (key>>1)&7; V^T[d][nkey] in natural key order with
//           16-byte chunks (8 keys) swizzled by (d>>2)&3.  Rows of keys >= Lk must be zero.
// The kernel feeds MFMA row i of a 32-key tile with key ds_attn_pi(i) (bits 2 and 3 swapped), which makes the score
// registers of a lane hold 8 consecutive keys per k-step -- so the V^T operand is a plain transpose.
__device__ __forceinline__ int ds_attn_k_off(int key, int d) {
    return key * 64 + ((((d >> 3) ^ ((key >> 1) & 7))) << 3) + (d & 7);
}
__device__ __forceinline__ int ds_attn_vt_off(int key, int d, int nkey) {
    return d * nkey + (((key >> 3) ^ ((d >> 2) & 3)) << 3) + (key & 7);
}
__device__ __forceinline__ int ds_attn_pi(int i) { return (i & 0x13) | (((i >> 3) & 1) << 2) | (((i >> 2) & 1) << 3); }

// ---- error plumbing (C ABI returns int; message kept per thread) -------------------------
void ds_set_error(const char* fmt, ...);
#define DS_CHECK_ARG(cond, msg)                                   \
    do {                                                          \
        if (!(cond)) {                                            \
            ds_set_error("%s: %s", __func__, msg);                \
            return -1;                                            \
        }                                                         \
    } while (0)
#define DS_CHECK_LAUNCH()                                                        \
    do {                                                                         \
        hipError_t e_ = hipGetLastError();                                       \
        if (e_ != hipSuccess) {                                                  \
            ds_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
            return -2;                                                           \
        }                                                                        \
    } while (0)

// Per-DEVICE "done once" flag for launch-side caches (hipFuncSetAttribute(MaxDynamicSharedMemorySize), CU counts,
// occupancy): a function-local `static DsOnce once;` is consulted with the CURRENT device, so a process that drives several
// GPUs sets the attribute on each of them (a plain static bool left the 140-161 KB LDS kernels un-launchable on the second
// device); the bit mask is atomic, a repeated set from two threads is harmless.
#include <atomic>
struct DsOnce {
    std::atomic<unsigned long long> mask{0};
    static int dev() {
        int d = 0;
        (void)hipGetDevice(&d);
        return d & 63;
    }
    bool need() const { return !((mask.load(std::memory_order_acquire) >> dev()) & 1ull); }
    void done() { mask.fetch_or(1ull << dev(), std::memory_order_release); }
};

// ---- gather-GEMM parameter block ----------------------------------------------------------
// C[m][n] = store( act( sum_k A(m,k) * W[n][k] + bias[n] ) + R[m][n] )
// A(m,k) is produced by one of the loaders below; W is always K-contiguous ([N][K], nn.Linear
// layout, conv kernels repacked to [Cout][tap][Cin]).

struct GemmParams {
    const float* A;     // activation base
    const float* W;     // [groups][N][K]
    const float* bias;  // [N] or null
    const float* R;     // residual, same addressing as C (row-major only) or null
    float* C;
    int M, N, K;        // per group
    int lda, ldw, ldc, ldr;     // ldw: row stride of W (>= K)
    int groups;                 // blockIdx.y; A/W/C advance by the strides below
    long long a_gstride, w_gstride, c_gstride;
    int pro, act, store;
    int f16_round;              // 1: outputs (and GELU2 intermediates) are rounded to the fp16 grid
    long long w3_plane;         // split kernels: W = 3 bf16 / 2 fp16 planes of [N][ldw], this many elements apart
    float out_scale;            // f16x2 kernel: 2^-s undoing the weight pre-scale
    // f16x2 kernel, store == DS_STORE_ATTN: column n = which*heads*64 + head*64 + d (which: 0 Q, 1 K, 2 V), row =
    // sample*rows_per_sample + pos; C = the Q planes (attn_qplane halves apart), attn_kv = the K / V^T images.
    // The tile is staged through LDS and leaves as 16-byte stores (8 d of a row for Q / K, 8 keys of a d for V^T).
    void* attn_kv;
    int attn_heads, attn_nkey, row_off;   // row_off: rows of this (sub-)problem start at this absolute row
    long long attn_qplane;
    int a_split, c_split;       // f16x2 kernel: A (and then W too) is given / C is written as packed split planes
    long long a_plane, c_plane; //   (ds_packed_off); plane strides in halves; lda / ldc = the row length K of that matrix
    // prologue: per-(sample, channel) affine  a' = a*pro_scale[b*Cin+c] + pro_shift[b*Cin+c]
    const float* pro_scale;
    const float* pro_shift;
    int rows_per_sample;        // rows of A belonging to one sample (dense prologue / BATCH_T store)
    // conv geometry (channels-last)
    int Cin;                    // channels per tap
    int H, W_;                 // conv2d: output height/width;  conv1d/convT: H unused, W_ = T (output / phase rows)
    int up;                     // conv2d: 1 => input is (H/2, W/2), nearest-upsampled on the fly
    int taps, dil;              // conv1d: taps, dilation (reflect padding)
    int ct_r, ct_p, ct_tin;     // convT1d: stride r, padding p, input length
    // f16x2 conv kernel, dense loader: a SECOND row source for the k-tiles k >= k_split (A2 [M][lda2], column k - k_split);
    // the prologue applies to the first source only.  (ds_melgan_resblock_tail: [LReLU(h) | x] x [W2 | Ws]^T in one GEMM)
    const float* A2;
    int k_split, lda2;
};

// padded-row forms used by the native denoiser driver (api.hip): Lp / logits_rows >= L rows per sample
int ds_embed_rows(const int64_t* tokens, const float* emb, const float* pos, float* out, int B, int L, int Lp, int D,
                  ds_stream_t stream);                                                                  // norm.hip
// u == nullptr: the uniforms are drawn in the kernel from the Philox stream of (seed; gids[b], call) (sampler.hip)
int ds_sample_tail_rows(const float* logits, int logits_rows, const int64_t* xt, const int64_t* t, const float* u,
                        const float* sched, int64_t* out_tokens, float* dbg_log_pred, float* dbg_trunc, float* dbg_post,
                        int B, int L, int K, int T, int initial, float trunc_r, int trunc_k, ds_stream_t stream,
                        const int64_t* gids = nullptr, unsigned long long seed = 0ull, int call = 0);  // sampler.hip

int ds_launch_gemm(const GemmParams& p, hipStream_t stream, int loader);
int ds_launch_gemm_f16x2(const GemmParams& p, hipStream_t stream);   // gemm_f16x2.hip
int ds_launch_conv2d_f16x2(const GemmParams& p, hipStream_t stream, int loader); // conv_f16x2.hip
