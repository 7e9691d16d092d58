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
};

int ds_launch_gemm(const GemmParams& p, hipStream_t stream, int loader);
int ds_launch_gemm_bf16x3(const GemmParams& p, hipStream_t stream);  // gemm_bf16x3.hip
int ds_launch_gemm_f16x2(const GemmParams& p, hipStream_t stream);   // gemm_f16x2.hip
