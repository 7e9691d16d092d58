#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
// mode 0: lane l reads the 8 bytes at l * 8 (lane-linear).  mode 1: a [16 rows][32 halves] tile with 64-byte rows: lane l of a
// 16-lane group g reads row 4 g' + (l >> 2) ... (see host)
__global__ void probe(const int* __restrict__ addr, unsigned short* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const unsigned a = (unsigned)(size_t)lds + addr[threadIdx.x];
    s4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main() {
    int* d_addr; unsigned short* d_out;
    hipMalloc(&d_addr, 64 * 4); hipMalloc(&d_out, 64 * 4 * 2);
    for (int mode = 0; mode < 3; ++mode) {
        std::vector<int> a(64);
        for (int l = 0; l < 64; ++l) {
            if (mode == 0) a[l] = l * 8;                                        // lane-linear
            else if (mode == 1) a[l] = ((l & 15) >> 2) * 64 + (l & 3) * 8 + (l >> 4) * 256;   // group g: rows 4g..4g+3 of 64-byte rows, 16 columns
            else a[l] = (l & 15) * 64 + (l >> 4) * 8;                           // 16 rows x (4 col chunks): lane = row, group = chunk
        }
        hipMemcpy(d_addr, a.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        std::vector<unsigned short> o(256);
        hipMemcpy(o.data(), d_out, 512, hipMemcpyDeviceToHost);
        printf("mode %d (values = half index in LDS; lane's own address/2 in brackets)\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d [%4d]: %4d %4d %4d %4d\n", l, a[l] / 2, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
    }
    return 0;
}
