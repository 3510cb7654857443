// Probe (MI355X): what does ds_read_b64_tr_b16 deliver?  LDS holds half-precision values equal to their own half index; lane l
// reads from byte address 8 * perm(l) (one 64-bit = 4-half segment per lane); prints, per lane, the 4 half indices it received.
//   hipcc --offload-arch=gfx950 -O3 -o bin/tr_read tr_read.hip && bin/tr_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__global__ void probe(const int* __restrict__ addr_of_lane, unsigned short* __restrict__ out) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;   // value = its own half index (as raw 16 bits)
    __syncthreads();
    const unsigned base = (unsigned)(size_t)lds;   // LDS byte address of the array
    const unsigned a = base + (unsigned)addr_of_lane[threadIdx.x];
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    out[threadIdx.x * 4 + 0] = (unsigned short)(v.x & 0xFFFF);
    out[threadIdx.x * 4 + 1] = (unsigned short)(v.x >> 16);
    out[threadIdx.x * 4 + 2] = (unsigned short)(v.y & 0xFFFF);
    out[threadIdx.x * 4 + 3] = (unsigned short)(v.y >> 16);
}

int main() {
    int h_addr[64];
    int* d_addr; unsigned short* d_out; unsigned short h_out[256];
    CK(hipMalloc(&d_addr, sizeof(h_addr))); CK(hipMalloc(&d_out, sizeof(h_out)));
    // experiment 1: lane l reads segment l (bytes 8 l .. 8 l + 7: halves 4 l .. 4 l + 3)
    // experiment 2: lane l reads segment at row (l % 16) of a [16][64-half] matrix, column block (l / 16): address = 128 * (l % 16) + 8 * (l / 16)
    for (int exp = 0; exp < 2; ++exp) {
        for (int l = 0; l < 64; ++l) h_addr[l] = exp == 0 ? 8 * l : 128 * (l % 16) + 8 * (l / 16);
        CK(hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        CK(hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost));
        printf("experiment %d (lane: address -> four half indices received)\n", exp);
        for (int l = 0; l < 64; ++l)
            printf("  lane %2d: byte %4d (half %4d) -> %4d %4d %4d %4d\n", l, h_addr[l], h_addr[l] / 2, h_out[4 * l], h_out[4 * l + 1], h_out[4 * l + 2], h_out[4 * l + 3]);
    }
    return 0;
}
