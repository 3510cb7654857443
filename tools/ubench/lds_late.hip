// Micro-benchmark (MI355X): does ds_read_b128 data ever land in its destination registers AFTER s_waitcnt lgkmcnt(0) has let the wave
// go on -- with one / two workgroups per CU hammering LDS?  Per iteration: a burst of ds_read_b128 (contention), the load under test into
// v[4:7], s_waitcnt lgkmcnt(0), [the ten-MFMA block reading v[4:7] as A], VALU overwrite of v[4:7] with a marker, a long drain, then v[4:7]
// must still hold the marker and the accumulators must equal those of a run with a long sleep behind the wait.
//   hipcc --offload-arch=gfx950 -O3 -o bin/lds_late lds_late.hip && bin/lds_late
#include <hip/hip_runtime.h>
#include <cstdio>
#define BLOCK                                                                  \
    "v_mfma_f32_32x32x16_f16 v[16:31], v[4:7], v[8:11], 0\n"                   \
    "v_mfma_f32_32x32x16_f16 v[32:47], v[4:7], v[8:11], 0\n"                   \
    "v_mfma_f32_32x32x16_f16 v[48:63], v[4:7], v[8:11], 0\n"                   \
    "v_mfma_f32_32x32x16_f16 v[64:79], v[4:7], v[8:11], 0\n"                   \
    "v_mfma_f32_32x32x16_f16 v[16:31], v[4:7], v[8:11], v[16:31]\n"            \
    "v_mfma_f32_32x32x16_f16 v[32:47], v[4:7], v[8:11], v[32:47]\n"            \
    "v_mfma_f32_32x32x16_f16 v[48:63], v[4:7], v[8:11], v[48:63]\n"            \
    "v_mfma_f32_32x32x16_f16 v[64:79], v[4:7], v[8:11], v[64:79]\n"            \
    "v_mfma_f32_32x32x16_f16 v[48:63], v[4:7], v[8:11], v[48:63]\n"            \
    "v_mfma_f32_32x32x16_f16 v[64:79], v[4:7], v[8:11], v[64:79]\n"
#define BURST "ds_read_b128 v[80:83], %8\nds_read_b128 v[84:87], %8 offset:4096\nds_read_b128 v[88:91], %8 offset:8192\nds_read_b128 v[92:95], %8 offset:12288\n" \
              "ds_read_b128 v[96:99], %8 offset:16384\nds_read_b128 v[100:103], %8 offset:20480\n"
#define MARK "v_mov_b32 v4, 1.0\nv_mov_b32 v5, 1.0\nv_mov_b32 v6, 1.0\nv_mov_b32 v7, 1.0\n"
#define DRAIN "s_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_sleep 4\n"
#define CLOB "v4","v5","v6","v7","v8","v9","v10","v11","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
    "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63", \
    "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79", \
    "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","memory"
__global__ void __launch_bounds__(256, 2) k(unsigned* bad, unsigned* badq, unsigned* badm, int iters, size_t lds_floats) {
    extern __shared__ float lds[];
    for (size_t i = threadIdx.x; i < 32768 / 4 + 4096; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0x38003800u + (unsigned)(i & 0xff);   // f16 pairs ~0.5
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned addr = (unsigned)(size_t)lds + (unsigned)threadIdx.x * 16u;
    unsigned mism = 0, q[4] = {0, 0, 0, 0}, mark = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned b0 = 0x38003800u + ((unsigned)(lane * 7 + it) & 0x3ff);
        float r[2][4]; unsigned m[2][4];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            float o0, o1, o2, o3; unsigned a0, a1, a2, a3;
            if (pass == 0)
                asm volatile("v_mov_b32 v8, %9\nv_mov_b32 v9, %9\nv_mov_b32 v10, %9\nv_mov_b32 v11, %9\n" BURST "ds_read_b128 v[4:7], %8 offset:24576\ns_waitcnt lgkmcnt(0)\ns_sleep 8\n"
                             BLOCK MARK DRAIN
                             "v_mov_b32 %0, v16\nv_mov_b32 %1, v47\nv_mov_b32 %2, v50\nv_mov_b32 %3, v79\nv_mov_b32 %4, v4\nv_mov_b32 %5, v5\nv_mov_b32 %6, v6\nv_mov_b32 %7, v7\n"
                             : "=v"(o0), "=v"(o1), "=v"(o2), "=v"(o3), "=v"(a0), "=v"(a1), "=&v"(a2), "=&v"(a3) : "v"(addr), "v"(b0) : CLOB);
            else
                asm volatile("v_mov_b32 v8, %9\nv_mov_b32 v9, %9\nv_mov_b32 v10, %9\nv_mov_b32 v11, %9\n" BURST "ds_read_b128 v[4:7], %8 offset:24576\ns_waitcnt lgkmcnt(0)\n"
                             BLOCK MARK DRAIN
                             "v_mov_b32 %0, v16\nv_mov_b32 %1, v47\nv_mov_b32 %2, v50\nv_mov_b32 %3, v79\nv_mov_b32 %4, v4\nv_mov_b32 %5, v5\nv_mov_b32 %6, v6\nv_mov_b32 %7, v7\n"
                             : "=v"(o0), "=v"(o1), "=v"(o2), "=v"(o3), "=v"(a0), "=v"(a1), "=&v"(a2), "=&v"(a3) : "v"(addr), "v"(b0) : CLOB);
            r[pass][0] = o0; r[pass][1] = o1; r[pass][2] = o2; r[pass][3] = o3; m[pass][0] = a0; m[pass][1] = a1; m[pass][2] = a2; m[pass][3] = a3;
        }
        bool ne = false, mk = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) { ne |= (__float_as_uint(r[0][j]) != __float_as_uint(r[1][j])); mk |= (m[1][j] != 0x3f800000u) | (m[0][j] != 0x3f800000u); }
        if (ne) { ++mism; ++q[lane >> 4]; }
        if (mk) ++mark;
    }
    if (mism) { atomicAdd(bad, mism); for (int j = 0; j < 4; ++j) if (q[j]) atomicAdd(badq + j, q[j]); }
    if (mark) atomicAdd(badm, mark);
}
int main() {
    unsigned *bad, *badq, *badm;
    hipMalloc(&bad, 4); hipMalloc(&badq, 16); hipMalloc(&badm, 4);
    const int iters = 2000;
    for (int two = 0; two < 2; ++two) {
        const size_t lds = two ? 70 * 1024 : 120 * 1024;
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipMemset(bad, 0, 4); hipMemset(badq, 0, 16); hipMemset(badm, 0, 4);
        hipLaunchKernelGGL(k, dim3(two ? 512 : 256), dim3(256), lds, 0, bad, badq, badm, iters, lds / 4);
        hipError_t e = hipDeviceSynchronize();
        unsigned h, hq[4], hm; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(hq, badq, 16, hipMemcpyDeviceToHost); hipMemcpy(&hm, badm, 4, hipMemcpyDeviceToHost);
        printf("A operand by ds_read_b128, lgkmcnt(0), ten MFMAs, marker written over A: %s per CU: %u lane-results differ from the slept run (by lane quarter: %u %u %u %u), marker lost in %u lane-iterations, of %u  [%s]\n",
               two ? "two workgroups" : "one workgroup", h, hq[0], hq[1], hq[2], hq[3], hm, (two ? 512u : 256u) * 256u * iters, hipGetErrorString(e));
    }
    return 0;
}
