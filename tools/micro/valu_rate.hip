// Issue rate of a few VALU instructions on gfx950 (one wave per SIMD, unrolled independent chains).
// hipcc --offload-arch=gfx950 -O3 -o tools/micro/valu_rate tools/micro/valu_rate.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int OP>
__global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t seed, int iters) {
    uint64_t a[8];
    uint32_t b[8], c = seed + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed * (i + 1) + threadIdx.x; b[i] = seed + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] += (uint64_t)b[i] | ((uint64_t)c << 32);            // 64-bit add
            if (OP == 1) b[i] = b[i] + c + b[(i + 1) & 7];                         // v_add3_u32
            if (OP == 2) b[i] = __builtin_amdgcn_sad_u8(b[i], c, b[(i + 1) & 7]);  // v_sad_u8
            if (OP == 3) b[i] = __popc(b[i] ^ c) + b[(i + 1) & 7];                 // v_bcnt (+xor)
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + b[i];
    if (s == 0x1234567) out[0] = s;
}

template <int OP>
static void run(const char* name, uint64_t* out) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256 * 4), dim3(256), 0, 0, out, 7u, 10);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(256 * 4), dim3(256), 0, 0, out, 7u, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // 1024 blocks x 4 waves over 1024 SIMDs = 4 waves per SIMD; instructions per SIMD = 4 * iters * 8 (* ops per statement)
    printf("%-22s %.3f ms  -> %.2f ns per statement per SIMD (4 waves interleaved)\n", name, ms, ms * 1e6 / (4.0 * iters * 8));
}

int main() {
    uint64_t* out;
    hipMalloc((void**)&out, 64);
    run<0>("u64 add", out);
    run<1>("v_add3_u32", out);
    run<2>("v_sad_u8", out);
    run<3>("xor + bcnt-add", out);
    return 0;
}
