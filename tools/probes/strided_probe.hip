// Probe: how fast does a wave stream text when each lane owns a contiguous strip (16-byte loads at a stride of 64 or
// 128 bytes between lanes) instead of the coalesced layout?  Decides the lane layout of the one-pass FASTQ kernel.
// hipcc --offload-arch=gfx950 -O3 -o strided_probe strided_probe.hip && ./strided_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int STRIP_VECS, bool NT>   // vectors of 16 bytes a lane owns contiguously; 1 = coalesced
__global__ __launch_bounds__(256) void k(const uint8_t* __restrict__ t, uint64_t n_chunks, uint64_t per_wave, uint32_t* out) {
    const int lane = threadIdx.x & 63;
    const uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    uint32_t acc = 0;
    for (uint64_t c = w * per_wave; c < (w + 1) * per_wave && c < n_chunks; ++c) {
        const uint8_t* base = t + c * 8192;
        u32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // vector i of the lane: strip s = i / STRIP_VECS, j = i % STRIP_VECS
            const int s = i / STRIP_VECS, j = i % STRIP_VECS;
            const uint64_t off = (uint64_t)s * (64 * STRIP_VECS * 16) + (uint64_t)lane * (STRIP_VECS * 16) + (uint64_t)j * 16;
            const u32x4* p = reinterpret_cast<const u32x4*>(base + off);
            v[i] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int SV, bool NT>
static void run(const char* name, const uint8_t* d, uint64_t n, uint32_t* out, int waves) {
    const uint64_t n_chunks = n / 8192;
    const uint64_t per_wave = (n_chunks + waves - 1) / waves;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k<SV, NT>), dim3(waves / 4), dim3(256), 0, 0, d, n_chunks, per_wave, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (it == 2) printf("%-28s waves %6d  %.3f ms  %.0f GB/s\n", name, waves, ms, n / ms / 1e6);
    }
}

int main() {
    const uint64_t n = 2200ull << 20;
    uint8_t* d; uint32_t* out;
    CK(hipMalloc(&d, n)); CK(hipMalloc(&out, 64)); CK(hipMemset(d, 65, n));
    for (int waves : {4096, 8192, 16384}) {
        run<1, true>("coalesced nt", d, n, out, waves);
        run<1, false>("coalesced", d, n, out, waves);
        run<2, false>("strip 32 B", d, n, out, waves);
        run<4, false>("strip 64 B", d, n, out, waves);
        run<8, false>("strip 128 B", d, n, out, waves);
        run<8, true>("strip 128 B nt", d, n, out, waves);
    }
    return 0;
}
