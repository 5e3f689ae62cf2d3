// Counter-based synthetic read generator (bench / tests only; no reference counterpart).
// Byte i of each stream depends only on (seed, i), so any slice can be regenerated on the host
// (pypore_amd/synth.py implements the same arithmetic in numpy) and full-size buffers are created in HBM without H2D.
//   word j = i / 4:  hb = mix64(seed ^ (2j) * M), hq = mix64(seed ^ (2j+1) * M), 16 bits of hash per byte
//   base : v < 7 -> 'N' (1.07e-4); v < 26874 -> G/C (P = 0.41) by v&1; else A/T by v&1
//   qual : s = sum of the four nibbles of v (Irwin-Hall, mean 30, sd 9.2); q = clamp(4s/9 - 1, 1, 60) (mean 12, sd 4.1)
#include "ppqc_internal.hpp"

namespace ppqc {

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ uint32_t synth_bases_word(uint64_t h) {
    uint32_t w = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const uint32_t v = (uint32_t)(h >> (16 * b)) & 0xFFFFu;
        uint32_t c;
        if (v < 7u) c = 'N';
        else if (v < 26874u) c = (v & 1u) ? 'G' : 'C';
        else c = (v & 1u) ? 'A' : 'T';
        w |= c << (8 * b);
    }
    return w;
}

__device__ __forceinline__ uint32_t synth_quals_word(uint64_t h) {
    uint32_t w = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const uint32_t v = (uint32_t)(h >> (16 * b)) & 0xFFFFu;
        const uint32_t s = (v & 15u) + ((v >> 4) & 15u) + ((v >> 8) & 15u) + (v >> 12);
        int q = (int)(4u * s / 9u) - 1;
        q = q < 1 ? 1 : (q > 60 ? 60 : q);
        w |= (uint32_t)(33 + q) << (8 * b);
    }
    return w;
}

// first and n are multiples of 4 except possibly the tail: one thread per 4-byte word, byte-granular tail.
__global__ __launch_bounds__(256) void synth_kernel(uint8_t* __restrict__ bases, uint8_t* __restrict__ quals,
                                                    uint64_t first, uint64_t n, uint64_t seed) {
    const uint64_t M = 0xD1342543DE82EF95ull;
    const uint64_t w0 = first / 4, w1 = (first + n + 3) / 4;
    for (uint64_t j = w0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < w1;
         j += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t wb = synth_bases_word(mix64(seed ^ ((2 * j) * M)));
        const uint32_t wq = synth_quals_word(mix64(seed ^ ((2 * j + 1) * M)));
        const uint64_t i0 = 4 * j;
        if (i0 >= first && i0 + 4 <= first + n && ((uintptr_t)(bases + (i0 - first)) & 3) == 0 &&
            ((uintptr_t)(quals + (i0 - first)) & 3) == 0) {
            *reinterpret_cast<uint32_t*>(bases + (i0 - first)) = wb;
            *reinterpret_cast<uint32_t*>(quals + (i0 - first)) = wq;
        } else {
            for (int b = 0; b < 4; ++b) {
                const uint64_t i = i0 + b;
                if (i >= first && i < first + n) {
                    bases[i - first] = (uint8_t)(wb >> (8 * b));
                    quals[i - first] = (uint8_t)(wq >> (8 * b));
                }
            }
        }
    }
}

hipError_t launch_synth(uint8_t* bases, uint8_t* quals, uint64_t first, uint64_t n, uint64_t seed, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint64_t words = (n + 3) / 4 + 1;
    uint64_t blocks = (words + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(synth_kernel, dim3((unsigned)blocks), dim3(256), 0, s, bases, quals, first, n, seed);
    return hipGetLastError();
}

}  // namespace ppqc
