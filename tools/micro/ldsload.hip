// Developer microbenchmark (not part of the product): do LDS-direct global loads (global_load_lds_dword, data to LDS at M0 + lane * 4,
// no VGPR destination) work from inline asm on gfx950, with sc1 (past the caches)?  Build: hipcc --offload-arch=gfx950 -O3 -o ldsload ldsload.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const unsigned long long* src, unsigned long long* dst) {
    __shared__ unsigned int buf[2 * 64];
    const unsigned lane = threadIdx.x;
    const unsigned int* p = reinterpret_cast<const unsigned int*>(src + lane);
    const unsigned ldsbase = (unsigned)(uintptr_t)buf;
    unsigned keep;
    // (the instruction offset is added to the memory address AND to the LDS address: the high dword takes its own pointer)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\tglobal_load_lds_dword %2, off sc1\n\ts_add_u32 m0, m0, 256\n\t"
                 "global_load_lds_dword %3, off sc1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(ldsbase), "v"(p), "v"(p + 1) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    dst[lane] = ((unsigned long long)buf[64 + lane] << 32) | buf[lane];
}
int main() {
    unsigned long long h[64], *s, *d;
    for (int i = 0; i < 64; ++i) h[i] = 0x1111000000000000ull * (i % 7) + 0xabcdef00ull + i;
    hipMalloc(&s, sizeof(h)); hipMalloc(&d, sizeof(h));
    hipMemcpy(s, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, d);
    unsigned long long o[64];
    hipMemcpy(o, d, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) bad += o[i] != h[i];
    printf("lds-direct loads: %d mismatches of 64\n", bad);
    return bad != 0;
}
