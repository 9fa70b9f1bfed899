// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds 16-bit values = their own element index.
// Each lane supplies address = its own choice; we dump what every lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(uint16_t* out, int mode) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int lane = threadIdx.x;
    // mode 0: lane l -> byte address l*8 (each lane points at 4 consecutive elements: 4l..4l+3)
    // mode 1: row-major [16 rows][64 cols] bf16 (128 B rows): lane l -> row (l&15), col-group (l>>4)*4
    uint32_t addr;
    if (mode == 0) addr = lane * 8;
    else addr = (lane & 15) * 128 + (lane >> 4) * 8;
    typedef __attribute__((address_space(3))) char lc;
    uint32_t base = (uint32_t)(uintptr_t)(lc*)lds;
    uint32_t a = base + addr;
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    out[lane * 4 + 0] = v.x & 0xffff; out[lane * 4 + 1] = v.x >> 16;
    out[lane * 4 + 2] = v.y & 0xffff; out[lane * 4 + 3] = v.y >> 16;
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int mode = 0; mode < 2; ++mode) {
        probe<<<1, 64>>>(d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    }
    return 0;
}
