// Probe: semantics of ds_read_b64_tr_b16 on gfx950. Fills LDS with element index (u16), each lane reads
// with address = lane's own 8-byte aligned address of a [rows][16] u16 tile; prints what each lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(uint16_t* out, int rowstride_elems) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    int l = threadIdx.x;
    int g = l >> 4, i = l & 15;
    // lane i of group g supplies address of row (i>>2) + 4*g, col chunk (i&3)*4
    int row = (i >> 2) + 4 * g;
    unsigned addr = (unsigned)((row * rowstride_elems + (i & 3) * 4) * 2);
    unsigned base = (unsigned)(uintptr_t)lds;  // LDS address (low 32 bits of generic? use offset)
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr + (unsigned)(size_t)((__attribute__((address_space(3))) uint16_t*)lds)) : "memory");
    out[l * 4 + 0] = v.x & 0xffff; out[l * 4 + 1] = v.x >> 16; out[l * 4 + 2] = v.y & 0xffff; out[l * 4 + 3] = v.y >> 16;
    (void)base;
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    for (int rs : {16, 64}) {
        k<<<1, 64>>>(d, rs);
        uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("rowstride=%d\n", rs);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d: ", l);
            for (int j = 0; j < 4; ++j) printf("(r%d,c%d) ", h[l * 4 + j] / rs, h[l * 4 + j] % rs);
            printf("\n");
        }
    }
    return 0;
}
