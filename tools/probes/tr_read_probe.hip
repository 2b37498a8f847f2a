// Probe (gfx950): what ds_read_b64_tr_b16 delivers.  LDS image [64 rows][64 cols] of 16-bit values v = row * 256 + col; every lane of a
// wave supplies the address &img[r0 + q][c0 + 4p] with q = (lane & 15) >> 2, p = lane & 3, (r0, c0) = block of its 16-lane group.
// Prints, per lane, the four 16-bit values received.  Build: hipcc --offload-arch=gfx950 tr_read_probe.hip -o tr_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s4* lp4;
__global__ void k(unsigned short* out)
{
    __shared__ __attribute__((aligned(16))) unsigned short img[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) img[i] = (unsigned short)((i / 64) * 256 + (i % 64));
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int r0 = 8 * g, c0 = 16 * (g & 1);                 // four different blocks
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(img + (r0 + q) * 64 + c0 + 4 * p));
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (unsigned short)v[e];
}
int main()
{
    unsigned short* d; unsigned short h[256];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4, i = lane & 15, r0 = 8 * g, c0 = 16 * (g & 1);
        printf("lane %2d:", lane);
        for (int e = 0; e < 4; ++e) {
            printf(" (r%2d,c%2d)", h[lane * 4 + e] >> 8, h[lane * 4 + e] & 255);
            if (h[lane * 4 + e] != (r0 + e) * 256 + c0 + i) ++bad;          // expectation: element e = row r0+e, column c0 + (lane & 15)
        }
        printf("\n");
    }
    printf("%s: lane i of a 16-lane group receives column i of the block's 4 rows (row q in element q): %d mismatches\n", bad ? "DIFFERENT" : "CONFIRMED", bad);
    return 0;
}
