// Which SIMD does wave w of a 512-thread workgroup land on?  (HW_REG_HW_ID of gfx9: wave_id [3:0], simd_id [5:4], cu_id [11:8], se_id [15:13].)
// hipcc --offload-arch=gfx950 -O2 tools/probes/wave_simd_map.hip -o /tmp/wave_simd_map && /tmp/wave_simd_map
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = id;
}
int main() {
  for (int threads : {256, 512}) {
    unsigned* d; hipMalloc(&d, 4096 * 4);
    k<<<4, threads>>>(d);
    unsigned h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 4; ++b) {
      printf("threads %d wg %d:", threads, b);
      for (int w = 0; w < threads / 64; ++w) { unsigned x = h[b * (threads / 64) + w]; printf("  w%d simd %u cu %u slot %u", w, (x >> 4) & 3, (x >> 8) & 15, x & 15); }
      printf("\n");
    }
    hipFree(d);
  }
  return 0;
}
