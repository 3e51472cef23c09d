#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(uint16_t* out) {
  __shared__ uint16_t lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  int lane = threadIdx.x;
  // each lane supplies the address of 4 contiguous b16 (8 bytes): lane*4 elements
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + lane * 4));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)r[j];
}
int main() {
  uint16_t* d; hipMalloc(&d, 256 * 2);
  hipLaunchKernelGGL(probe, 1, 64, 0, 0, d);
  uint16_t h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  return 0;
}
