// probe: semantics of __builtin_amdgcn_fdot2_f32_bf16 (v_dot2c_f32_bf16) on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned *a, const unsigned *b, float *o) {
  unsigned x = a[threadIdx.x], y = b[threadIdx.x];
  o[threadIdx.x] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, x), __builtin_bit_cast(bf2, y), 100.f, false);
}
static unsigned short bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }
int main() {
  float lo[4] = {1.f, 2.f, 0.5f, -3.f}, hi[4] = {4.f, 8.f, 0.25f, 7.f};
  unsigned ha[64], hb[64];
  for (int i = 0; i < 64; ++i) { ha[i] = bf(lo[i % 4]) | ((unsigned)bf(hi[i % 4]) << 16); hb[i] = bf(hi[(i + 1) % 4]) | ((unsigned)bf(lo[(i + 2) % 4]) << 16); }
  unsigned *da, *db; float *d;
  hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&d, 256);
  hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(da, db, d);
  float ho[64]; hipMemcpy(ho, d, 256, hipMemcpyDeviceToHost);
  for (int i = 0; i < 4; ++i) {
    float e = lo[i % 4] * hi[(i + 1) % 4] + hi[i % 4] * lo[(i + 2) % 4] + 100.f;
    printf("lane %d: got %g expect %g\n", i, ho[i], e);
  }
  return 0;
}
