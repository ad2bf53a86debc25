#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
  unsigned l = threadIdx.x;
  unsigned X = 100 + l, Y = 200 + l;
  u32x2 r = __builtin_amdgcn_permlane16_swap(X, Y, false, false);
  out[l] = r.x; out[64 + l] = r.y;
  u32x2 q = __builtin_amdgcn_permlane32_swap(X, Y, false, false);
  out[128 + l] = q.x; out[192 + l] = q.y;
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); unsigned h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* names[4] = {"p16.x", "p16.y", "p32.x", "p32.y"};
  for (int a = 0; a < 4; ++a) { printf("%s:", names[a]); for (int r = 0; r < 4; ++r) printf(" row%d=%u..%u", r, h[64 * a + 16 * r], h[64 * a + 16 * r + 15]); printf("\n"); }
  return 0;
}
