// mx_probe.hip — what v_mfma_scale_f32_16x16x128_f8f6f4 does with its scale operands, measured (one wave, one instruction per test).
//   hipcc --offload-arch=gfx950 -O2 -o build/mx_probe tools/probe/mx_probe.hip && build/mx_probe
// All data bytes are e4m3 1.0 (0x38) or 0, so D[i][j] = sum over the k of (row i, column j) of 2^(eA - 127) * 2^(eB - 127): with distinct
// exponents per (lane group, scale byte) the SET of scale bytes the hardware used can be read off the bits of D.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int OA, int OB>
__global__ void k_probe(const int* A, const int* B, const int* SA, const int* SB, float* D) {
  const int l = threadIdx.x;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = A[l * 8 + i]; b[i] = B[l * 8 + i]; }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, OA, SA[l], OB, SB[l]);
  for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];      // row (l >> 4) * 4 + r, column l & 15
}

static int *dA, *dB, *dSA, *dSB; static float* dD;
static int hA[512], hB[512], hSA[64], hSB[64]; static float hD[256];
template <int OA, int OB> static void run() {
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipMemcpy(dSA, hSA, sizeof hSA, hipMemcpyHostToDevice); hipMemcpy(dSB, hSB, sizeof hSB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((k_probe<OA, OB>), dim3(1), dim3(64), 0, 0, dA, dB, dSA, dSB, dD);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
}
static float Dat(int row, int col) { return hD[((row >> 2) * 16 + col) * 4 + (row & 3)]; }
static void ones(int* X) { for (int i = 0; i < 512; ++i) X[i] = 0x38383838; }
static void unit(int* S) { for (int i = 0; i < 64; ++i) S[i] = 0x7f7f7f7f; }
// scale pattern: lane l, byte b -> exponent (l / 16) * 4 + b
static void pattern(int* S) { for (int l = 0; l < 64; ++l) { unsigned v = 0; for (int b = 0; b < 4; ++b) v |= (unsigned)(127 + (l / 16) * 4 + b) << (8 * b); S[l] = (int)v; } }
static void show(const char* what, int row, int col, double per_term) {
  const double v = Dat(row, col) / per_term;
  printf("  %-46s D[%d][%d] / %g = %10.1f = bits", what, row, col, per_term, v);
  unsigned long long m = (unsigned long long)llround(v);
  for (int e = 0; e < 24; ++e) if (m >> e & 1) printf(" %d(lane group %d, byte %d)", e, e / 4, e % 4);
  printf("\n");
}
int main() {
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dSA, sizeof hSA); hipMalloc(&dSB, sizeof hSB); hipMalloc(&dD, sizeof hD);
  printf("T0: all ones, unit scales: D[0][0] = %g (expect 128), D[5][9] = %g\n", (ones(hA), ones(hB), unit(hSA), unit(hSB), run<0, 0>(), Dat(0, 0)), Dat(5, 9));
  printf("T1: B scale = pattern (exponent = lane group * 4 + byte), A scale = 1: which bytes feed column j\n");
  ones(hA); ones(hB); unit(hSA); pattern(hSB);
  run<0, 0>(); show("opsel B = 0", 0, 0, 32); show("opsel B = 0", 7, 3, 32);
  run<0, 1>(); show("opsel B = 1", 0, 0, 32);
  run<0, 2>(); show("opsel B = 2", 0, 0, 32);
  run<0, 3>(); show("opsel B = 3", 0, 0, 32);
  printf("T2: A scale = pattern, B scale = 1: which bytes feed row i\n");
  unit(hSB); pattern(hSA);
  run<0, 0>(); show("opsel A = 0", 0, 0, 32); show("opsel A = 0", 7, 3, 32);
  run<1, 0>(); show("opsel A = 1", 0, 0, 32);
  run<2, 0>(); show("opsel A = 2", 0, 0, 32);
  run<3, 0>(); show("opsel A = 3", 0, 0, 32);
  printf("T3: only ONE lane's B scale differs (+20 on every byte): which outputs move (lists D != 128)\n");
  for (int L : {0, 5, 17, 40, 63}) {
    ones(hA); ones(hB); unit(hSA); unit(hSB); hSB[L] = (int)0x93939393u;
    run<0, 0>();
    printf("  lane %2d:", L);
    int n = 0;
    for (int r = 0; r < 16; ++r) for (int cc = 0; cc < 16; ++cc) if (Dat(r, cc) != 128.f) { if (n < 6) printf(" D[%d][%d]=%g", r, cc, Dat(r, cc)); ++n; }
    printf("  (%d outputs)\n", n);
  }
  printf("T4: B data = 1.0 only in dword d of every lane (A all ones), B scale = pattern, opsel 0: which scale a lane's dword d gets\n");
  for (int d = 0; d < 8; ++d) {
    ones(hA); memset(hB, 0, sizeof hB); for (int l = 0; l < 64; ++l) hB[l * 8 + d] = 0x38383838;
    unit(hSA); pattern(hSB); run<0, 0>();
    char w[64]; snprintf(w, sizeof w, "dword %d", d); show(w, 0, 0, 4);
  }
  printf("T5: B data = 1.0 only in lanes of group g (A all ones), B scale = pattern, opsel 0\n");
  for (int g = 0; g < 4; ++g) {
    ones(hA); memset(hB, 0, sizeof hB); for (int l = g * 16; l < g * 16 + 16; ++l) for (int d = 0; d < 8; ++d) hB[l * 8 + d] = 0x38383838;
    unit(hSA); pattern(hSB); run<0, 0>();
    char w[64]; snprintf(w, sizeof w, "lane group %d", g); show(w, 0, 0, 32);
  }
  printf("T6: upper 4 dwords of A and B zero (the G = 16 form), B scale pattern: D[0][0] / 16\n");
  ones(hA); ones(hB); for (int l = 0; l < 64; ++l) for (int d = 4; d < 8; ++d) hA[l * 8 + d] = hB[l * 8 + d] = 0;
  unit(hSA); pattern(hSB); run<0, 0>(); show("upper halves zero", 0, 0, 16);
  return 0;
}
