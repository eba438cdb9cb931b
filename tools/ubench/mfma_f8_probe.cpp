// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3) on gfx950: operand lane layout, C layout, where a lane's E8M0 scale applies.
// hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_f8_probe.cpp -o tools/ubench/mfma_f8_probe.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// a[lane][32 bytes], b[lane][32 bytes], sa[lane], sb[lane] (E8M0 in byte `opsel`), out[lane][4]
__global__ void probe(const uint8_t* a, const uint8_t* b, const uint32_t* sa, const uint32_t* sb, float* out, int mode) {
  const int l = threadIdx.x;
  i32x8 av, bv;
  for (int i = 0; i < 8; ++i) { av[i] = ((const int*)(a + l * 32))[i]; bv[i] = ((const int*)(b + l * 32))[i]; }
  f32x4 c = {0, 0, 0, 0};
  if (mode == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, (int)sa[l], 0, (int)sb[l]);
  else if (mode == 1) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 1, (int)sa[l], 2, (int)sb[l]);
  for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
}

static float e4m3(uint8_t v) {  // OCP e4m3fn
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float r = e == 0 ? ldexpf((float)m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
  return s ? -r : r;
}

int main() {
  uint8_t ha[64 * 32], hb[64 * 32];
  uint32_t hsa[64], hsb[64];
  float ho[256];
  uint8_t *da, *db; uint32_t *dsa, *dsb; float* dout;
  hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dsa, sizeof hsa); hipMalloc(&dsb, sizeof hsb); hipMalloc(&dout, sizeof ho);
  auto run = [&](int mode) {
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    hipMemcpy(dsa, hsa, sizeof hsa, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, sizeof hsb, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dout, mode);
    hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
  };
  // (1) random values, unit scales: C[row][col] with row = A lane & 15, col = B lane & 15, sum over (lane group, byte) pairs of equal index
  srand(1);
  for (int i = 0; i < 64 * 32; ++i) { ha[i] = (uint8_t)(rand() % 0x78); if (rand() & 1) ha[i] |= 0x80; hb[i] = (uint8_t)(rand() % 0x78); if (rand() & 1) hb[i] |= 0x80; }
  for (int i = 0; i < 64; ++i) hsa[i] = hsb[i] = 127;
  run(0);
  double worst = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int col = l & 15, row = (l >> 4) * 4 + r;  // C/D map of the 16x16 shapes
      double ref = 0;
      for (int g = 0; g < 4; ++g)
        for (int j = 0; j < 32; ++j) ref += (double)e4m3(ha[(g * 16 + row) * 32 + j]) * e4m3(hb[(g * 16 + col) * 32 + j]);
      worst = fmax(worst, fabs(ref - ho[l * 4 + r]) / (1 + fabs(ref)));
    }
  printf("(1) A = first operand rows, B cols, same (group, byte) pairing, C map col=l&15 row=4*(l>>4)+r: worst rel err %.3g\n", worst);
  for (int range = 0; range < 3; ++range) {  // value ranges: which magnitudes break the exact match?
    const int top = range == 0 ? 0x40 : (range == 1 ? 0x60 : 0x78), bot = range == 2 ? 0x60 : 0x08;
    for (int i = 0; i < 64 * 32; ++i) { ha[i] = (uint8_t)(bot + rand() % (top - bot)); hb[i] = (uint8_t)(bot + rand() % (top - bot)); if (rand() & 1) hb[i] |= 0x80; }
    run(0);
    int bad = 0; double w2 = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        const int col = l & 15, row = (l >> 4) * 4 + r;
        double ref = 0;
        for (int g = 0; g < 4; ++g)
          for (int j = 0; j < 32; ++j) ref += (double)e4m3(ha[(g * 16 + row) * 32 + j]) * e4m3(hb[(g * 16 + col) * 32 + j]);
        const double e = fabs(ref - ho[l * 4 + r]) / (1e-30 + fabs(ref));
        if (e > 1e-5) { if (++bad <= 3) printf("    mismatch C[%d][%d]: device %g, host %g\n", row, col, ho[l * 4 + r], ref); }
        w2 = fmax(w2, e);
      }
    printf("    bytes [0x%02x, 0x%02x): %d of 256 outputs off by > 1e-5 relative (worst %.3g)\n", bot, top, bad, w2);
  }
  // (2) where does lane L's A-scale apply?  A = B = 1.0 everywhere restricted to B lane group gM: overlap[gL][gM]
  for (int which = 0; which < 2; ++which) {
    printf("(2) scale of %s lane (row 3, group gL) doubled; other operand nonzero only in group gM: C[3][5] - 32 (expected 32 on the diagonal if a lane's scale covers its own 32 values):\n", which ? "B" : "A");
    for (int gL = 0; gL < 4; ++gL) {
      for (int gM = 0; gM < 4; ++gM) {
        for (int i = 0; i < 64 * 32; ++i) { ha[i] = 0x38; hb[i] = 0x38; }
        uint8_t* other = which ? ha : hb;
        for (int l = 0; l < 64; ++l) if ((l >> 4) != gM) for (int j = 0; j < 32; ++j) other[l * 32 + j] = 0;
        for (int i = 0; i < 64; ++i) hsa[i] = hsb[i] = 127;
        (which ? hsb : hsa)[gL * 16 + (which ? 5 : 3)] = 128;
        run(0);
        // C[row 3][col 5]: lane with col 5 and row group 0 -> l = 5, r = 3
        printf(" %6.1f", ho[5 * 4 + 3] - 32);
      }
      printf("\n");
    }
  }
  // (3) opsel: byte 1 of scale_a, byte 2 of scale_b
  for (int i = 0; i < 64 * 32; ++i) { ha[i] = 0x38; hb[i] = 0x38; }
  for (int i = 0; i < 64; ++i) { hsa[i] = 127u | (128u << 8) | (127u << 16) | (127u << 24); hsb[i] = 127u | (127u << 8) | (129u << 16) | (127u << 24); }
  run(1);
  printf("(3) opsel_a = 1 (byte 1 = 2^1), opsel_b = 2 (byte 2 = 2^2): C[0][0] = %.1f (128 * 8 = 1024 expected)\n", ho[0]);
  run(0);
  printf("    opsel 0 / 0 with the same registers: C[0][0] = %.1f (128 expected)\n", ho[0]);
  return 0;
}
