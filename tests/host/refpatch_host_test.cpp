// Host build of pl-svo_amd/csrc/align_refpatch.hpp (g++ -ffp-contract=off): the record form of the reference-patch cache (PLSVO_BYTE_CACHE:
// 7 rows of 8 image bytes + the two sub-pixel fractions, rebuilt every iteration by the slot's lane, one patch row at a time) against the
// direct form the precompute writes (four lanes, one patch row each).  Bitwise comparison of all 48 floats of a patch; prints
// "<cases> <mismatching float4s>".
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
struct float4 { float x, y, z, w; };
struct uint4 { uint32_t x, y, z, w; };
struct uint2 { uint32_t x, y; };
#include "align_refpatch.hpp"
using namespace plsvo_hip;

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 32); }

int main(int argc, char** argv) {
  const int cases = argc > 1 ? atoi(argv[1]) : 20000;
  long bad = 0;
  for (int n = 0; n < cases; ++n) {
    unsigned char win[7][8];   // image rows vi-3 .. vi+3, columns ui-3 .. ui+4 (the 8th byte is whatever follows in the image)
    const int mode = n % 5;
    for (int r = 0; r < 7; ++r) for (int c = 0; c < 8; ++c)
      win[r][c] = mode == 0 ? (unsigned char)rnd() : mode == 1 ? 255 : mode == 2 ? 0 : mode == 3 ? (unsigned char)(20 * r + 9 * c) : (unsigned char)(rnd() & 1 ? 255 : 0);
    float u = 3.0f + (float)(rnd() % 600) + (float)rnd() / 4294967296.0f, v = 3.0f + (float)(rnd() % 400) + (float)rnd() / 4294967296.0f;
    if (n % 7 == 0) u = floorf(u);                       // fraction exactly 0
    if (n % 11 == 0) v = floorf(v) + 5.9604645e-8f * 8;  // tiny fraction
    if (n % 13 == 0) u = nextafterf(floorf(u) + 1.0f, 0.0f);   // fraction just below 1
    // ---- direct form (align_fused_kernel's precompute): lane `row` computes patch row `row` from image rows vi-3+row .. +3
    const PatchW pw = patch_weights(u, v);
    float4 dr[4], dx[4], dy[4];
    for (int row = 0; row < 4; ++row) {
      float I[4][7];
      for (int rr = 0; rr < 4; ++rr) for (int c = 0; c < 7; ++c) I[rr][c] = (float)win[row + rr][c];
      ref_row_direct(I, pw.wTL, pw.wTR, pw.wBL, pw.wBR, dr[row], dx[row], dy[row]);
    }
    // ---- record form: 64 bytes = 7 rows x 8 B, then su, sv; the slot's lane reads the record as four 16-byte words
    unsigned char rec[64];
    for (int r = 0; r < 7; ++r) memcpy(rec + 8 * r, win[r], 8);
    const float frac[2] = { u - floorf(u), v - floorf(v) };
    memcpy(rec + 56, frac, 8);
    uint4 q[4];
    memcpy(q, rec, 64);
    float4 rr[4], rx[4], ry[4];
    ref_patch_from_record(q, rr, rx, ry);
    for (int row = 0; row < 4; ++row) {
      bad += memcmp(&rr[row], &dr[row], 16) != 0; bad += memcmp(&rx[row], &dx[row], 16) != 0; bad += memcmp(&ry[row], &dy[row], 16) != 0;
    }
  }
  printf("%d %ld\n", cases, bad);
  return bad != 0;
}
