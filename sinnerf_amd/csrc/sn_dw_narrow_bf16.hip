// sn_dw_narrow_bf16.hip -- the NARROW weight-gradient contractions  dW[m, n] = sum_p G[p, m] X[p, n]  (+ db[m]) of the bf16-state
// training step with HAND-SCHEDULED inner loops: everything of models/nerf.py:66-103 that is not 256 x 256 (xyz_encoding_1, the skip
// layer's xyz columns, both halves of dir_encoding, sigma, rgb: variants 1..7 of sn_dw_common.h, 3.9 / 3.5 KB per sample point).  Same
// tasks, K-split plan, tiles, swizzled DMA image, transpose-read fragments and results as dw_narrow_bf16_kernel (sn_dw.hip run_task)
// -- two workgroups per CU, one workgroup = one K-range of one problem -- what changes is who lays out the instruction stream:
// tools/gen_dw_narrow.py, one asm statement per PAIR of 16-point chunks.  The compiler-scheduled loop ran at 5.0 TB/s with its waves
// parked 74 % of the time and 12 SALU instructions per MFMA (profiles/r03_run16_train_pmc_cycles.json).
//
// A wave owns 128 VGPRs + 128 AGPRs (two waves per SIMD): the accumulators of its MT x NT blocks ARE a[0 : 16 MT NT) for the whole
// task, across the per-pair statements and the C++ glue between them (tools/check_agpr.py: no compiler AGPR, no scratch).
#include "sn_dw_common.h"
#include "sn_dw_narrow_chunk.inc"

namespace snd {

constexpr int DWN_LDS_BYTES = 81920;             // per workgroup; two share a CU

template <int R>
SN_DEV float dwn_acc_read() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(R));
  return x;
}
template <int NT, int A, int B, int R>
SN_DEV void dwn_store_block(const Task& t, int m0, int n0, int i, int h) {
  if constexpr (R < 16) {
    const int m = m0 + 32 * A + (R & 3) + 8 * (R >> 2) + 4 * h;
    t.c[(long)m * t.ldc + n0 + 32 * B + i] = dwn_acc_read<16 * (NT * A + B) + R>();
    dwn_store_block<NT, A, B, R + 1>(t, m0, n0, i, h);
  }
}
template <int MT, int NT, int AB>
SN_DEV void dwn_store_all(const Task& t, int m0, int n0, int i, int h) {
  if constexpr (AB < MT * NT) {
    dwn_store_block<NT, AB / NT, AB % NT, 0>(t, m0, n0, i, h);
    dwn_store_all<MT, NT, AB + 1>(t, m0, n0, i, h);
  }
}

// A transpose read touches 4 consecutive rows x 64 bytes per half-wave: rows a multiple of 256 B apart would sit on the same banks.
// The DMA therefore builds a swizzled image (the LDS side of global_load_lds is lane-linear, each lane's GLOBAL address is free):
// LDS piece (row, p) holds global 16-byte piece (row, swz(row, p)) --
//   rows of >= 256 B (128+ columns): p ^ 4 (row & 3)        (sn_dw.hip RowStager)
//   rows of 128 B (64 columns):      p ^ 4 ((row >> 1) & 1)  (rows r and r + 2 would collide)
//   rows of 64 B (32 columns):       none (four rows fill the 256-byte bank window by themselves)
template <int PER_ROW>
SN_DEV int dwn_swz(int row, int p) { return PER_ROW >= 16 ? (p ^ (4 * (row & 3))) : PER_ROW == 8 ? (p ^ (4 * ((row >> 1) & 1))) : p; }
// byte offset inside a staged W-wide bf16 tile of the 8-byte group (row, columns col .. col+3), col % 4 == 0
template <int W>
SN_DEV unsigned dwn_tr_offset(int row, int col) {
  return (unsigned)(row * W * 2 + dwn_swz<W * 2 / 16>(row, col / 8) * 16 + (col * 2) % 16);
}

// One task of variant V.  MT x NT blocks per wave, WM x WN waves, EB = element size of the B tile (4: the embedded inputs stay fp32).
#define SN_DWN_TASK(V, MT_, NT_, WM_, WN_, EB_)                                                                                    \
SN_DEV void dwn_task_##V(const Task& t, int tid) {                                                                                 \
  constexpr int MT = MT_, NT = NT_, WM = WM_, WN = WN_, EB = EB_;                                                                  \
  constexpr int WA = WM * MT * 32, WB = WN * NT * 32;                                                                              \
  constexpr int A_BYTES = KB * WA * 2, B_BYTES = KB * WB * EB, BUF = A_BYTES + B_BYTES;                                            \
  constexpr int R = SN_DWN##V##_RING;                                                                                              \
  static_assert(BUF == SN_DWN##V##_BUF && R * BUF <= DWN_LDS_BYTES && (R & 1) == 0 && R >= 6, "ring of the generated statement");  \
  constexpr int CH_A = A_BYTES / 16, CH_B = B_BYTES / 16;                                                                          \
  constexpr int IT_A = (CH_A + 255) / 256, IT_B = (CH_B + 255) / 256;                                                              \
  /* a tile of fewer than 256 pieces is staged by its first waves only: a 32-wide A tile (64 pieces) by wave 0, a 64-wide bf16 B   \
     tile (128 pieces) by waves 0 and 1 -- the waves that stage BOTH tiles run the W0 form of the statement, the others WX */      \
  constexpr int A_WAVES = CH_A % 256 == 0 ? 4 : CH_A / 64, B_WAVES = CH_B % 256 == 0 ? 4 : CH_B / 64;                              \
  constexpr int N_FULL = A_WAVES < B_WAVES ? A_WAVES : B_WAVES;                                                                    \
  static_assert(A_WAVES == 4 || B_WAVES == 4, "one of the tiles is staged by every wave");                                         \
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);                                                      \
  const int i = lane & 31, h = lane >> 5;                                                                                          \
  const int wr = wave / WN, wc = wave % WN;                                                                                        \
  const int m0 = wr * MT * 32, n0 = wc * NT * 32;                                                                                  \
  const long k0 = t.k0, k1 = t.k1;                                                                                                 \
  if (k0 >= k1) return;                                                                                                            \
  const int n_chunks = (int)((k1 - k0) / KB);                                                                                      \
  const int n_pairs = n_chunks >> 1;                                                                                               \
  /* per-thread global byte offsets of the DMA pieces of a chunk (piece = 256 threads x 16 B; swizzled tiles: see dwn_tr_offset) */ \
  unsigned oa[2] = {0u, 0u}, ob[2] = {0u, 0u};                                                                                     \
  _Pragma("unroll") for (int it = 0; it < IT_A; ++it) {                                                                            \
    const int c = it * 256 + tid, per_row = WA * 2 / 16;                                                                           \
    const int row = c / per_row, lp = c % per_row;                                                                                 \
    oa[it] = (unsigned)(row * t.lda * 2 + dwn_swz<WA * 2 / 16>(row, lp) * 16);                                                     \
  }                                                                                                                                \
  _Pragma("unroll") for (int it = 0; it < IT_B; ++it) {                                                                            \
    const int c = it * 256 + tid, per_row = WB * EB / 16;                                                                          \
    const int row = c / per_row, lp = c % per_row;                                                                                 \
    ob[it] = (unsigned)(row * t.ldb * EB + (EB == 2 ? dwn_swz<WB * EB / 16>(row, lp) : lp) * 16);                                  \
  }                                                                                                                                \
  /* fragment addresses of this lane inside a slot: transpose reads (lane (q, G): feature block G & 1, point rows 8 (G >> 1) +     \
     (q >> 2)); fp32 B tile: point row 8 h of feature n0 + i */                                                                    \
  unsigned ta[4] = {0u, 0u, 0u, 0u}, tb[4] = {0u, 0u, 0u, 0u}, tbf = 0u;                                                           \
  {                                                                                                                                \
    const int q = lane & 15, G = lane >> 4;                                                                                        \
    _Pragma("unroll") for (int a = 0; a < MT; ++a) ta[a] = dwn_tr_offset<WA>(8 * (G >> 1) + (q >> 2), m0 + 32 * a + 16 * (G & 1) + 4 * (q & 3)); \
    if (EB == 2) {                                                                                                                 \
      _Pragma("unroll") for (int b = 0; b < NT; ++b)                                                                               \
        tb[b] = dwn_tr_offset<WB>(8 * (G >> 1) + (q >> 2), n0 + 32 * b + 16 * (G & 1) + 4 * (q & 3)) + A_BYTES;                    \
    } else {                                                                                                                       \
      tbf = (unsigned)((8 * h * WB + n0 + i) * 4 + A_BYTES);                                                                       \
    }                                                                                                                              \
  }                                                                                                                                \
  const char* ga_base = reinterpret_cast<const char*>(t.a);                                                                        \
  const char* gb_base = reinterpret_cast<const char*>(t.b);                                                                        \
  auto chunk_a = [&](long k) __attribute__((always_inline)) { const long kc = k < k1 ? k : k1 - KB; return ga_base + kc * t.lda * 2; };   \
  auto chunk_b = [&](long k) __attribute__((always_inline)) { const long kc = k < k1 ? k : k1 - KB; return gb_base + kc * t.ldb * EB; };  \
  /* prologue: R - 2 chunks in flight (dynamic LDS starts at address 0) */                                                         \
  _Pragma("unroll") for (int c = 0; c < R - 2; ++c) {                                                                              \
    const char* ba = chunk_a(k0 + (long)c * KB);                                                                                   \
    const char* bb = chunk_b(k0 + (long)c * KB);                                                                                   \
    if (wave < A_WAVES) {                                                                                                          \
      _Pragma("unroll") for (int it = 0; it < IT_A; ++it)                                                                          \
        __builtin_amdgcn_global_load_lds((gbl_cvoid*)(ba + oa[it]), (lds_void*)(size_t)(c * BUF + it * 4096 + wave * 1024), 16, 0, 2);    \
    }                                                                                                                              \
    if (wave < B_WAVES) {                                                                                                          \
      _Pragma("unroll") for (int it = 0; it < IT_B; ++it)                                                                          \
        __builtin_amdgcn_global_load_lds((gbl_cvoid*)(bb + ob[it]), (lds_void*)(size_t)(c * BUF + A_BYTES + it * 4096 + wave * 1024), 16, 0, 2); \
    }                                                                                                                              \
  }                                                                                                                                \
  asm volatile(SN_DWN##V##_ZERO_ASM ::: SN_DWN_AGPR_CLOBBERS);                                                                     \
  float bs0 = 0.0f, bs1 = 0.0f, bs2 = 0.0f, bs3 = 0.0f;                                                                            \
  unsigned one = 0x3f803f80u;                      /* bf16 (1, 1): v_dot2c_f32_bf16 with it = fp32 sum of a packed pair */        \
  asm volatile("" : "+v"(one));                                                                                                    \
  int s0 = 0;                                      /* ring slot of chunk c = 2 p (R need not be a power of two) */                  \
  _Pragma("unroll 1") for (int p = 0; p < n_pairs; ++p) {                                                                          \
    const int c = 2 * p;                                                                                                           \
    const int s1 = s0 + 1, sn0 = (s0 + R - 2 >= R) ? s0 - 2 : s0 + R - 2, sn1 = sn0 + 1;   /* R even, s0 even: no wrap inside a pair */ \
    const unsigned sl0 = (unsigned)s0 * BUF, sl1 = (unsigned)s1 * BUF;                                                             \
    const long kn0 = k0 + (long)(c + R - 2) * KB, kn1 = kn0 + KB;                                                                  \
    const char* ga0 = chunk_a(kn0); const char* gb0 = chunk_b(kn0);                                                                \
    const char* ga1 = chunk_a(kn1); const char* gb1 = chunk_b(kn1);                                                                \
    const unsigned md0 = (unsigned)sn0 * BUF + (unsigned)wave * 1024u, md1 = (unsigned)sn1 * BUF + (unsigned)wave * 1024u;         \
    if (wave < N_FULL) {                                                                                                           \
      asm volatile(SN_DWN##V##_PAIR_W0_ASM                                                                                         \
                   : [bs0] "+v"(bs0), [bs1] "+v"(bs1), [bs2] "+v"(bs2), [bs3] "+v"(bs3)                                            \
                   : [ta0] "v"(ta[0]), [ta1] "v"(ta[1]), [ta2] "v"(ta[2]), [ta3] "v"(ta[3]),                                       \
                     [tb0] "v"(tb[0]), [tb1] "v"(tb[1]), [tb2] "v"(tb[2]), [tb3] "v"(tb[3]), [tbf] "v"(tbf),                       \
                     [one] "v"(one), [oa0] "v"(oa[0]), [oa1] "v"(oa[1]), [ob0] "v"(ob[0]), [ob1] "v"(ob[1]),                       \
                     [sl0] "s"(sl0), [sl1] "s"(sl1), [ga0] "s"(ga0), [gb0] "s"(gb0), [ga1] "s"(ga1), [gb1] "s"(gb1),               \
                     [md0] "s"(md0), [md1] "s"(md1)                                                                                \
                   : SN_DWN_VGPR_CLOBBERS, SN_DWN_AGPR_CLOBBERS, "memory", "scc");                                                 \
    } else {                                                                                                                       \
      asm volatile(SN_DWN##V##_PAIR_WX_ASM                                                                                         \
                   : [bs0] "+v"(bs0), [bs1] "+v"(bs1), [bs2] "+v"(bs2), [bs3] "+v"(bs3)                                            \
                   : [ta0] "v"(ta[0]), [ta1] "v"(ta[1]), [ta2] "v"(ta[2]), [ta3] "v"(ta[3]),                                       \
                     [tb0] "v"(tb[0]), [tb1] "v"(tb[1]), [tb2] "v"(tb[2]), [tb3] "v"(tb[3]), [tbf] "v"(tbf),                       \
                     [one] "v"(one), [oa0] "v"(oa[0]), [oa1] "v"(oa[1]), [ob0] "v"(ob[0]), [ob1] "v"(ob[1]),                       \
                     [sl0] "s"(sl0), [sl1] "s"(sl1), [ga0] "s"(ga0), [gb0] "s"(gb0), [ga1] "s"(ga1), [gb1] "s"(gb1),               \
                     [md0] "s"(md0), [md1] "s"(md1)                                                                                \
                   : SN_DWN_VGPR_CLOBBERS, SN_DWN_AGPR_CLOBBERS, "memory", "scc");                                                 \
    }                                                                                                                              \
    s0 = (s0 + 2 == R) ? 0 : s0 + 2;                                                                                               \
  }                                                                                                                                \
  if (n_chunks & 1) {                              /* odd last chunk (it landed long ago) */                                       \
    const unsigned sl0 = (unsigned)s0 * BUF;                                                                                       \
    asm volatile(SN_DWN##V##_TAIL_ASM                                                                                              \
                 : [bs0] "+v"(bs0), [bs1] "+v"(bs1), [bs2] "+v"(bs2), [bs3] "+v"(bs3)                                              \
                 : [ta0] "v"(ta[0]), [ta1] "v"(ta[1]), [ta2] "v"(ta[2]), [ta3] "v"(ta[3]),                                         \
                   [tb0] "v"(tb[0]), [tb1] "v"(tb[1]), [tb2] "v"(tb[2]), [tb3] "v"(tb[3]), [tbf] "v"(tbf), [one] "v"(one), [sl0] "s"(sl0) \
                 : SN_DWN_VGPR_CLOBBERS, SN_DWN_AGPR_CLOBBERS, "memory", "scc");                                                   \
  }                                                                                                                                \
  /* drain the over-issued tail chunks before the LDS is released; MFMA (8 passes) -> accumulator read: 11 wait states */          \
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");                                                           \
  dwn_store_all<MT, NT, 0>(t, m0, n0, i, h);                                                                                       \
  if (t.bias != nullptr && wc == 0) {                                                                                              \
    const float b[4] = {bs0, bs1, bs2, bs3};                                                                                       \
    _Pragma("unroll") for (int a = 0; a < MT; ++a) {                                                                               \
      const float v = b[a] + __shfl_xor(b[a], 32, 64);                                                                             \
      if (h == 0) t.bias[m0 + 32 * a + i] = v;                                                                                     \
    }                                                                                                                              \
  }                                                                                                                                \
}

// The same task with NCH = 4 chunks per sync point (tools/gen_dw_narrow.py QUAD: the shapes whose chunks are 5 - 6 KB; a workgroup's
// streaming rate follows the bytes it consumes per barrier).  One A and one B DMA piece per thread and chunk (bf16 tiles only).
#define SN_DWN_TASK_Q(V, MT_, NT_, WM_, WN_)                                                                                      \
SN_DEV void dwn_task_##V(const Task& t, int tid) {                                                                                 \
  constexpr int MT = MT_, NT = NT_, WM = WM_, WN = WN_, EB = 2, NCH = SN_DWN##V##_NCH;                                             \
  constexpr int WA = WM * MT * 32, WB = WN * NT * 32;                                                                              \
  constexpr int A_BYTES = KB * WA * 2, B_BYTES = KB * WB * EB, BUF = A_BYTES + B_BYTES;                                            \
  constexpr int R = SN_DWN##V##_RING;                                                                                              \
  static_assert(BUF == SN_DWN##V##_BUF && R * BUF <= DWN_LDS_BYTES && R % NCH == 0 && R >= 3 * NCH && NCH == 4 && MT <= 2 && NT == 1, "generated statement"); \
  constexpr int CH_A = A_BYTES / 16, CH_B = B_BYTES / 16;                                                                          \
  static_assert(CH_A <= 256 && CH_B <= 256, "one DMA piece per thread, tile and chunk");                                           \
  constexpr int A_WAVES = CH_A / 64, B_WAVES = CH_B / 64;                                                                          \
  constexpr int N_FULL = A_WAVES < B_WAVES ? A_WAVES : B_WAVES;                                                                    \
  static_assert(A_WAVES == 4 || B_WAVES == 4, "one of the tiles is staged by every wave");                                         \
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);                                                      \
  const int i = lane & 31, h = lane >> 5;                                                                                          \
  const int wr = wave / WN, wc = wave % WN;                                                                                        \
  const int m0 = wr * MT * 32, n0 = wc * NT * 32;                                                                                  \
  const long k0 = t.k0, k1 = t.k1;                                                                                                 \
  if (k0 >= k1) return;                                                                                                            \
  const int n_chunks = (int)((k1 - k0) / KB);                                                                                      \
  const int n_groups = n_chunks / NCH;                                                                                             \
  unsigned oa, ob;                                                                                                                 \
  {                                                                                                                                \
    int per_row = WA * 2 / 16, row = tid / per_row, lp = tid % per_row;                                                            \
    oa = (unsigned)(row * t.lda * 2 + dwn_swz<WA * 2 / 16>(row, lp) * 16);                                                         \
    per_row = WB * 2 / 16; row = tid / per_row; lp = tid % per_row;                                                                \
    ob = (unsigned)(row * t.ldb * 2 + dwn_swz<WB * 2 / 16>(row, lp) * 16);                                                         \
  }                                                                                                                                \
  unsigned ta[2] = {0u, 0u}, tb;                                                                                                   \
  {                                                                                                                                \
    const int q = lane & 15, G = lane >> 4;                                                                                        \
    _Pragma("unroll") for (int a = 0; a < MT; ++a) ta[a] = dwn_tr_offset<WA>(8 * (G >> 1) + (q >> 2), m0 + 32 * a + 16 * (G & 1) + 4 * (q & 3)); \
    tb = dwn_tr_offset<WB>(8 * (G >> 1) + (q >> 2), n0 + 16 * (G & 1) + 4 * (q & 3)) + A_BYTES;                                    \
  }                                                                                                                                \
  const char* ga_base = reinterpret_cast<const char*>(t.a);                                                                        \
  const char* gb_base = reinterpret_cast<const char*>(t.b);                                                                        \
  auto chunk_a = [&](long k) __attribute__((always_inline)) { const long kc = k < k1 ? k : k1 - KB; return ga_base + kc * t.lda * 2; };   \
  auto chunk_b = [&](long k) __attribute__((always_inline)) { const long kc = k < k1 ? k : k1 - KB; return gb_base + kc * t.ldb * 2; };   \
  _Pragma("unroll") for (int c = 0; c < R - NCH; ++c) {                                                                            \
    if (wave < A_WAVES)                                                                                                            \
      __builtin_amdgcn_global_load_lds((gbl_cvoid*)(chunk_a(k0 + (long)c * KB) + oa), (lds_void*)(size_t)(c * BUF + wave * 1024), 16, 0, 2);            \
    if (wave < B_WAVES)                                                                                                            \
      __builtin_amdgcn_global_load_lds((gbl_cvoid*)(chunk_b(k0 + (long)c * KB) + ob), (lds_void*)(size_t)(c * BUF + A_BYTES + wave * 1024), 16, 0, 2);  \
  }                                                                                                                                \
  asm volatile(SN_DWN##V##_ZERO_ASM ::: SN_DWN_AGPR_CLOBBERS);                                                                     \
  float bs0 = 0.0f, bs1 = 0.0f;                                                                                                    \
  unsigned one = 0x3f803f80u;                                                                                                      \
  asm volatile("" : "+v"(one));                                                                                                    \
  int s0 = 0;                                      /* ring slot of the group's first chunk (R % NCH == 0: no wrap inside a group) */ \
  _Pragma("unroll 1") for (int g = 0; g < n_groups; ++g) {                                                                         \
    const int sn = s0 >= NCH ? s0 - NCH : s0 + R - NCH;          /* slots of the group R - NCH chunks ahead = those of the previous group */ \
    const unsigned sl0 = (unsigned)s0 * BUF, sl1 = sl0 + BUF, sl2 = sl0 + 2 * BUF, sl3 = sl0 + 3 * BUF;                            \
    const unsigned md0 = (unsigned)sn * BUF + (unsigned)wave * 1024u, md1 = md0 + BUF, md2 = md0 + 2 * BUF, md3 = md0 + 3 * BUF;   \
    const long kn = k0 + (long)(NCH * g + R - NCH) * KB;                                                                           \
    const char* ga0 = chunk_a(kn); const char* gb0 = chunk_b(kn);                                                                  \
    const char* ga1 = chunk_a(kn + KB); const char* gb1 = chunk_b(kn + KB);                                                        \
    const char* ga2 = chunk_a(kn + 2 * KB); const char* gb2 = chunk_b(kn + 2 * KB);                                                \
    const char* ga3 = chunk_a(kn + 3 * KB); const char* gb3 = chunk_b(kn + 3 * KB);                                                \
    if (wave < N_FULL) {                                                                                                           \
      asm volatile(SN_DWN##V##_GROUP_W0_ASM                                                                                        \
                   : [bs0] "+v"(bs0), [bs1] "+v"(bs1)                                                                              \
                   : [ta0] "v"(ta[0]), [ta1] "v"(ta[1]), [tb0] "v"(tb), [one] "v"(one), [oa0] "v"(oa), [ob0] "v"(ob),              \
                     [sl0] "s"(sl0), [sl1] "s"(sl1), [sl2] "s"(sl2), [sl3] "s"(sl3),                                               \
                     [ga0] "s"(ga0), [gb0] "s"(gb0), [ga1] "s"(ga1), [gb1] "s"(gb1), [ga2] "s"(ga2), [gb2] "s"(gb2), [ga3] "s"(ga3), [gb3] "s"(gb3), \
                     [md0] "s"(md0), [md1] "s"(md1), [md2] "s"(md2), [md3] "s"(md3)                                                \
                   : SN_DWN_VGPR_CLOBBERS, SN_DWN_AGPR_CLOBBERS, "memory", "scc");                                                 \
    } else {                                                                                                                       \
      asm volatile(SN_DWN##V##_GROUP_WX_ASM                                                                                        \
                   : [bs0] "+v"(bs0), [bs1] "+v"(bs1)                                                                              \
                   : [ta0] "v"(ta[0]), [ta1] "v"(ta[1]), [tb0] "v"(tb), [one] "v"(one), [oa0] "v"(oa), [ob0] "v"(ob),              \
                     [sl0] "s"(sl0), [sl1] "s"(sl1), [sl2] "s"(sl2), [sl3] "s"(sl3),                                               \
                     [ga0] "s"(ga0), [gb0] "s"(gb0), [ga1] "s"(ga1), [gb1] "s"(gb1), [ga2] "s"(ga2), [gb2] "s"(gb2), [ga3] "s"(ga3), [gb3] "s"(gb3), \
                     [md0] "s"(md0), [md1] "s"(md1), [md2] "s"(md2), [md3] "s"(md3)                                                \
                   : SN_DWN_VGPR_CLOBBERS, SN_DWN_AGPR_CLOBBERS, "memory", "scc");                                                 \
    }                                                                                                                              \
    s0 = (s0 + NCH == R) ? 0 : s0 + NCH;                                                                                           \
  }                                                                                                                                \
  _Pragma("unroll 1") for (int c = n_groups * NCH; c < n_chunks; ++c) {     /* up to three left-over chunks (staged long ago), one at a time */ \
    const unsigned sl0 = (unsigned)s0 * BUF;                                                                                       \
    asm volatile(SN_DWN##V##_TAIL_ASM                                                                                              \
                 : [bs0] "+v"(bs0), [bs1] "+v"(bs1)                                                                                \
                 : [ta0] "v"(ta[0]), [ta1] "v"(ta[1]), [tb0] "v"(tb), [one] "v"(one), [sl0] "s"(sl0)                               \
                 : SN_DWN_VGPR_CLOBBERS, SN_DWN_AGPR_CLOBBERS, "memory", "scc");                                                   \
    s0 = (s0 + 1 == R) ? 0 : s0 + 1;                                                                                              \
  }                                                                                                                                \
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");                                                         \
  dwn_store_all<MT, NT, 0>(t, m0, n0, i, h);                                                                                       \
  if (t.bias != nullptr && wc == 0) {                                                                                              \
    const float b[2] = {bs0, bs1};                                                                                                 \
    _Pragma("unroll") for (int a = 0; a < MT; ++a) {                                                                               \
      const float v = b[a] + __shfl_xor(b[a], 32, 64);                                                                             \
      if (h == 0) t.bias[m0 + 32 * a + i] = v;                                                                                     \
    }                                                                                                                              \
  }                                                                                                                                \
}

// the PAIR_WX form only exists for the variants whose A tile is staged by wave 0 alone
#define SN_DWN1_PAIR_WX_ASM SN_DWN1_PAIR_W0_ASM
#define SN_DWN2_PAIR_WX_ASM SN_DWN2_PAIR_W0_ASM
#define SN_DWN3_PAIR_WX_ASM SN_DWN3_PAIR_W0_ASM

SN_DWN_TASK(1, 4, 1, 2, 2, 4)
SN_DWN_TASK(2, 2, 4, 2, 2, 2)
SN_DWN_TASK(3, 2, 1, 2, 2, 4)
SN_DWN_TASK(4, 1, 2, 1, 4, 2)
SN_DWN_TASK_Q(5, 1, 1, 1, 4)
SN_DWN_TASK(6, 4, 1, 2, 2, 2)
SN_DWN_TASK_Q(7, 2, 1, 2, 2)

__global__ void __launch_bounds__(256, 2) dw_narrow_bf16_asm_kernel(const Plan plan) {
  asm volatile("" ::: "a0", "a127");               // the hand-managed accumulator file
  const Task t = task_of(plan, (int)blockIdx.x);
  const int tid = threadIdx.x;
  switch (t.variant & 0xff) {
    case 1: dwn_task_1(t, tid); break;
    case 2: dwn_task_2(t, tid); break;
    case 3: dwn_task_3(t, tid); break;
    case 4: dwn_task_4(t, tid); break;
    case 5: dwn_task_5(t, tid); break;
    case 6: dwn_task_6(t, tid); break;
    default: dwn_task_7(t, tid); break;
  }
}

}  // namespace snd

extern "C" int sn_dw_narrow_bf16_asm_launch(const snd::Plan* plan_host, hipStream_t stream) {
  using namespace snd;
  if (plan_host->n_tasks <= 0) return 0;
  SN_ENSURE_DYN_LDS(dw_narrow_bf16_asm_kernel, DWN_LDS_BYTES);
  hipLaunchKernelGGL(dw_narrow_bf16_asm_kernel, dim3((unsigned)plan_host->n_tasks), dim3(256), DWN_LDS_BYTES, stream, *plan_host);
  return (int)hipGetLastError();
}
