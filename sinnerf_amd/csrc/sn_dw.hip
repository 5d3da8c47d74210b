// sn_dw.hip -- weight-gradient contractions of the NeRF MLP backward for gfx950 (fp32):
//     dW[m, n] = sum_p  G[p, m] * X[p, n]          (+ optionally  db[m] = sum_p G[p, m])
// i.e. the "gW = g_y^T x" terms torch autograd accumulates for every nn.Linear of models/nerf.py:66-103, with the
// contraction running over ALL sample points p (K = 0.25..1 M) and a small M x N <= 256 x 256 result.
//
// G (pre-activation gradients, written by sn_mlp_bwd.hip) and X (activations / embedded inputs, written by the
// training forward) are row-major [P][ld]: a row = one point.  With v_mfma_f32_32x32x2_f32 the A operand is
// A[i][k] = G[p=k][m0+i] and the B operand B[k][j] = X[p=k][n0+j]: both read 32 CONSECUTIVE floats of a row per lane
// half, so the tiles are staged row-major (global_load_lds DMA, lane-linear) and fragment reads are conflict-free
// ds_read_b32 -- no transposes anywhere.
//
// One workgroup = one task = (problem, K-range): 2x2 waves, each wave owns an (MT*32) x (NT*32) block of accumulators
// (MT=NT=4: 256 accumulator registers) and walks its K-range in chunks of 16 points.  G and X stream from HBM exactly
// once per problem, so unlike the L2-resident weight slabs of the MLP kernels the DMA needs depth: a 4-deep LDS ring,
// three chunks in flight, COUNTED s_waitcnt vmcnt(N) + raw s_barrier (a __syncthreads() would drain the queue).
// Partial results go to a per-task slab; the K-split partials are summed afterwards (deterministic, no atomics).
#include "sn_dw_common.h"

namespace snd {

#ifndef SN_DW_NARROW_2WG
#define SN_DW_NARROW_2WG 1      // bf16-state narrow problems: two workgroups per CU (0: comparison build)
#endif
#ifndef SN_DW_CPOL
#define SN_DW_CPOL 2      // nt: G and X are streamed once (measured -3 % in the bandwidth-bound bf16 mode, neutral in fp32)
#endif
#if SN_DW_CPOL == 2
#define SN_DW_CPOL_ASM " nt"
#elif SN_DW_CPOL == 0
#define SN_DW_CPOL_ASM ""
#else
#error "SN_DW_CPOL: 0 or 2 (nt)"
#endif

// copy KB x W floats (row-major, W*4 bytes per row) global -> LDS.  A chunk past k_end (the ring's prefetch overrun) is
// replaced by the last real chunk of the task ((k1-k0) % KB == 0) -- a wave-uniform select on the chunk base, so the
// per-thread part of the address is a 32-bit byte offset computed once per task (off[it]) and a DMA instruction costs one
// address add instead of a 64-bit multiply + per-row clamp.
// bf16 tiles are read back with ds_read_b64_tr_b16 (below): a half-wave then touches 4 consecutive rows x 64 bytes, and rows
// that are a multiple of 256 B apart would all sit on the same 16 banks.  The DMA therefore builds a SWIZZLED image: the LDS
// side of global_load_lds is lane-linear, but each lane's GLOBAL address is free, so LDS piece (row, lp) receives the global
// 16-byte piece (row, lp ^ 4 (row & 3)) -- rows r..r+3 of a 64-byte column group land in four different 64-byte bank groups.
// (32-wide tiles: a row is 64 B, four rows fill the 256-byte bank window by themselves -- no swizzle.)
// SPLIT tiles (bf16x3 training state, slots 0..8 of acts / G -- sn_layout.h "x3 state"): a row of W features is W*4 bytes like an fp32
// row, but holds per 8 features 16 B of hi parts (bf16) then 16 B of lo parts.  Fragments come out by transpose reads of 8 B = four
// features of one part: a half-wave touches, per row, every second 16-byte piece of a 128-byte span -- the swizzle puts rows r, r+1
// on the even / odd pieces (xor 1: hi and lo pieces trade places on odd rows) and rows r+2, r+3 on the next 128 B (xor 8): the 32 lanes
// cover one 256-byte bank window exactly.
template <int W, int ES, int SPLIT = 0, bool SPLIT_ASM_DMA = false>   // ES = element size in bytes (4: fp32 tile, 2: bf16 tile); SPLIT: (hi, lo)
struct RowStager {                                                    // tile, ES = 4; SPLIT_ASM_DMA: the DMA from inline asm (bf16x3 modes)
  static constexpr int CHUNKS = KB * W * ES / 16;   // 16-byte pieces per chunk
  static constexpr int PER_ROW = W * ES / 16;
  static constexpr int IT = (CHUNKS + 255) / 256;
  static constexpr bool SWZ = (ES == 2) && PER_ROW >= 16;
  static_assert(!SPLIT || (ES == 4 && PER_ROW >= 16), "split tiles are at least 64 features wide");
  unsigned off[IT];
  static SN_DEV int swz(int row) { return SPLIT ? ((row & 1) | ((row & 2) << 2)) : SWZ ? 4 * (row & 3) : 0; }
  SN_DEV void init(int ld, int tid) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int c = it * 256 + tid;
      const int row = c / PER_ROW, lp = c % PER_ROW;
      const int gp = lp ^ swz(row);
      off[it] = (unsigned)(row * ld * ES + gp * 16);
    }
  }
  // byte offset inside a staged chunk of the 8-byte group (row, columns col .. col+3), col % 4 == 0
  static SN_DEV unsigned tr_offset(int row, int col) {
    const int cp = col * ES / 16;
    const int lp = cp ^ swz(row);
    return (unsigned)(row * W * ES + lp * 16 + (col * ES) % 16);
  }
  // split tile: the 8-byte group of the HI parts of features f .. f+3 (f % 4 == 0) of a row; the lo parts sit at this offset ^ 16
  static SN_DEV unsigned tr_offset_split(int row, int f) {
    const int lp = (2 * (f >> 3)) ^ swz(row);
    return (unsigned)(row * W * 4 + lp * 16 + 8 * ((f >> 2) & 1));
  }
  SN_DEV void stage(const void* __restrict__ g, int ld, long k, long k_end, char* lds, int tid) const {
    const long kc = k < k_end ? k : k_end - KB;
    const char* base = reinterpret_cast<const char*>(g) + kc * ld * ES;        // wave-uniform
    if (SPLIT_ASM_DMA) {                                                        // ... and provably so for the "s" operand of the asm below
      const unsigned long long b = reinterpret_cast<unsigned long long>(base);
      base = reinterpret_cast<const char*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                                           (unsigned)__builtin_amdgcn_readfirstlane((int)b));
    }
    const int wbase = __builtin_amdgcn_readfirstlane((tid & ~63) * 16);
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      if (CHUNKS % 256 == 0 || it * 256 + tid < CHUNKS) {
        if (SPLIT_ASM_DMA) {
          // inline asm: while a __builtin_amdgcn_global_load_lds is pending, every wait hipcc inserts is vmcnt(0) / lgkmcnt(0)
          // (sn_mlp_pipe.h) -- here that would drain the fragment reads of the NEXT chunk in front of this chunk's MFMAs
          asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" SN_DW_CPOL_ASM
                       :: "v"(off[it]), "s"(base), "s"((unsigned)(size_t)(lds + it * 4096 + wbase)) : "memory");
        } else {
          __builtin_amdgcn_global_load_lds((gbl_cvoid*)(base + off[it]), (lds_void*)(lds + it * 4096 + wbase), 16, 0, SN_DW_CPOL);
        }
      }
    }
  }
};

template <int N>
SN_DEV void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// WM x WN waves, each wave an (MT*32) x (NT*32) accumulator block.
typedef __bf16 dw_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned dw_u32x4 __attribute__((ext_vector_type(4)));
typedef float dw_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 dw_bf16x2 __attribute__((ext_vector_type(2)));
typedef short dw_i16x4 __attribute__((ext_vector_type(4)));
typedef short dw_i16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) dw_i16x4 lds_i16x4;
// hardware transpose read: within a 16-lane group, lane p supplies the address of 4 consecutive bf16 of row (p >> 2),
// columns 4 (p & 3) .. +3; lane q receives column q of the 4 x 16 block, rows 0..3 (checked by tools/ubench/tr_probe.hip).
// A builtin: the compiler tracks its lgkmcnt like any LDS read.
SN_DEV dw_i16x4 tr_read(const char* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4*)p); }
// {bf16(a), bf16(b)}, RNE: one v_cvt_pk_bf16_f32.  A builtin, NOT inline asm: the MFMAs of this kernel are builtins too and
// the compiler must see the VALU write -> MFMA read dependence to pad it (an asm conversion right in front of the MFMA
// that consumes it returned garbage in the narrow variants).
SN_DEV unsigned dw_pack2(float a, float b) {
  const dw_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, dw_bf16x2));
}

// BF16: the same fp32 row-major tiles are staged (nothing changes on the memory side), but a chunk of 16 points is ONE
// k-step of v_mfma_f32_32x32x16_bf16: lane (i, h) gathers its 8 points of feature i from the LDS tile (stride = row pitch,
// conflict-free across lanes), converts them to a bf16x8 fragment (RNE) and keeps the fp32 column sums for the bias
// gradient.  16 MFMAs of 32 cycles per chunk instead of 128 of 64: the kernel becomes HBM-bound.
// MODE 0: fp32 MFMAs.  1: bf16 MFMAs, fp32 tiles.  2: bf16 MFMAs, bf16 A and B tiles.  3: bf16 MFMAs, bf16 A tile, fp32 B tile.
// MODE 4 (SN_DTYPE_BF16X3): fp32 tiles as in mode 1, but every gathered fragment is split into its (hi, lo) bf16 pair -- hi = RNE(x),
//         lo = RNE(x - hi) -- and the chunk's k-step is THREE MFMAs per accumulator tile, Gh.Xh + Gl.Xh + Gh.Xl: fp32-level accuracy
//         (the dropped Gl.Xl is 2^-16 relative per product) at 3 x 32 MFMA cycles per tile and chunk instead of 8 x 64.
//         Instruction-issue-bound (7.6 VALU per MFMA: both waves of a row / column of the 2 x 2 wave grid split the same tile).
//         Measured and NOT kept: the workgroup splitting each chunk ONCE into hi / lo bf16 planes in LDS (fragments then by
//         ds_read_b64_tr_b16, 4.6 VALU per MFMA) -- 5.5 ms against 4.6 for the fine pass: the extra LDS round trip (32 KB read +
//         32 KB written per chunk beside the 64 KB of transpose reads, 18 % bank conflicts) and its sync cost more than the
//         redundant conversions (tools/experiments/dw_x3_planes.cpp.txt, profiles/r04_x3_dw_forms.txt).
// MODES 5, 6, 7 (SN_DTYPE_BF16X3, what the library runs): the 256-wide slots of the training state hold the (hi, lo) pairs the forward
//         and the chain computed anyway (SPLIT tiles, RowStager) -- their fragments are four transpose reads and NO conversion; only the
//         operands that stay fp32 (embedded inputs, slot 9: dir_encoding's 128 columns and the head block) are split in registers.
//         5: A and B split.  6: A split, B fp32.  7: A fp32, B split.
template <int MT, int NT, int WM, int WN, int MODE, int LDSB = DW_LDS_BYTES>
SN_DEV void run_task(const Task& t, char* smem, int tid) {
  constexpr bool BF16 = MODE != 0;
  constexpr bool X3 = MODE >= 4;
  constexpr bool SA = MODE == 5 || MODE == 6, SB = MODE == 5 || MODE == 7;       // operand arrives as a split tile
  constexpr int EA = (MODE == 2 || MODE == 3) ? 2 : 4, EB = (MODE == 2) ? 2 : 4;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr int WA = WM * MT * 32, WB = WN * NT * 32;
  constexpr int A_BYTES = KB * WA * EA, B_BYTES = KB * WB * EB, BUF = A_BYTES + B_BYTES;
  // Ring depth: as many chunks as the 128 KB of LDS hold, at most 16 (fp32 256x256: 4 x 32 KB as before; bf16 state: 8 x 16 KB;
  // the narrow problems 16 x 5..9 KB).  One CU streams (NBUF-1) chunks per HBM latency: with the bf16 tiles at depth 4 the
  // narrow problems ran at 8 GB/s per CU (15 KB in flight), latency-bound far below their share of the HBM rate.
  // (round 3: the depth is no longer rounded down to a power of two -- a 12 KB chunk in 64 KB of LDS got a 4-deep ring, 36 KB in
  //  flight per workgroup, and the narrow bf16-state problems ran latency-bound at 4.7 TB/s; slots are tracked by running indices)
  constexpr int NFIT = LDSB / BUF >= 16 ? 16 : LDSB / BUF;
#ifndef SN_DW_SY_MIN
#define SN_DW_SY_MIN 8                               // rings at least this deep meet every 2nd chunk only (timing builds: 6)
#endif
  constexpr int SY0 = (BF16 && NFIT >= SN_DW_SY_MIN) ? (NFIT >= 16 ? 4 : 2) : 1;
  constexpr int NBUF = NFIT / SY0 * SY0;
  static_assert(NBUF >= 3 && NBUF * BUF <= LDSB, "ring fits the LDS allocation");
  // DMA instructions per thread per chunk.  A 32-wide A tile (variants 4/5) is only 128 16-byte pieces: waves 2,3 issue
  // none of it, so their vmcnt budget is one instruction per chunk smaller (the wait must be exact per wave).
  constexpr int CH_A = KB * WA * EA / 16, CH_B = KB * WB * EB / 16;
  static_assert(CH_B % 256 == 0 && CH_A % 64 == 0, "staging predicates must be wave-uniform");
  constexpr int IT_A = (CH_A + 255) / 256, IT_B = CH_B / 256, PART_A = (CH_A % 256) / 64;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int wr = wave / WN, wc = wave % WN;
  const int m0 = wr * MT * 32, n0 = wc * NT * 32;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
  float bsum[MT];
#pragma unroll
  for (int a = 0; a < MT; ++a) bsum[a] = 0.0f;

  const long k0 = t.k0, k1 = t.k1;
  if (k0 >= k1) return;
  RowStager<WA, EA, SA, X3> sa;
  RowStager<WB, EB, SB, X3> sb;
  sa.init(t.lda, tid);
  sb.init(t.ldb, tid);
  const int n_chunks = (int)((k1 - k0 + KB - 1) / KB);
  // bf16 modes with a deep ring meet at the barrier every SY-th chunk only (SY chunks are waited for and SY slots restaged per
  // sync point): at 16 points per chunk the fixed cost of a sync point (counted wait, barrier skew between the four waves) is
  // as long as the chunk's share of the HBM stream
  constexpr int SY = SY0;                                     // ring of 8..15: every 2nd chunk, ring of 16 (narrow problems): every 4th
  static_assert(NBUF % SY == 0 && NBUF >= 2 * SY, "sync period divides the ring");
  // prologue: NBUF-SY chunks in flight (chunks past the end are staged as clamped copies and never consumed)
#pragma unroll
  for (int c = 0; c < NBUF - SY; ++c) {
    sa.stage(t.a, t.lda, k0 + (long)c * KB, k1, smem + c * BUF, tid);
    sb.stage(t.b, t.ldb, k0 + (long)c * KB, k1, smem + c * BUF + A_BYTES, tid);
  }
  if (BF16) {
    // One 32x32x16 k-step per 16-point chunk.  The fragment gathers of chunk c are ISSUED in iteration c and consumed
    // (packed, summed, multiplied) in iteration c+1: their LDS latency hides behind the 16 MFMAs of chunk c-1 -- consumed in
    // place hipcc waits on them 19 times per chunk (measured: 2.5 k cycles per chunk against 512 of MFMA work).
    static_assert(KB == 16, "one 32x32x16 k-step per chunk");
    constexpr int RA_N = EA == 2 ? 4 : 8, RB_N = EB == 2 ? 4 : 8;
    unsigned ra0[MT][RA_N], rb0[NT][RB_N];           // fragments as read: packed bf16 pairs (transpose reads: hi [0..3], lo [4..7] of a split tile) or raw fp32 bits
    unsigned ra1[X3 ? MT : 1][RA_N], rb1[X3 ? NT : 1][RB_N];   // bf16x3: a second set, the gathers run one chunk further ahead (below)
    // transpose-read addresses of this lane inside a staged chunk: lane (q, G): feature block G & 1, point rows 8 (G >> 1) + (q >> 2)
    unsigned ta[MT], tb[NT];
    {
      const int q = lane & 15, G = lane >> 4;
      if (EA == 2) {
#pragma unroll
        for (int a = 0; a < MT; ++a) ta[a] = RowStager<WA, EA>::tr_offset(8 * (G >> 1) + (q >> 2), m0 + 32 * a + 16 * (G & 1) + 4 * (q & 3));
      }
      if (EB == 2) {
#pragma unroll
        for (int b = 0; b < NT; ++b) tb[b] = RowStager<WB, EB>::tr_offset(8 * (G >> 1) + (q >> 2), n0 + 32 * b + 16 * (G & 1) + 4 * (q & 3)) + A_BYTES;
      }
      if constexpr (SA) {
#pragma unroll
        for (int a = 0; a < MT; ++a) ta[a] = RowStager<WA, 4, 1>::tr_offset_split(8 * (G >> 1) + (q >> 2), m0 + 32 * a + 16 * (G & 1) + 4 * (q & 3));
      }
      if constexpr (SB) {
#pragma unroll
        for (int b = 0; b < NT; ++b) tb[b] = RowStager<WB, 4, 1>::tr_offset_split(8 * (G >> 1) + (q >> 2), n0 + 32 * b + 16 * (G & 1) + 4 * (q & 3)) + A_BYTES;
      }
    }
    const bool want_bias = t.bias != nullptr && wc == 0;
    auto sync_point = [&](int c, int slot_c) __attribute__((always_inline)) {
        // sync point, every SY-th chunk: chunks c .. c+SY-1 have landed for every wave (the NBUF-2*SY younger ones may still be
        // in flight), chunks c-SY .. c-1 are fully gathered -> their slots are restaged with chunks c+NBUF-SY .. c+NBUF-1
        const long k = k0 + (long)c * KB;
        if (PART_A != 0 && wave >= PART_A) wait_vmcnt<(NBUF - 2 * SY) * (IT_A - 1 + IT_B)>();
        else wait_vmcnt<(NBUF - 2 * SY) * (IT_A + IT_B)>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own gathers of chunk c-1 done before its slot is restaged
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int u = 0; u < SY; ++u) {
          const int sn = slot_c + NBUF - SY + u;     // < 2 NBUF
          char* bn = smem + (sn >= NBUF ? sn - NBUF : sn) * BUF;
          sa.stage(t.a, t.lda, k + (long)(NBUF - SY + u) * KB, k1, bn, tid);
          sb.stage(t.b, t.ldb, k + (long)(NBUF - SY + u) * KB, k1, bn + A_BYTES, tid);
        }
    };
    auto compute = [&](auto& ra, auto& rb, auto&& mid) __attribute__((always_inline)) {   // one chunk: pack, column sums, MFMAs (mid(): between the passes)
        dw_bf16x8 af[MT], bf[NT];
        dw_bf16x8 afl[X3 ? MT : 1], bfl[X3 ? NT : 1];  // bf16x3: the lo fragments
#pragma unroll
        for (int a = 0; a < MT; ++a) {
          dw_u32x4 q;
          if (SA) {                                  // split tile: hi and lo fragments as read
            dw_u32x4 ql;
#pragma unroll
            for (int w = 0; w < 4; ++w) { q[w] = ra[a][w]; ql[w] = ra[a][4 + w]; }
            // column sums for the bias gradient, by every wave (a wave-uniform `if (want_bias)` here splits the loop body into basic
            // blocks, and hipcc drains the NEXT chunk's fragment reads -- s_waitcnt lgkmcnt(0) -- at each of their joins)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(bsum[a]) : "v"(q[w]), "v"(0x3f803f80u));
              asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(bsum[a]) : "v"(ql[w]), "v"(0x3f803f80u));
            }
            afl[a] = __builtin_bit_cast(dw_bf16x8, ql);
          } else if (EA == 2) {                      // already the operand: 8 points of one feature, packed in k order
#pragma unroll
            for (int w = 0; w < 4; ++w) q[w] = ra[a][w];
            if (want_bias) {                         // column sums for the bias gradient: fp32 accumulation of the bf16 values
#pragma unroll
              for (int w = 0; w < 4; ++w) asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(bsum[a]) : "v"(q[w]), "v"(0x3f803f80u));
            }
          } else {
            float sum = 0.0f;
            dw_u32x4 ql;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              const float v0 = __builtin_bit_cast(float, ra[a][2 * w]), v1 = __builtin_bit_cast(float, ra[a][2 * w + 1]);
              q[w] = dw_pack2(v0, v1);
              if (X3) ql[w] = dw_pack2(v0 - __builtin_bit_cast(float, q[w] << 16), v1 - __builtin_bit_cast(float, q[w] & 0xffff0000u));
              sum += v0 + v1;
            }
            bsum[a] += sum;
            if (X3) afl[a] = __builtin_bit_cast(dw_bf16x8, ql);
          }
          af[a] = __builtin_bit_cast(dw_bf16x8, q);
        }
#pragma unroll
        for (int b = 0; b < NT; ++b) {
          dw_u32x4 q, ql;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            if (SB) { q[w] = rb[b][w]; ql[w] = rb[b][4 + w]; }
            else if (EB == 2) q[w] = rb[b][w];
            else {
              const float v0 = __builtin_bit_cast(float, rb[b][2 * w]), v1 = __builtin_bit_cast(float, rb[b][2 * w + 1]);
              q[w] = dw_pack2(v0, v1);
              if (X3) ql[w] = dw_pack2(v0 - __builtin_bit_cast(float, q[w] << 16), v1 - __builtin_bit_cast(float, q[w] & 0xffff0000u));
            }
          }
          bf[b] = __builtin_bit_cast(dw_bf16x8, q);
          if (X3) bfl[b] = __builtin_bit_cast(dw_bf16x8, ql);
        }
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
        mid();
        if (X3) {                                    // the two cross terms as passes of their own: an accumulator tile is revisited
                                                     // MT x NT MFMAs later, never by the next instruction
#pragma unroll
          for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afl[a], bf[b], acc[a][b], 0, 0, 0);
#pragma unroll
          for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfl[b], acc[a][b], 0, 0, 0);
        }
    };
    auto gather_a = [&](char* bc, auto& ra) __attribute__((always_inline)) {     // lane (i, h) takes rows 8h .. 8h+7 of its feature
#pragma unroll
        for (int a = 0; a < MT; ++a) {
          if (SA) {                                  // hi: two transpose reads as for a bf16 tile; lo: the same 8-byte groups of the neighbouring piece
            const dw_i16x4 h0 = tr_read(bc + ta[a]), h1 = tr_read(bc + ta[a] + 4 * WA * 4);
            const dw_i16x4 l0 = tr_read(bc + (ta[a] ^ 16u)), l1 = tr_read(bc + (ta[a] ^ 16u) + 4 * WA * 4);
            const dw_u32x4 uh = __builtin_bit_cast(dw_u32x4, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
            const dw_u32x4 ul = __builtin_bit_cast(dw_u32x4, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
            for (int w = 0; w < 4; ++w) { ra[a][w] = uh[w]; ra[a][4 + w] = ul[w]; }
          } else if (EA == 2) {                      // two transpose reads: points 8h .. 8h+3 and 8h+4 .. 8h+7 of this lane's feature
            const dw_i16x4 lo = tr_read(bc + ta[a]), hi = tr_read(bc + ta[a] + 4 * WA * EA);
            const dw_i16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            const dw_u32x4 u = __builtin_bit_cast(dw_u32x4, v);
#pragma unroll
            for (int w = 0; w < 4; ++w) ra[a][w] = u[w];
          } else {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) ra[a][jj] = reinterpret_cast<const unsigned*>(bc)[(8 * h + jj) * WA + m0 + i + 32 * a];
          }
        }
    };
    auto gather_b = [&](char* bc, auto& rb) __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < NT; ++b) {
          if (SB) {
            const dw_i16x4 h0 = tr_read(bc + tb[b]), h1 = tr_read(bc + tb[b] + 4 * WB * 4);
            const dw_i16x4 l0 = tr_read(bc + (tb[b] ^ 16u)), l1 = tr_read(bc + (tb[b] ^ 16u) + 4 * WB * 4);
            const dw_u32x4 uh = __builtin_bit_cast(dw_u32x4, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
            const dw_u32x4 ul = __builtin_bit_cast(dw_u32x4, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
            for (int w = 0; w < 4; ++w) { rb[b][w] = uh[w]; rb[b][4 + w] = ul[w]; }
          } else if (EB == 2) {
            const dw_i16x4 lo = tr_read(bc + tb[b]), hi = tr_read(bc + tb[b] + 4 * WB * EB);
            const dw_i16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            const dw_u32x4 u = __builtin_bit_cast(dw_u32x4, v);
#pragma unroll
            for (int w = 0; w < 4; ++w) rb[b][w] = u[w];
          } else {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) rb[b][jj] = reinterpret_cast<const unsigned*>(bc + A_BYTES)[(8 * h + jj) * WB + n0 + i + 32 * b];
          }
        }
    };
    int slot_c = 0;                                  // = c % NBUF (NBUF need not be a power of two)
    if constexpr (!X3) {
      for (int c = 0; c <= n_chunks; ++c, slot_c = (slot_c + 1 == NBUF) ? 0 : slot_c + 1) {
        if (c < n_chunks && c % SY == 0) sync_point(c, slot_c);
        if (c > 0) compute(ra0, rb0, [] {});         // chunk c-1
        if (c < n_chunks) { gather_a(smem + slot_c * BUF, ra0); gather_b(smem + slot_c * BUF, rb0); }
      }
    } else {
      // bf16x3: with split tiles a chunk is 32 transpose reads and 48 MFMAs and nothing else -- the reads of chunk c go out IN FRONT of
      // the MFMAs of chunk c-1 (two register sets, the loop unrolled by two), so their latency hides behind 1.5 k cycles of MFMA work
      // instead of behind the next sync point.  No control flow inside the loop that touches the accumulators (hipcc answers a
      // conditional compute with ~70 v_accvgpr_mov per iteration): step c = sync point, gathers of chunk c, MFMAs of chunk c-1 for
      // c = 1 .. n_chunks -- the gathers of "chunk n_chunks" read a staged clamp copy nobody consumes -- an odd step count is evened
      // out by one peeled step in front.
      // The reads are pinned by scheduling fences -- hipcc otherwise sinks them to their uses a step later (register pressure) and
      // waits for each with lgkmcnt(0) -- in two halves, the second between the MFMA passes: a counted LDS wait cannot leave more
      // than 15 reads outstanding.
      auto step = [&](int c, auto& ga, auto& gb, auto& ca, auto& cb) __attribute__((always_inline)) {
        if (c % SY == 0) sync_point(c, slot_c);
        char* bc = smem + slot_c * BUF;
        gather_a(bc, ga);
        __builtin_amdgcn_sched_barrier(0);
        compute(ca, cb, [&]() __attribute__((always_inline)) {           // chunk c-1
          __builtin_amdgcn_sched_barrier(0);
          gather_b(bc, gb);
          __builtin_amdgcn_sched_barrier(0);
        });
        slot_c = (slot_c + 1 == NBUF) ? 0 : slot_c + 1;
      };
      sync_point(0, 0);
      gather_a(smem, ra0);
      gather_b(smem, rb0);
      slot_c = 1;
      int c = 1;
      if (n_chunks & 1) {
        step(1, ra1, rb1, ra0, rb0);
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int w = 0; w < RA_N; ++w) ra0[a][w] = ra1[a][w];
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
          for (int w = 0; w < RB_N; ++w) rb0[b][w] = rb1[b][w];
        c = 2;
      }
      for (; c < n_chunks; c += 2) {
        step(c, ra1, rb1, ra0, rb0);
        step(c + 1, ra0, rb0, ra1, rb1);
      }
    }
  } else {
    int slot_c = 0;                                  // = c % NBUF
    for (int c = 0; c < n_chunks; ++c, slot_c = (slot_c + 1 == NBUF) ? 0 : slot_c + 1) {
      const long k = k0 + (long)c * KB;
      // chunk c was issued NBUF-1 chunks ago: everything but the (NBUF-2) younger chunks must have landed
      if (PART_A != 0 && wave >= PART_A) wait_vmcnt<(NBUF - 2) * (IT_A - 1 + IT_B)>();
      else wait_vmcnt<(NBUF - 2) * (IT_A + IT_B)>();
      __builtin_amdgcn_s_barrier();               // all waves' pieces of chunk c landed; chunk c-1 fully consumed
      {
        const int slot = slot_c == 0 ? NBUF - 1 : slot_c - 1;   // = slot of chunk c-1
        char* bn = smem + slot * BUF;
        sa.stage(t.a, t.lda, k + (long)(NBUF - 1) * KB, k1, bn, tid);
        sb.stage(t.b, t.ldb, k + (long)(NBUF - 1) * KB, k1, bn + A_BYTES, tid);
      }
      char* bc = smem + slot_c * BUF;
      const float* la = reinterpret_cast<const float*>(bc) + h * WA + m0 + i;
      const float* lb = reinterpret_cast<const float*>(bc + A_BYTES) + h * WB + n0 + i;
      // (measured and rejected: requesting the fragments of pair s+1 ahead of the MFMAs of pair s behind scheduling fences --
      //  hipcc reads them right in front of their first use -- made the 256x256 problems 3 % and the one-MFMA-per-pair narrow
      //  ones 30 % slower: the fences also pin the accumulator traffic of the builtin MFMAs.  A rolled loop over the k-step
      //  pairs with the next pair's fragments carried across the back edge was worse still: 6.0 against 5.1 ms on the
      //  256x256 problems, 2x on the narrow ones)
#pragma unroll
      for (int s = 0; s < KB / 2; ++s) {
        float av[MT], bv[NT];
#pragma unroll
        for (int a = 0; a < MT; ++a) av[a] = la[(2 * s) * WA + 32 * a];
#pragma unroll
        for (int b = 0; b < NT; ++b) bv[b] = lb[(2 * s) * WB + 32 * b];
#pragma unroll
        for (int a = 0; a < MT; ++a) {
          bsum[a] += av[a];
#pragma unroll
          for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
      }
    }
  }
  wait_vmcnt<0>();                               // drain the over-issued tail chunks before the LDS is released
  // epilogue: accumulator (row = (r&3)+8(r>>2)+4h, col = i) -> c[m][n], 128 B per lane-half per store
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
        t.c[(long)m * t.ldc + n0 + 32 * b + i] = acc[a][b][r];
      }
  if (t.bias != nullptr && wc == 0) {
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      const float v = bsum[a] + __shfl_xor(bsum[a], 32, 64);
      if (h == 0) t.bias[m0 + 32 * a + i] = v;
    }
  }
}

constexpr int DW_KERNEL_LDS_BYTES = 163840;     // the launch asks for the whole 160 KB; the bf16x3 modes use it (a 5th chunk in flight), the others 128 KB
__global__ void __launch_bounds__(256) dw_kernel(const Task* __restrict__ tasks, const Plan plan) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // (bf16x3 plans run several workgroups per CU in turn: last problems first -- the narrow ones, whose tasks are the longest)
  const int wg = (tasks == nullptr && (plan.p[0].variant & 0x400)) ? plan.n_tasks - 1 - (int)blockIdx.x : (int)blockIdx.x;
  const Task t = tasks != nullptr ? tasks[blockIdx.x] : task_of(plan, wg);
  const int tid = threadIdx.x;
  if (t.variant & 0x100) {
    const int mode = (t.variant & 0x400) ? 4 : (t.variant & 0x200) ? 2 : 1;     // 0x200: G and the activations are stored as bf16; 0x400: bf16x3 on fp32 state
#define SN_DW_CASES(MODE_, MODE_EMB_)                                   \
    switch (t.variant & 0xff) {                                         \
      case 0: run_task<4, 4, 2, 2, MODE_>(t, smem, tid); break;         \
      case 1: run_task<4, 1, 2, 2, MODE_EMB_>(t, smem, tid); break;     \
      case 2: run_task<2, 4, 2, 2, MODE_>(t, smem, tid); break;         \
      case 3: run_task<2, 1, 2, 2, MODE_EMB_>(t, smem, tid); break;     \
      case 4: run_task<1, 2, 1, 4, MODE_>(t, smem, tid); break;         \
      default: run_task<1, 1, 1, 4, MODE_>(t, smem, tid); break;        \
    }
    // variants 1 / 3 contract with the embedded inputs, which stay fp32 in every mode
    if (mode == 1) { SN_DW_CASES(1, 1) } else if (mode == 4) {
      // bf16x3: G slots 0..8 and acts slots 0..8 are split tiles, slot 9 (dir_encoding's 128 columns + the head block, h2) and emb are fp32
      switch (t.variant & 0xff) {
        case 0: run_task<4, 4, 2, 2, 5, DW_KERNEL_LDS_BYTES>(t, smem, tid); break;         // G[l] x acts[l-1]   (5 chunks of 32 KB)
        case 1: run_task<4, 1, 2, 2, 6, DW_KERNEL_LDS_BYTES>(t, smem, tid); break;         // G[0], G[4] x emb
        case 2: run_task<2, 4, 2, 2, 7, DW_KERNEL_LDS_BYTES>(t, smem, tid); break;         // G[9][:, :128] x acts[8]
        case 3: run_task<2, 1, 2, 2, 4, DW_KERNEL_LDS_BYTES>(t, smem, tid); break;         // G[9][:, :128] x emb
        case 4: run_task<1, 2, 1, 4, 7, DW_KERNEL_LDS_BYTES>(t, smem, tid); break;         // head block x acts[7]
        default: run_task<1, 1, 1, 4, 4, DW_KERNEL_LDS_BYTES>(t, smem, tid); break;        // head block x acts[9]
      }
    } else { SN_DW_CASES(2, 3) }
#undef SN_DW_CASES
    return;
  }
  switch (t.variant) {
    case 0: run_task<4, 4, 2, 2, 0>(t, smem, tid); break;
    case 1: run_task<4, 1, 2, 2, 0>(t, smem, tid); break;
    case 2: run_task<2, 4, 2, 2, 0>(t, smem, tid); break;
    case 3: run_task<2, 1, 2, 2, 0>(t, smem, tid); break;
    case 4: run_task<1, 2, 1, 4, 0>(t, smem, tid); break;
    default: run_task<1, 1, 1, 4, 0>(t, smem, tid); break;
  }
}

// The narrow problems of the bf16-state mode in a kernel of their own: half the LDS and at most 256 registers per wave, so that
// TWO workgroups share a CU -- these problems are latency-bound (few MFMAs per chunk), a second set of waves fills the stalls.
#ifndef SN_DW_NARROW_LDS
#define SN_DW_NARROW_LDS 81920                      // two workgroups share the CU's 160 KB (timing builds: 65536, 73728)
#endif
constexpr int DW_NARROW_LDS_BYTES = SN_DW_NARROW_LDS;
#ifndef SN_DW_NARROW_WGS
#define SN_DW_NARROW_WGS 2
#endif
__global__ void __launch_bounds__(256, SN_DW_NARROW_WGS) dw_narrow_bf16_kernel(const Plan plan) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Task t = task_of(plan, (int)blockIdx.x);
  const int tid = threadIdx.x;
  switch (t.variant & 0xff) {                    // variants 1 / 3 contract with the embedded inputs, which stay fp32 (MODE 3)
    case 1: run_task<4, 1, 2, 2, 3, DW_NARROW_LDS_BYTES>(t, smem, tid); break;
    case 2: run_task<2, 4, 2, 2, 2, DW_NARROW_LDS_BYTES>(t, smem, tid); break;
    case 3: run_task<2, 1, 2, 2, 3, DW_NARROW_LDS_BYTES>(t, smem, tid); break;
    case 4: run_task<1, 2, 1, 4, 2, DW_NARROW_LDS_BYTES>(t, smem, tid); break;
    default: run_task<1, 1, 1, 4, 2, DW_NARROW_LDS_BYTES>(t, smem, tid); break;
  }
}

// ... and the narrow problems of the fp32 step the same way (round 3): their compiler-scheduled loop leaves the MFMA pipe 67 % busy
// with the waves parked 21 % of the time; none of the narrow variants needs more than 128 accumulator registers.
#ifndef SN_DW_NARROW_F32_2WG
#define SN_DW_NARROW_F32_2WG 1
#endif
constexpr int DW_NARROW_F32_LDS_BYTES = 81920;
__global__ void __launch_bounds__(256, 2) dw_narrow_f32_kernel(const Plan plan) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Task t = task_of(plan, (int)blockIdx.x);
  const int tid = threadIdx.x;
  switch (t.variant & 0xff) {
    case 1: run_task<4, 1, 2, 2, 0, DW_NARROW_F32_LDS_BYTES>(t, smem, tid); break;
    case 2: run_task<2, 4, 2, 2, 0, DW_NARROW_F32_LDS_BYTES>(t, smem, tid); break;
    case 3: run_task<2, 1, 2, 2, 0, DW_NARROW_F32_LDS_BYTES>(t, smem, tid); break;
    case 4: run_task<1, 2, 1, 4, 0, DW_NARROW_F32_LDS_BYTES>(t, smem, tid); break;
    default: run_task<1, 1, 1, 4, 0, DW_NARROW_F32_LDS_BYTES>(t, smem, tid); break;
  }
}

// ---- finish: deterministic sum of the K-split partials, written straight into the parameter-shaped gradients -------------
struct Seg {                                    // dst[r][dst_col0 + c] (+)= sum_j src[j*stride + (src_row0 + r)*src_ld + src_col0 + c]
  float* dst;
  const float* src;
  int dst_ld, dst_col0, rows, cols, src_ld, src_row0, src_col0, ns, stride, first;   // first = prefix sum of rows*cols
  int perm, pad;                                  // 1 / 2: source column of c = K-slot position of xyz / dir embedding column c
};
// SN_DTYPE_EMB_BF16: the training forward stores the embedded inputs as the bf16 MFMA operands it built (K-slot order: lane half h
// owns positions 32 h + e of the xyz embedding, 16 h + e of the direction embedding -- sn_mlp_common.h store_emb_xyz / _dir give the
// column each slot e holds), so the 64-wide gradient blocks come out in that order and are permuted back here
SN_DEV int emb_xyz_pos(int c) {                   // column c of Embedding(3, 10) (63 columns) -> position in the stored row
  if (c < 3) return c == 0 ? 30 : c == 1 ? 31 : 62;
  const int k = (c - 3) % 30, h = (c - 3) / 30;
  return 32 * h + 2 * (3 * (k / 6) + k % 3) + (k % 6) / 3;
}
SN_DEV int emb_dir_pos(int c) {                   // column c of Embedding(3, 4) (27 columns)
  if (c < 3) return c == 0 ? 12 : c == 1 ? 13 : 28;
  const int k = (c - 3) % 12, h = (c - 3) / 12;
  return 16 * h + 2 * (3 * (k / 6) + k % 3) + (k % 6) / 3;
}
constexpr int MAX_SEGS = 32;
struct Segs {
  Seg s[MAX_SEGS];
  int n_segs, total, accumulate, pad;
};

__global__ void __launch_bounds__(256) dw_finish_kernel(const Segs segs) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= segs.total) return;
  int q = 0;
#pragma unroll 1
  for (int i = 1; i < segs.n_segs; ++i) q = (e >= segs.s[i].first) ? i : q;
  const Seg& g = segs.s[q];
  const int l = e - g.first;
  const int r = l / g.cols, c = l - r * g.cols;
  const int sc = g.perm == 0 ? c : g.perm == 1 ? emb_xyz_pos(c) : emb_dir_pos(c);
  const float* src = g.src + (long)(g.src_row0 + r) * g.src_ld + g.src_col0 + sc;
  // fixed summation order (run-to-run deterministic), but the loads of eight partials are issued together: as a dependent chain
  // of ns ~ 32 strided loads per thread the kernel ran at 1.5 TB/s (50 us per network, 2.5 % of a bf16 training step)
  float acc = 0.0f;
  const long st = g.stride;
  int j = 0;
  for (; j + 8 <= g.ns; j += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(src + (long)(j + u) * st);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; j < g.ns; ++j) acc += src[(long)j * st];
  float* d = g.dst + (long)r * g.dst_ld + g.dst_col0 + c;
  *d = segs.accumulate ? *d + acc : acc;
}

// ---- host: the plan of one network (the K-split cost model that used to live in sinnerf_amd/autograd.py) -------------------
struct VariantInfo { int m, n; };
static const VariantInfo VARIANTS[8] = {{256, 256}, {256, 64}, {128, 256}, {128, 64}, {32, 256}, {32, 128}, {256, 64}, {128, 64}};
// cost of one point of a K-range on one CU (cycles, variant 0 = 512): max(MFMA issue time of the wave block, tile bytes over
// the per-CU streaming rate) -- measured per mode (tools/dw_time.py): the narrow problems are DMA-bound, and splitting by
// FLOPs alone left the 32x128 problem streaming 168 MB through a single CU
static const int COST_F32[8] = {512, 161, 260, 95, 101, 59, 0, 0};
static const int COST_BF16[8] = {512, 189, 226, 126, 138, 125, 0, 0};          // bf16 operands, fp32 state
// bf16x3 on the fp32 state (run_task MODE 4): instruction-issue-bound -- the cost of a point follows the tiles a wave splits and its
// MFMAs, not the bytes (tools/dw_x3_time.py: each variant's tasks alone; with the bf16 table above the 128 x 256 problem's 11
// workgroups ran 3.07 ms while the 208 of the eight 256 x 256 problems were done after 2.34)
static const int COST_X3[8] = {512, 182, 231, 162, 123, 117, 0, 0};
// bf16 operands, bf16 state (transpose-read fragments, a sync point every 2nd / 4th chunk): the 256x256 problems run at their
// share of the HBM rate (52 ns per point per CU = 1 KB / 19.7 GB/s), the narrower ones at 19..35 ns per point
#ifndef SN_DW_COST_STATE
#define SN_DW_COST_STATE 512, 343, 348, 226, 261, 145, 290, 174     // (narrow shapes of the bf16-emb step, 2 / 4 / 5 / 6 / 7: their bytes per
                                                                    //  point x 0.453 -- tools/r3_run32.sh: 0.4 % on the step against the round-2 table)
#endif
static const int COST_BF16_STATE[8] = {SN_DW_COST_STATE};
constexpr int TARGET_WGS = 256;                 // one workgroup per CU
#ifndef SN_DW_X3_WAVES_OF_WGS
#define SN_DW_X3_WAVES_OF_WGS 4                 // bf16x3: several workgroups per CU IN TURN (160 KB of LDS each): the tail of an imperfect K-split is one short task
#endif

struct HostPlan {
  Plan plan;                                    // all problems (task numbering of the single-launch modes)
  long c_off[MAX_PROBS], b_off[MAX_PROBS];      // workspace byte offsets of the partial buffers (b_off < 0: no bias)
  long bytes;
  bool two_launches;                            // fp32: group 0 = the 256x256 problems (sn_dw_f32.hip), group 1 = the rest
  int group[MAX_PROBS], first_in_group[MAX_PROBS];
};
// the problems of one group as a plan of their own (task numbering restarts at 0)
static Plan group_plan(const HostPlan& hp, int g) {
  Plan out;
  out.P = hp.plan.P;
  out.n_probs = 0;
  out.n_tasks = 0;
  for (int i = 0; i < hp.plan.n_probs; ++i) {
    if (hp.group[i] != g) continue;
    Prob q = hp.plan.p[i];
    q.first = hp.first_in_group[i];
    out.p[out.n_probs++] = q;
    out.n_tasks += q.ns;
  }
  return out;
}
// problems in the order build_plan emits them
enum { W0 = 0, W1 = 1, W2 = 2, W3 = 3, W4 = 4, W4E = 5, W5 = 6, W6 = 7, W7 = 8, WF = 9, WD = 10, WDE = 11, SIG = 12, RGB = 13 };

// dtype: 0 fp32, 1 bf16 operands / fp32 state, 2 bf16 operands / bf16 state, 3 bf16x3 (hi/lo split: fp32-level) / fp32 state.
// Pointers may be null (size query).
static void build_plan(HostPlan& hp, const char* acts, const char* emb, const char* G, long rows, int dtype, bool emb16 = false) {
  const long es = dtype == 2 ? 2 : 4;           // element size of acts / G (emb: fp32, or bf16 in K-slot order with emb16)
  const int v_e1 = emb16 ? 6 : 1, v_e3 = emb16 ? 7 : 3;
  const long ees = emb16 ? 2 : 4;
  const long slot = rows * 256 * es;
  const int flags = (dtype >= 1 ? 0x100 : 0) | (dtype == 2 ? 0x200 : 0) | (dtype == 3 ? 0x400 : 0);
  const int* cost = dtype == 0 ? COST_F32 : dtype == 1 ? COST_BF16 : dtype == 3 ? COST_X3 : COST_BF16_STATE;
  struct P { const char* a; const char* b; int lda, ldb, var; bool bias; };
  P pr[MAX_PROBS];
  int n = 0;
  auto Gs = [&](int i, int col) { return G + i * slot + col * es; };
  auto As = [&](int i) { return acts + i * slot; };
  for (int i = 0; i < 8; ++i) {                                   // xyz_encoding_{i+1}
    if (i == 0) pr[n++] = {Gs(0, 0), emb, 256, 128, v_e1, true};
    else {
      pr[n++] = {Gs(i, 0), As(i - 1), 256, 256, 0, true};
      if (i == 4) pr[n++] = {Gs(4, 0), emb, 256, 128, v_e1, false};  // skip: cat([input_xyz, h4])  nerf.py:133
    }
  }
  pr[n++] = {Gs(8, 0), As(7), 256, 256, 0, true};                 // xyz_encoding_final
  pr[n++] = {Gs(9, 0), As(8), 256, 256, 2, true};                 // dir_encoding[:, :256]
  pr[n++] = {Gs(9, 0), emb + 64 * ees, 256, 128, v_e3, false};         // dir_encoding[:, 256:]
  pr[n++] = {Gs(9, 128), As(7), 256, 256, 4, false};              // sigma (nerf.py:136): row 3 of the 32-wide head block
  pr[n++] = {Gs(9, 128), As(9), 256, 256, 5, true};               // rgb (nerf.py:144): rows 0..2; bias = [g_rgb(3), g_sigma(1)]
  // fp32: the 256x256 problems run in their own launch (sn_dw_f32.hip, hand-scheduled inner loop), the narrow ones in a second
  // launch of the kernel above -- each group is K-split over one workgroup per CU by itself (group[i]: 0 / 1; other modes: one
  // group, one launch)
  // ... and so do the bf16-state 256x256 problems (sn_dw_bf16.hip)
  int group[MAX_PROBS];
  const bool split_launch = dtype == 0 || dtype == 2;
  for (int i = 0; i < n; ++i) group[i] = (split_launch && pr[i].var != 0) ? 1 : 0;
  hp.two_launches = split_launch;
  int splits[MAX_PROBS];
  for (int gsel = 0; gsel < 2; ++gsel) {
    double tot = 0;
    for (int i = 0; i < n; ++i) if (group[i] == gsel) tot += cost[pr[i].var];
    if (tot == 0) continue;
    // (the bf16-state narrow problems run two workgroups per CU: dw_narrow_bf16_kernel)
    const int target = (gsel == 1 && dtype == 2 && SN_DW_NARROW_2WG) ? SN_DW_NARROW_WGS * TARGET_WGS
                       : (gsel == 1 && dtype == 0 && SN_DW_NARROW_F32_2WG) ? 2 * TARGET_WGS
                       : dtype == 3 ? SN_DW_X3_WAVES_OF_WGS * TARGET_WGS : TARGET_WGS;
    double frac[MAX_PROBS];
    int sum = 0;
    for (int i = 0; i < n; ++i) {
      if (group[i] != gsel) continue;
      const double ideal = target * cost[pr[i].var] / tot;
      splits[i] = (int)ideal < 1 ? 1 : (int)ideal;
      frac[i] = ideal - (int)ideal;
      sum += splits[i];
    }
    bool used[MAX_PROBS] = {};
    while (sum < target) {                                        // largest remainders get the slack
      int best = -1;
      for (int i = 0; i < n; ++i) if (group[i] == gsel && !used[i] && (best < 0 || frac[i] > frac[best])) best = i;
      if (best < 0) break;
      used[best] = true; ++splits[best]; ++sum;
    }
  }
  const long max_split = rows / (4 * KB) < 1 ? 1 : rows / (4 * KB);
  long off = 0;
  int first = 0, first_g[2] = {0, 0};
  hp.plan.P = rows;
  hp.plan.n_probs = n;
  for (int i = 0; i < n; ++i) {
    long ns = splits[i] < max_split ? splits[i] : max_split;
    long per = (rows + ns - 1) / ns;
    const long gran = (dtype == 2 && pr[i].var == 0) ? 2 * KB : KB;      // sn_dw_bf16.hip consumes chunk PAIRS (an odd tail
    per = (per + gran - 1) / gran * gran;                                // chunk of the last K-range costs a statement of its own)
    ns = (rows + per - 1) / per;
    const VariantInfo v = VARIANTS[pr[i].var];
    Prob& q = hp.plan.p[i];
    q.a = pr[i].a; q.b = pr[i].b; q.c = nullptr; q.bias = nullptr;
    q.lda = pr[i].lda; q.ldb = pr[i].ldb; q.ldc = v.n; q.variant = pr[i].var | flags;
    q.ns = (int)ns; q.per = (int)per; q.first = first; q.m = v.m;
    first += (int)ns;
    hp.group[i] = group[i];
    hp.first_in_group[i] = first_g[group[i]];
    first_g[group[i]] += (int)ns;
    hp.c_off[i] = off; off += ns * v.m * v.n * 4;
    hp.b_off[i] = pr[i].bias ? off : -1;
    if (pr[i].bias) off += ns * v.m * 4;
  }
  hp.plan.n_tasks = first;
  hp.bytes = (off + 255) / 256 * 256;
}

}  // namespace snd

extern "C" int sn_dw_f32_asm_launch(const snd::Plan* plan_host, hipStream_t stream);      // sn_dw_f32.hip
extern "C" int sn_dw_bf16_asm_launch(const snd::Plan* plan_host, hipStream_t stream);     // sn_dw_bf16.hip
extern "C" int sn_dw_narrow_bf16_asm_launch(const snd::Plan* plan_host, hipStream_t stream);   // sn_dw_narrow_bf16.hip
#ifndef SN_DW_NARROW_ASM
#define SN_DW_NARROW_ASM 1      // bf16-state narrow problems on the generated instruction streams (0: comparison build)
#endif
// SINNERF_DW_NARROW_COMPILER=1 in the environment keeps the compiler-scheduled narrow kernel (A/B runs; read once)
static bool narrow_compiler_scheduled() {
  static const bool v = [] { const char* e = getenv("SINNERF_DW_NARROW_COMPILER"); return e != nullptr && e[0] == '1'; }();
  return v;
}

extern "C" long sn_weight_grads_workspace_bytes_impl(long slot_rows, int dtype, int emb16) {
  // the same refusal as the launch below: a caller that asks here first (sinnerf_amd/autograd.py does) never stores a bf16 emb
  // that the backward cannot read
  if (emb16 && (dtype != 2 || !SN_DW_NARROW_ASM || narrow_compiler_scheduled())) return -4;
  snd::HostPlan hp;
  snd::build_plan(hp, nullptr, nullptr, nullptr, slot_rows, dtype, emb16 != 0);
  return hp.bytes;
}

extern "C" int sn_weight_grads_launch(const void* acts, const float* emb, const void* G, long slot_rows, int dtype, int emb16,
                                      void* workspace, float* const* grads, int accumulate, hipStream_t stream) {
  using namespace snd;
  if (emb16 && (dtype != 2 || !SN_DW_NARROW_ASM || narrow_compiler_scheduled())) return -4;     // SN_E_UNSUPPORTED: only the generated narrow kernel reads it
  HostPlan hp;
  build_plan(hp, (const char*)acts, (const char*)emb, (const char*)G, slot_rows, dtype, emb16 != 0);
  char* ws = (char*)workspace;
  for (int i = 0; i < hp.plan.n_probs; ++i) {
    hp.plan.p[i].c = (float*)(ws + hp.c_off[i]);
    hp.plan.p[i].bias = hp.b_off[i] >= 0 ? (float*)(ws + hp.b_off[i]) : nullptr;
  }
  SN_ENSURE_DYN_LDS(dw_kernel, DW_KERNEL_LDS_BYTES);
  int rc;
  if (hp.two_launches) {
    const Plan pa = group_plan(hp, 0), pb = group_plan(hp, 1);
    rc = dtype == 0 ? sn_dw_f32_asm_launch(&pa, stream) : sn_dw_bf16_asm_launch(&pa, stream);
    if (rc) return rc;
    if (dtype == 2 && SN_DW_NARROW_ASM && !narrow_compiler_scheduled()) {
      rc = sn_dw_narrow_bf16_asm_launch(&pb, stream);
      if (rc) return rc;
    } else if (dtype == 2 && SN_DW_NARROW_2WG) {
      SN_ENSURE_DYN_LDS(dw_narrow_bf16_kernel, DW_NARROW_LDS_BYTES);
      hipLaunchKernelGGL(dw_narrow_bf16_kernel, dim3((unsigned)pb.n_tasks), dim3(256), DW_NARROW_LDS_BYTES, stream, pb);
    } else if (dtype == 0 && SN_DW_NARROW_F32_2WG) {
      SN_ENSURE_DYN_LDS(dw_narrow_f32_kernel, DW_NARROW_F32_LDS_BYTES);
      hipLaunchKernelGGL(dw_narrow_f32_kernel, dim3((unsigned)pb.n_tasks), dim3(256), DW_NARROW_F32_LDS_BYTES, stream, pb);
    } else {
      hipLaunchKernelGGL(dw_kernel, dim3((unsigned)pb.n_tasks), dim3(256), DW_KERNEL_LDS_BYTES, stream, (const Task*)nullptr, pb);
    }
  } else {
    hipLaunchKernelGGL(dw_kernel, dim3((unsigned)hp.plan.n_tasks), dim3(256), DW_KERNEL_LDS_BYTES, stream, (const Task*)nullptr, hp.plan);
  }
  rc = (int)hipGetLastError();
  if (rc) return rc;
  Segs sg;
  int n = 0, total = 0;
  auto seg = [&](float* dst, int dst_ld, int dst_col0, int rows, int cols, int prob, bool from_bias, int src_row0, int src_col0, int perm = 0) {
    if (dst == nullptr) return;
    const Prob& q = hp.plan.p[prob];
    Seg& g = sg.s[n++];
    g.dst = dst; g.dst_ld = dst_ld; g.dst_col0 = dst_col0; g.rows = rows; g.cols = cols;
    g.src = from_bias ? q.bias : q.c;
    g.src_ld = from_bias ? 1 : q.ldc; g.src_row0 = src_row0; g.src_col0 = src_col0;
    g.ns = q.ns; g.stride = from_bias ? q.m : q.m * q.ldc;
    g.perm = emb16 ? perm : 0; g.pad = 0;
    g.first = total; total += rows * cols;
  };
  const int wl[8] = {W0, W1, W2, W3, W4, W5, W6, W7};
  for (int i = 0; i < 8; ++i) {
    if (i == 0) seg(grads[0], 63, 0, 256, 63, W0, false, 0, 0, 1);
    else if (i == 4) {
      seg(grads[8], 319, 0, 256, 63, W4E, false, 0, 0, 1);         // cat([input_xyz, h4]): embedded columns first
      seg(grads[8], 319, 63, 256, 256, W4, false, 0, 0);
    } else seg(grads[2 * i], 256, 0, 256, 256, wl[i], false, 0, 0);
    seg(grads[2 * i + 1], 1, 0, 256, 1, wl[i], true, 0, 0);
  }
  seg(grads[16], 256, 0, 256, 256, WF, false, 0, 0);
  seg(grads[17], 1, 0, 256, 1, WF, true, 0, 0);
  seg(grads[18], 283, 0, 128, 256, WD, false, 0, 0);
  seg(grads[18], 283, 256, 128, 27, WDE, false, 0, 0, 2);
  seg(grads[19], 1, 0, 128, 1, WD, true, 0, 0);
  seg(grads[20], 256, 0, 1, 256, SIG, false, 3, 0);                // sigma.weight (1, 256) = row 3 of the head block
  seg(grads[21], 1, 0, 1, 1, RGB, true, 3, 0);                     // sigma.bias = column sum of g_sigma
  seg(grads[22], 128, 0, 3, 128, RGB, false, 0, 0);                // rgb.weight (3, 128)
  seg(grads[23], 1, 0, 3, 1, RGB, true, 0, 0);
  if (n == 0) return 0;
  sg.n_segs = n; sg.total = total; sg.accumulate = accumulate; sg.pad = 0;
  hipLaunchKernelGGL(dw_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, sg);
  return (int)hipGetLastError();
}

extern "C" int sn_dw_launch(const void* tasks, int n_tasks, hipStream_t stream) {
  using namespace snd;
  if (n_tasks <= 0) return 0;
  static_assert(sizeof(Task) == 64, "Task must be 64 bytes (host packs it as 8 x int64)");
  SN_ENSURE_DYN_LDS(dw_kernel, DW_KERNEL_LDS_BYTES);
  Plan none;
  none.n_probs = 0; none.n_tasks = 0; none.P = 0;
  hipLaunchKernelGGL(dw_kernel, dim3((unsigned)n_tasks), dim3(256), DW_KERNEL_LDS_BYTES, stream,
                     reinterpret_cast<const Task*>(tasks), none);
  return (int)hipGetLastError();
}
