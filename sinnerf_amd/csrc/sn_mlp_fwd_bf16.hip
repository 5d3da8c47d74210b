// sn_mlp_fwd_bf16.hip -- fused NeRF MLP forward for gfx950, bf16-operand / fp32-accumulate path
// (v_mfma_f32_32x32x16_bf16: 16x the fp32 MFMA rate; BASELINE configs 3 and 5, north_star "MFMA bf16 for the dense
// W.x contractions").  Same algorithm and reference lines as sn_mlp_fwd.hip; what changes with the data type:
//
// * a wave owns TWO 32-point column tiles (PT = 2): every A fragment read from LDS (one ds_read_b128 = 8 bf16 of 32
//   weight rows) feeds two MFMAs on two independent accumulators;
// * activations live in registers as packed bf16x8 B operands: the fp32 accumulators of an output tile are biased,
//   activated and converted (v_cvt_pk_bf16_f32, RNE) straight into the two k-steps they form for the next layer
//   (K-slot order of sn_layout.h: k-step 2t holds accumulator registers 0..7 of tile t, k-step 2t+1 registers 8..15);
// * the two narrow heads never touch bf16: sigma (nerf.py:136) is accumulated on the VALU in fp32 from the ReLU'd fp32
//   accumulators of layer 8 while they are being packed, rgb (nerf.py:144) likewise from the fp32 ShiftedSoftplus
//   outputs of dir_encoding -- so the 128-wide dir activation is never materialised at all;
// * weights stream as bf16 slabs (K*64 bytes, a quarter of the fp32 LDS traffic per point), biases stay fp32.
//
// REGISTER PLAN.  At bf16 rate an MFMA is 32 cycles and the instruction mix, not the matrix pipe, decides the speed
// (rocprofv3 SQ_INSTS_*: the first version of this kernel issued 9.6 non-MFMA instructions per MFMA, 6 VALU per
// activation value: hipcc keeps every builtin-MFMA accumulator in AGPRs -- v_accvgpr_read before each VALU use,
// v_accvgpr_mov for each accumulator hand-over -- and shuffles 4-register B tuples between the two register files).
// So the MFMAs are inline asm and the accumulation-register half of the file is managed BY HAND:
//     a[0:127], a[128:255]   two activation sets.  A layer reads its B operands from one set (MFMA reads B from AGPRs
//                            at no cost) while its epilogues v_accvgpr_write the next layer's into the other; the roles
//                            swap every layer, nothing is ever copied.  The AGPR numbers are immediates in the asm
//                            text, the compiler never sees these registers (it is told a255 is clobbered so the kernel
//                            is sized for all 256; tools/check_agpr.py verifies it allocated none itself).
//     VGPRs                  two accumulator sets (C/D) used alternately by consecutive slabs -- the epilogue reads them
//                            with plain VALU instructions, the bias of slab s+1 is ds_read straight into the set slab
//                            s-1 just vacated --, the A-fragment ring, the xyz / dir embeddings (B operands of layer
//                            0, the skip layer and dir_encoding).
// ReLU of a hidden layer is done AFTER the conversion on the packed pair (v_pk_max_i16 with 0: a negative bf16 is a
// negative int16), one instruction per two values; only layer 8, whose fp32 ReLU output feeds the sigma head, uses v_max.
// => 1.5 VALU instructions per activation value (cvt_pk, pk_max, accvgpr_write per pair).
// The compiler does not know the asm is an MFMA, so the MFMA -> VALU-read hazard is kept by construction: an accumulator
// set is read (a) by the deferred epilogue, which sits behind two MFMAs of the NEXT slab (64 cycles) and is pinned there
// by sched_barrier, or (b) at a layer end behind an explicit s_nop run.
//
// Persistent workgroups, 3-slot weight ring with a mid-slab barrier, compile-time DMA piece counts: sn_mlp_pipe.h.
#include "sn_mlp_bf16.h"
#ifndef SN_BF16_COUNTED
#define SN_BF16_COUNTED 1      // counted vmcnt at the sync points of the bf16-state training forward (0: comparison build)
#endif

namespace snk {

// STORE: training forward with bf16 contractions -- additionally writes the fp32 activations of every layer
// (acts[10][slot_rows][256]: the fp32 values BEFORE their bf16 rounding) and the fp32 embedded inputs (emb[slot_rows][128])
// that the fp32 backward (sn_mlp_bwd.hip, sn_dw.hip) consumes: bf16 forward + fp32 backward, i.e. mixed precision.
// STORE 2: the activations are stored as bf16 (acts is then a bf16 array of the same shape: exactly the values the next
// layer consumed); emb stays fp32.  The unused half of slot 9 (dir_encoding is 128 wide: columns 128..255, 256 B per point)
// receives the ReLU sign words of layers 1..8 (sn_mlp_bf16.h epi_relu_bits): for the 64 points p_wave .. p_wave+63 of a wave,
// the word of (layer l, output tile t) sits in row p_wave + 8 l + t, dword `lane` -- 256 B per point in total, what the
// backward chain reads instead of the 4 KB of activations.
template <bool SIGMA_ONLY, int INPUT_MODE, int STORE>
__global__ void __launch_bounds__(256)
mlp_fwd_bf16_kernel(const char* __restrict__ blob, const float* __restrict__ in0, const float* __restrict__ in1,
                    long P, int S, float* __restrict__ out, float* __restrict__ acts, float* __restrict__ emb,
                    long slot_rows) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds_bias = reinterpret_cast<float*>(smem);
  const float* lds_aux = lds_bias + snl::BIAS_FLOATS;
  asm volatile("" ::: "a0", "a255");             // size the kernel for the whole hand-managed AGPR file

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int j = lane & 31;
  const int h = lane >> 5;
  constexpr int TILE_PTS = 4 * PT * 32;                      // 256 points per workgroup pass
  const long n_tiles = (P + TILE_PTS - 1) / TILE_PTS;
  const long my_tiles = (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;

  RingB ring;
  ring.blob = blob;
  ring.gnext = blob;
  ring.base = smem + TAIL_LDS_BYTES;
  ring.n_used = SIGMA_ONLY ? snl::SLAB_FIN : snl::N_SLABS;
  ring.stage_id = 0;
  ring.stage_slot = 0;
  ring.remaining = my_tiles * ring.n_used;       // (only the prologue staging checks it)
  ring.tid = tid;
  ring.wbase = __builtin_amdgcn_readfirstlane((tid & ~63) * 16);
  ring.pieces = 0; ring.piece = 0; ring.slab_bytes = 0;
  ring.stage_whole();
  ring.stage_whole();
  {
    const float4* gb = reinterpret_cast<const float4*>(blob + snl::bias_byte_offset(snl::DT_BF16));
    float4* lb = reinterpret_cast<float4*>(lds_bias);
    for (int i = tid; i < snl::TAIL_FLOATS / 4; i += 256) lb[i] = gb[i];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int cslot = 0;
  u32x4 af[4];
#pragma unroll
  for (int i = 0; i < 3; ++i) af[i] = *reinterpret_cast<const u32x4*>(ring.slot(0) + lane * 16 + i * 1024);
  f32x16 acc0[PT], acc1[PT];                                 // the two accumulator sets (VGPRs)
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) acc0[pt] = load_bias(lds_bias, 0, h);
  const int n_used = ring.n_used;
  // training forward: per-wave staging tiles (one per point tile) of the row-coalesced activation stores (sn_mlp_pipe.h)
  char* const xp = smem + MLP_BF16_LDS_BYTES + wave * (PT * XPOSE_WAVE_BYTES);
  const unsigned xp_w = (unsigned)(j * XPOSE_PITCH + 4 * h) * 4u;
  const unsigned xp_r = (unsigned)((lane >> 3) * XPOSE_PITCH + 4 * (lane & 7)) * 4u;
  const unsigned g_off = (unsigned)((lane >> 3) * 256 + 4 * (lane & 7)) * 4u;
  const unsigned xp16_w = (unsigned)(j * XS16_PITCH + 8 * h);                       // bf16 state: this lane's packed pairs
  const unsigned xp16_lds = (unsigned)(MLP_BF16_LDS_BYTES + wave * (PT * XPOSE_WAVE_BYTES)) + xp16_w;   // ... as an LDS byte address
  const unsigned xp16_r = (unsigned)((lane >> 3) * XS16_PITCH + 16 * (lane & 7));    // row lane>>3, 16-byte chunk lane&7 of a tile PAIR
  const unsigned g16_off = (unsigned)((lane >> 3) * 512 + 16 * (lane & 7));

  for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long p_wave = (tile * 4 + __builtin_amdgcn_readfirstlane(wave)) * (PT * 32);    // wave-uniform
    int ht = h;
    asm volatile("" : "+v"(ht));               // per-tile opaque copy of the lane half: keeps the embedding's frequency scales and
                                                 // column selects from being hoisted out of the tile loop (16-32 VGPRs the training
                                                 // variant does not have)
    long p_raw[PT], p[PT];
    bool valid[PT];
    u32x4 xe[4 * PT];                                        // [k-step][PT]
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      p_raw[pt] = ((tile * 4 + wave) * PT + pt) * 32 + j;
      valid[pt] = p_raw[pt] < P;
      p[pt] = valid[pt] ? p_raw[pt] : P - 1;
      float f[32];
      if (INPUT_MODE == 0) {
        const float* rp = in0 + (p[pt] / S) * 8;
        const float zz = in1[p[pt]];
        const float x = __fadd_rn(rp[0], __fmul_rn(rp[3], zz));
        const float y = __fadd_rn(rp[1], __fmul_rn(rp[4], zz));
        const float z = __fadd_rn(rp[2], __fmul_rn(rp[5], zz));
        embed_xyz_bf16(x, y, z, ht, f);
      } else {
        const float* row = in0 + p[pt] * (long)S;
        int hh = h;
        asm volatile("" : "+v"(hh));             // keep the 32 column selects inside the tile loop (else hoisted: +32 VGPRs)
#pragma unroll
        for (int e = 0; e < 32; ++e) {
          const int c0 = snl::xyz_slot_col(0, e), c1 = snl::xyz_slot_col(1, e);
          const int c = hh ? c1 : c0;
          f[e] = (c >= 0) ? row[c < 0 ? 0 : c] : 0.0f;
        }
      }
      if (STORE && INPUT_MODE == 0) {            // rows are allocated for whole 256-point tiles: no predicate.  (Pre-embedded
                                                 // rows, INPUT_MODE 1: the caller builds emb itself, it is a column re-layout of x)
        store_emb_xyz(emb + p_raw[pt] * 128, f, ht);   // columns [0, 63); the pad columns 63, 91..127 are never read back
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        xe[ks * PT + pt] = pack8(f + 8 * ks);
        asm volatile("" : "+v"(xe[ks * PT + pt]));           // convert here (32 fp32 temporaries per point tile die)
      }
      __builtin_amdgcn_sched_barrier(0);                     // ... and do not interleave the two point tiles
    }

    int s = 0;
    float sg[PT];                                            // sigma head partial (fp32, this lane half)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) sg[pt] = 0.0f;

    // training forward: four fp32 values (accumulator registers 4qq..4qq+3 of point tile pt) go to the wave's staging
    // tile; store_tile() then writes the staged 32-point x 32-feature tiles of both point tiles to
    // acts[slot][point][32t..32t+31] as whole 128-byte rows, non-temporal.
    int cur_slot = 0;
    uint32_t sign_bits = 0;                                  // STORE 2: ReLU sign word of the tile being finalised
    auto stage = [&](int pt, int qq, const float (&v)[4]) __attribute__((always_inline)) {
      if (STORE == 1) {
        f32x4 o;
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
        *reinterpret_cast<f32x4*>(xp + pt * XPOSE_WAVE_BYTES + xp_w + 32 * qq) = o;
      }
    };
    auto stage16 = [&](int pt, int t, int qq, uint32_t t0, uint32_t t1) __attribute__((always_inline)) {
      if (STORE == 2) {                          // tile t goes to half t & 1 of the 128-byte staged rows
        lds_write_b64(xp16_lds + pt * XPOSE_WAVE_BYTES, 64 * (t & 1) + 16 * qq, t0, t1);    // (ds_write2 offsets reach 1020 B)
      }
    };
    // memory operation k of finished tile t: k = 0..7 row group (pt = k >> 2, i = k & 3), k = 8 the tile's ReLU sign word
    constexpr int N_MEM_OPS = 9;
    auto mem_op = [&](int slot, int t, int k) __attribute__((always_inline)) {
      const int pt = k >> 2, i = k & 3;
      if (STORE == 1 && k < 8) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(xp + pt * XPOSE_WAVE_BYTES + xp_r + 8 * i * XPOSE_PITCH * 4);
        char* base = reinterpret_cast<char*>(acts) + (((long)slot * slot_rows + p_wave + pt * 32 + 8 * i) * 256 + 32 * t) * 4;
        unsigned go = g_off;
        asm volatile("" : "+v"(go));             // opaque per store: no hoisted per-slot address registers
        __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(base + go));
      } else if (STORE == 2 && k < 8) {
        if (t & 1) {                             // tiles t-1, t: whole 128-byte rows, 8 rows per instruction
          const f32x4 o = *reinterpret_cast<const f32x4*>(xp + pt * XPOSE_WAVE_BYTES + xp16_r + 8 * i * XS16_PITCH);
          char* base = reinterpret_cast<char*>(acts) + (((long)slot * slot_rows + p_wave + pt * 32 + 8 * i) * 256 + 32 * (t - 1)) * 2;
          unsigned go = g16_off;
          asm volatile("" : "+v"(go));
          __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(base + go));
        }
      } else if (STORE == 2 && k == 8) {
        if (slot < 8) {                          // ReLU layers: the tile's sign word, 256 contiguous bytes per wave
          char* base = reinterpret_cast<char*>(acts) + (((long)9 * slot_rows + p_wave + 8 * slot + t) * 256 + 128) * 2;
          unsigned go = (unsigned)lane * 4u;
          asm volatile("" : "+v"(go));
          __builtin_nontemporal_store(sign_bits, reinterpret_cast<uint32_t*>(base + go));
        }
      }
    };
    // the operations of memory step `step` of `n` (sn_mlp_bf16.h: dealt evenly over the k-steps behind the DMA pieces)
    auto mem_step = [&](int slot, int t, int step, int n) __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < N_MEM_OPS; ++k)
        if (k >= step * N_MEM_OPS / n && k < (step + 1) * N_MEM_OPS / n) mem_op(slot, t, k);
    };
    auto store_tile = [&](int slot, int t) __attribute__((always_inline)) { mem_step(slot, t, 0, 1); };
    // Epilogues of output tile t (results r) writing activation set W: dword q of the tile = accumulator registers
    // 2q, 2q+1 -> k-steps 2t, 2t+1 of the next layer (dwords q, q+1 for even q are adjacent registers).
    auto relu_tile = [&](auto wset, int t, const f32x16 (&r)[PT]) __attribute__((always_inline)) {
      constexpr int W = decltype(wset)::value;
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
          const int reg = act_reg(W, 2 * t + (q >> 2), pt) + (q & 3);
          if (STORE == 1) {
            float v[4];
            epi_relu_f32(reg, r[pt][2 * q], r[pt][2 * q + 1], r[pt][2 * q + 2], r[pt][2 * q + 3], v);
            stage(pt, q >> 1, v);
          } else if (STORE == 2) {
            uint32_t t0, t1;
            if (pt == 0 && q == 0) sign_bits = 0;
            epi_relu_bits(reg, r[pt][2 * q], r[pt][2 * q + 1], r[pt][2 * q + 2], r[pt][2 * q + 3], t0, t1, sign_bits);
            stage16(pt, t, q >> 1, t0, t1);
          } else {
            uint32_t t0, t1;
            epi_relu(reg, r[pt][2 * q], r[pt][2 * q + 1], r[pt][2 * q + 2], r[pt][2 * q + 3], t0, t1);
          }
        }
    };
    // layer 8: fp32 ReLU first, its output also feeds the sigma head (nerf.py:136)
    auto relu_sigma_tile = [&](auto wset, int t, const f32x16 (&r)[PT]) __attribute__((always_inline)) {
      constexpr int W = decltype(wset)::value;
      const f32x4* ws = reinterpret_cast<const f32x4*>(lds_aux + snl::AUX_SIGW + h * 128 + 16 * t);
      if (STORE == 2) sign_bits = 0;            // (block order here: q outermost -> steps 2q + 2pt, 2q + 2pt + 1 of the sign word)
#pragma unroll
      for (int q = 0; q < 8; q += 2) {
        const f32x4 w = ws[q >> 1];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
          float v[4];
          uint32_t t0, t1;
          if (STORE == 2)
            epi_relu_f32_bits(act_reg(W, 2 * t + (q >> 2), pt) + (q & 3), r[pt][2 * q], r[pt][2 * q + 1], r[pt][2 * q + 2], r[pt][2 * q + 3], v, t0, t1, sign_bits);
          else
            epi_relu_f32(act_reg(W, 2 * t + (q >> 2), pt) + (q & 3), r[pt][2 * q], r[pt][2 * q + 1], r[pt][2 * q + 2], r[pt][2 * q + 3], v, t0, t1);
          stage16(pt, t, q >> 1, t0, t1);
          sg[pt] = __builtin_fmaf(w[0], v[0], sg[pt]);
          sg[pt] = __builtin_fmaf(w[1], v[1], sg[pt]);
          sg[pt] = __builtin_fmaf(w[2], v[2], sg[pt]);
          sg[pt] = __builtin_fmaf(w[3], v[3], sg[pt]);
          stage(pt, q >> 1, v);
        }
      }
    };
    auto copy_tile = [&](auto wset, int t, const f32x16 (&r)[PT]) __attribute__((always_inline)) {     // xyz_encoding_final
      constexpr int W = decltype(wset)::value;
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
          uint32_t t0, t1;
          epi_copy(act_reg(W, 2 * t + (q >> 2), pt) + (q & 3), r[pt][2 * q], r[pt][2 * q + 1], r[pt][2 * q + 2], r[pt][2 * q + 3], t0, t1);
          const float v[4] = {r[pt][2 * q], r[pt][2 * q + 1], r[pt][2 * q + 2], r[pt][2 * q + 3]};
          stage(pt, q >> 1, v);
          stage16(pt, t, q >> 1, t0, t1);
        }
    };
#define SNB_LW_CUR (ring.slot(cslot) + lane * 16)
#define SNB_LW_NEXT (ring.slot(cslot == 2 ? 0 : cslot + 1) + lane * 16)
#define SNB_SNEXT (s + 1 == n_used ? 0 : s + 1)
#define SNB_ADVANCE() do { ++s; cslot = (cslot == 2) ? 0 : cslot + 1; } while (0)
#define SNB_W(W_) std::integral_constant<int, W_>{}
    // slab of output tile T_ (compile-time: selects the accumulator set); EPI_ = the previous tile's epilogue into set W_
#define SNB_SLAB(T_, NK0_, NK1_, S0_, S1_, GB_, PH_, NB_, BV_, EPI_, W_)                                           \
  do {                                                                                                             \
    /* bf16 state: the eight row stores of an odd finished tile sit behind the DMA pieces of their slab; the next   \
       slab's sync point leaves them in flight (tiles 3, 5, 7, and tile 0 behind the previous layer's last tile) */ \
    constexpr int VW_ = (STORE == 2 && SN_BF16_COUNTED && (NK0_) + (NK1_) >= 8 && ((T_) == 0 || (((T_) & 1) && (T_) >= 3))) ? 8 : 0; \
    if (((T_) & 1) == 0)                                                                                           \
      slab_bf16<NK0_, NK1_, S0_, S1_, GB_, PH_, NB_, VW_>(acc0, acc1, af, SNB_LW_CUR, BV_, SNB_LW_NEXT, lds_bias,       \
                                                     SNB_SNEXT, h, ring,                                           \
                                                     [&]() __attribute__((always_inline)) { if ((T_) > 0) EPI_(SNB_W(W_), (T_) - 1, acc1); }, \
                                                     [&](int st, int n) __attribute__((always_inline)) { if ((T_) > 0) mem_step(cur_slot, (T_) - 1, st, n); }); \
    else                                                                                                           \
      slab_bf16<NK0_, NK1_, S0_, S1_, GB_, PH_, NB_, VW_>(acc1, acc0, af, SNB_LW_CUR, BV_, SNB_LW_NEXT, lds_bias,       \
                                                     SNB_SNEXT, h, ring, [&]() __attribute__((always_inline)) { EPI_(SNB_W(W_), (T_) - 1, acc0); }, \
                                                     [&](int st, int n) __attribute__((always_inline)) { mem_step(cur_slot, (T_) - 1, st, n); }); \
    SNB_ADVANCE();                                                                                                 \
  } while (0)
    // the 8 output tiles of a layer, T_ literal (it ends up in asm immediates); tiles 6,7 stage the NEXT layer's slabs
#define SNB_LAYER(NK0_, NK1_, S0_, S1_, GB_, NBA_, NBB_, BV_, EPI_, W_)   \
  do {                                                                    \
    SNB_SLAB(0, NK0_, NK1_, S0_, S1_, GB_, 0, NBA_, BV_, EPI_, W_);       \
    SNB_SLAB(1, NK0_, NK1_, S0_, S1_, GB_, 0, NBA_, BV_, EPI_, W_);       \
    SNB_SLAB(2, NK0_, NK1_, S0_, S1_, GB_, 0, NBA_, BV_, EPI_, W_);       \
    SNB_SLAB(3, NK0_, NK1_, S0_, S1_, GB_, 0, NBA_, BV_, EPI_, W_);       \
    SNB_SLAB(4, NK0_, NK1_, S0_, S1_, GB_, 0, NBA_, BV_, EPI_, W_);       \
    SNB_SLAB(5, NK0_, NK1_, S0_, S1_, GB_, 0, NBA_, BV_, EPI_, W_);       \
    SNB_SLAB(6, NK0_, NK1_, S0_, S1_, GB_, 0, NBB_, BV_, EPI_, W_);       \
    SNB_SLAB(7, NK0_, NK1_, S0_, S1_, GB_, 0, NBB_, BV_, EPI_, W_);       \
    mfma_result_fence();                                                  \
    EPI_(SNB_W(W_), 7, acc1);                                             \
    store_tile(cur_slot, 7);                                              \
  } while (0)
    // bytes of the slab kinds (K * 64): the NB_ argument is the slab TWO ahead in the stream
    constexpr int B_L0 = 64 * 64, B_H = 256 * 64, B_SKIP = 320 * 64, B_DIR = 288 * 64;

    // ---- layer 0: reads the xyz embedding (VGPRs), writes set 0
    cur_slot = 0;
    SNB_LAYER(4, 0, -1, -1, 1, B_L0, B_H, xe, relu_tile, 0);

    // ---- layers 1..7: odd layers read set 0 and write set 1, even layers the reverse; skip concat at layer 4;
    //      layer 7's epilogues also feed the sigma head
#pragma unroll 1
    for (int l = 1; l < 8; ++l) {
      cur_slot = l;
      if (l == 4) {
        SNB_LAYER(4, 16, -1, 1, 2, B_SKIP, B_H, xe, relu_tile, 0);
      } else if (l == 7) {
        if (SIGMA_ONLY) SNB_LAYER(16, 0, 0, 0, 2, B_H, B_L0, xe, relu_sigma_tile, 1);     // next point tile's layer 0
        else SNB_LAYER(16, 0, 0, 0, 2, B_H, B_H, xe, relu_sigma_tile, 1);
      } else if (l == 3) {
        SNB_LAYER(16, 0, 0, 0, 2, B_H, B_SKIP, xe, relu_tile, 1);
      } else if (l & 1) {
        SNB_LAYER(16, 0, 0, 0, 2, B_H, B_H, xe, relu_tile, 1);
      } else {
        SNB_LAYER(16, 0, 1, 1, 2, B_H, B_H, xe, relu_tile, 0);
      }
    }

    float sigma[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) sigma[pt] = sg[pt] + __shfl_xor(sg[pt], 32, 64) + lds_aux[snl::AUX_HEADB];
    if (SIGMA_ONLY) {
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
        if (valid[pt] && h == 0) out[p_raw[pt]] = sigma[pt];
      continue;
    }

    // ---- xyz_encoding_final (no activation): reads set 1, writes set 0
    cur_slot = 8;
    SNB_LAYER(16, 0, 1, 1, 2, B_H, B_DIR, xe, copy_tile, 0);

    // ---- dir_encoding + ShiftedSoftplus: reads set 0 and the dir embedding (VGPRs); the rgb head is accumulated from
    //      the fp32 softplus outputs
    u32x4 de[2 * PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      float f[16];
      if (INPUT_MODE == 0) {
        const float* rp = in0 + (p[pt] / S) * 8;
        embed_dir_bf16(rp[3], rp[4], rp[5], ht, f);
      } else {
        const float* row = in0 + p[pt] * (long)S;
        int hh = h;
        asm volatile("" : "+v"(hh));
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int c0 = snl::dir_slot_col(0, e), c1 = snl::dir_slot_col(1, e);
          const int c = hh ? c1 : c0;
          f[e] = (c >= 0) ? row[63 + (c < 0 ? 0 : c)] : 0.0f;
        }
      }
      if (STORE && INPUT_MODE == 0) store_emb_dir(emb + p_raw[pt] * 128 + 64, f, ht);      // columns [64, 91)
      de[0 * PT + pt] = pack8(f);
      de[1 * PT + pt] = pack8(f + 8);
      asm volatile("" : "+v"(de[0 * PT + pt]), "+v"(de[1 * PT + pt]));
      __builtin_amdgcn_sched_barrier(0);
    }
    f32x2 c3[PT][3];                                            // rgb head partial sums, (even, odd) pairs: ssp4_rgb
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) c3[pt][0] = c3[pt][1] = c3[pt][2] = f32x2{0.0f, 0.0f};
    // ShiftedSoftplus + rgb head contribution: ssp4_rgb (sn_mlp_bf16.h), four values at a time (register pressure).
    auto ssp_tile = [&](auto, int t, const f32x16 (&r)[PT]) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 w[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
          w[c] = *reinterpret_cast<const f32x4*>(lds_aux + snl::AUX_RGBW + c * 128 + h * 64 + 16 * t + 4 * q);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
          float x[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) x[i] = r[pt][4 * q + i];
          // the chunk's inputs and the running sums pass through one volatile asm: chunks execute strictly in order
          asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(c3[pt][0]), "+v"(c3[pt][1]), "+v"(c3[pt][2]));
          float v[4];
          ssp4_rgb(x, w, c3[pt], v);
          stage(pt, q, v);
          if (STORE == 2) stage16(pt, t, q, pack2(v[0], v[1]), pack2(v[2], v[3]));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // 18 k-steps per slab: the fragment-ring phase alternates 0,2,0,2 (static); tiles 2,3 stage the next point tile
    cur_slot = 9;
    SNB_SLAB(0, 16, 2, 0, -1, 2, 0, B_DIR, de, ssp_tile, 0);
    SNB_SLAB(1, 16, 2, 0, -1, 2, 2, B_DIR, de, ssp_tile, 0);
    SNB_SLAB(2, 16, 2, 0, -1, 2, 0, B_L0, de, ssp_tile, 0);
    SNB_SLAB(3, 16, 2, 0, -1, 2, 2, B_L0, de, ssp_tile, 0);
    mfma_result_fence();
    ssp_tile(SNB_W(0), 3, acc1);
    store_tile(9, 3);
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      float o3[3];
#pragma unroll
      for (int c = 0; c < 3; ++c)
        o3[c] = rgb_activation(hsum(c3[pt][c]) + __shfl_xor(hsum(c3[pt][c]), 32, 64) + lds_aux[snl::AUX_HEADB + 1 + c]);
      if (valid[pt] && h == 0) {
        float4 o;
        o.x = o3[0]; o.y = o3[1]; o.z = o3[2]; o.w = sigma[pt];
        reinterpret_cast<float4*>(out)[p_raw[pt]] = o;
      }
    }
#undef SNB_LW_CUR
#undef SNB_LW_NEXT
#undef SNB_SNEXT
#undef SNB_ADVANCE
#undef SNB_SLAB
#undef SNB_LAYER
#undef SNB_W
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace snk

extern "C" int SN_LAUNCH_NAME(sn_mlp_forward_bf16)(const void* blob, const float* in0, const float* in1, long n_points, int s_or_ld,
                                          int sigma_only, int input_mode, float* out, float* acts, float* emb,
                                          long slot_rows, int state_bf16, hipStream_t stream) {
  using namespace snk;
  if (n_points <= 0) return 0;
  const long tiles = (n_points + 255) / 256;
  const bool store = acts != nullptr;
  if (store && (sigma_only || emb == nullptr || slot_rows < tiles * 256)) return -1;
  const int n_cu = snh::cu_count();
  dim3 grid((unsigned)(tiles < n_cu ? tiles : n_cu)), block(256);
  const size_t lds = MLP_BF16_LDS_BYTES + (store ? BF16_XPOSE_LDS_BYTES : 0);
  const char* b = reinterpret_cast<const char*>(blob);
#define SN_LAUNCH(SO, IM, ST)                                                                                    \
  do {                                                                                                           \
    auto kfn = mlp_fwd_bf16_kernel<SO, IM, ST>;                                                                  \
    SN_ENSURE_DYN_LDS(kfn, lds);                                                                                 \
    hipLaunchKernelGGL(kfn, grid, block, lds, stream, b, in0, in1, n_points, s_or_ld, out, acts, emb, slot_rows); \
  } while (0)
  if (store) {
    if (input_mode == 0) { if (state_bf16) SN_LAUNCH(false, 0, 2); else SN_LAUNCH(false, 0, 1); }
    else { if (state_bf16) SN_LAUNCH(false, 1, 2); else return -4; }    // embedded rows + fp32 state: out of registers, not built
  }
#ifdef SN_CLASSIC_HEADS                         // the sigma-only kernels never reach the heads: sn_api.hip routes them to the main pass
  else if (sigma_only) return -4;
  else if (input_mode == 0) SN_LAUNCH(false, 0, 0);
  else SN_LAUNCH(false, 1, 0);
#else
  else if (input_mode == 0) { if (sigma_only) SN_LAUNCH(true, 0, 0); else SN_LAUNCH(false, 0, 0); }
  else { if (sigma_only) SN_LAUNCH(true, 1, 0); else SN_LAUNCH(false, 1, 0); }
#endif
#undef SN_LAUNCH
  return (int)hipGetLastError();
}
