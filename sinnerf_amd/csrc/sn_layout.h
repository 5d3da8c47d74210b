// sn_layout.h -- packed-weight ("MFMA fragment order") layout of one NeRF MLP, shared by the host
// packer (table builder below, plain C++) and the device kernels (compile-time constants).
//
// Reference network: models/nerf.py:66-103 (NeRF(D=8,W=256,63,27,skips=[4],use_new_activation=True)).
//
// The fused MLP kernel computes every layer "transposed":  D[out_feature, point] = W * X^T, with the
// weight matrix as the MFMA A operand (streamed through LDS) and the activations as the B operand
// held in registers.  A 32x32 MFMA leaves D in the accumulator layout
//      col (point)   = lane & 31
//      row (feature) = (r & 3) + 8*(r >> 2) + 4*(lane >> 5),   r = accumulator register 0..15
// and consumes B as  B[k][col = lane & 31]  with the k index split over (lane >> 5, register slot).
// Because the order of the k index inside a contraction is free, the accumulators of layer n are fed
// *unchanged* as the B operand of layer n+1; the host permutes the K order of layer n+1's weights to
// match ("K-slot order").  No activation ever leaves the register file.
//
// K-slot order of a 32-feature hidden tile t (features 32t..32t+31), lane-half h = lane>>5:
//      slot r (0..15)  <->  feature 32t + (r&3) + 8*(r>>2) + 4*h
// fp32 path  (v_mfma_f32_32x32x2_f32):  one k-step = one slot      (k = h)
// bf16 path  (v_mfma_f32_32x32x16_bf16): one k-step = 8 slots r = 8*(s&1)+i (k = 8h+i)
//
// Blob = sequence of "slabs" (one 32-row output tile x full K each, in execution order) followed by a
// bias area.  A slab holds the A fragments in the exact order the wave reads them:
//   fp32: [K/8 groups][64 lanes][4 k-steps]  float   (one ds_read_b128 feeds 4 MFMAs)
//   bf16: [K/16 k-steps][64 lanes][8]        bf16    (one ds_read_b128 feeds 1 MFMA per point tile)
// lane = 32*h + i  holds output row i of the tile.  Bias area: per slab 32 floats [h][r] (accumulator order).
#pragma once
#include <stdint.h>

namespace snl {

// DT_BF16X3 (sn_mlp_fwd_bf16x3.hip): every weight as a (hi, lo) pair of bf16 -- hi = RNE(w), lo = RNE(w - hi) -- for the 3-term split
// product  W.x ~= Wh.xh + Wl.xh + Wh.xl  on the bf16 MFMA at fp32-level accuracy.  Slabs are K x 128 B like the fp32 ones:
//   bf16x3: [K/16 k-steps][hi, lo][64 lanes][8] bf16    (two ds_read_b128 feed three MFMAs)
enum { DT_F32 = 0, DT_BF16 = 1, DT_BF16X3 = 3, DT_F16 = 4 };     // DT_F16 (round 6): the DT_BF16 layout with fp16-rounded weights
// "x3 state" -- the training state of the bf16x3 kernels (acts / G, slots 0..8; slot 9 and emb stay fp32): a row of 256 features is
// 1 KB like an fp32 row, but holds every value as its (hi, lo) bf16 pair: per 8 consecutive features 16 B of hi parts, then 16 B of
// lo parts.  The forward / chain epilogues have the pairs in registers anyway (they are the next layer's B operand); the
// weight-gradient kernel (sn_dw.hip, modes 5..7) stages the rows by DMA with the swizzle of RowStager<.., SPLIT> and takes MFMA
// fragments by ds_read_b64_tr_b16 -- no conversion anywhere.  tests/helpers.py x3_state_decode / _encode restate it in numpy.

// ---- network constants (padded K per layer kind)
constexpr int W_HID = 256;
constexpr int K_XYZ = 64;    // 63 embedded xyz features + 1 zero pad          (nerf.py:68)
constexpr int K_DIRE = 32;   // 27 embedded dir features + 5 zero pads        (nerf.py:82)
constexpr int K_L0 = K_XYZ;
constexpr int K_HID = W_HID;
constexpr int K_SKIP = K_XYZ + W_HID;    // cat([input_xyz, h])  nerf.py:133 (xyz FIRST)
constexpr int K_DIR = W_HID + K_DIRE;    // cat([final, input_dir]) nerf.py:142 (dir LAST)
constexpr int K_RGB = W_HID / 2;

// ---- slab sequence (execution order)
//  0.. 7  L0   (xyz_encoding_1)            K_L0
//  8..31  L1-3 (xyz_encoding_2..4)         K_HID
// 32..39  L4   (xyz_encoding_5, skip)      K_SKIP
// 40..63  L5-7 (xyz_encoding_6..8)         K_HID
// 64..71  FIN  (xyz_encoding_final)        K_HID
// 72..75  DIR  (dir_encoding)              K_DIR
// The two narrow heads -- sigma (1 row, nerf.py:86) and rgb (3 rows, nerf.py:89) -- are NOT padded to 32-row MFMA tiles:
// their weights live in the "aux" table behind the biases (K-slot order, per lane half) and are applied on the VALU
// straight from the register-resident activations.
constexpr int N_SLABS = 76;
constexpr int SLAB_FIN = 64, SLAB_DIR = 72;

constexpr int slab_k(int s) {
  return s < 8 ? K_L0 : s < 32 ? K_HID : s < 40 ? K_SKIP : s < 72 ? K_HID : K_DIR;
}
// elements (of the weight dtype) per slab = 32 rows x K
constexpr int slab_elems(int s) { return 32 * slab_k(s); }
constexpr long slab_elem_offset(int s) {
  long o = 0;
  for (int i = 0; i < s; ++i) o += slab_elems(i);
  return o;
}
constexpr long TOTAL_W_ELEMS = slab_elem_offset(N_SLABS);           // 593920
constexpr int esize(int dt) { return (dt == DT_BF16 || dt == DT_F16) ? 2 : 4; }        // bytes per weight in the blob (bf16x3: a 2 + 2 byte pair)
constexpr long bias_byte_offset(int dt) { return TOTAL_W_ELEMS * esize(dt); }
constexpr int BIAS_FLOATS = N_SLABS * 32;
// aux table (fp32) behind the biases:  sigma_w[2][128] | rgb_w[3][2][64] | sigma_b, rgb_b[3] | pad  -> 648 floats
constexpr int AUX_SIGW = 0, AUX_RGBW = 256, AUX_HEADB = 640, AUX_FLOATS = 648;
constexpr int TAIL_FLOATS = BIAS_FLOATS + AUX_FLOATS;                // 3080 floats = 12320 B (16 B multiple)
constexpr long blob_bytes(int dt) { return bias_byte_offset(dt) + (long)TAIL_FLOATS * 4; }
constexpr int MAX_SLAB_K = K_SKIP;

// ---- raw tensor ids (order of NeRF.state_dict(): nerf.py:66-103)
//  2*l, 2*l+1 : xyz_encoding_{l+1}.0.{weight,bias}  l=0..7
//  16,17 xyz_encoding_final ; 18,19 dir_encoding.0 ; 20,21 sigma ; 22,23 rgb.0
constexpr int N_RAW = 24;
constexpr int RAW_FIN = 16, RAW_DIR = 18, RAW_SIG = 20, RAW_RGB = 22;

// accumulator register r, lane half h -> row inside the 32-row tile
constexpr int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- embedded-input slot maps (what each lane half computes itself, see sn_mlp_fwd.hip)
// xyz: 32 slots per half.  e<30: pair p=e>>1 -> (band = 5h + p/3, coord = p%3), e&1: 0 sin, 1 cos
//      e=30: h0 -> x, h1 -> z ; e=31: h0 -> y, h1 -> pad.  Reference column: nerf.py:36-41
//      [x y z | sin(f0 xyz) cos(f0 xyz) | sin(f1 xyz) ...] -> 3 + 6*band + 3*fn + coord
constexpr int xyz_slot_col(int h, int e) {
  if (e < 30) { int p = e >> 1; return 3 + 6 * (5 * h + p / 3) + 3 * (e & 1) + p % 3; }
  if (e == 30) return h == 0 ? 0 : 2;
  return h == 0 ? 1 : -1;
}
// dir: 16 slots per half. e<12: pair p=e>>1 -> (band = 2h + p/3, coord = p%3); e=12,13: h0 -> dx,dy ; h1 -> dz,pad
constexpr int dir_slot_col(int h, int e) {
  if (e < 12) { int p = e >> 1; return 3 + 6 * (2 * h + p / 3) + 3 * (e & 1) + p % 3; }
  if (e == 12) return h == 0 ? 0 : 2;
  if (e == 13) return h == 0 ? 1 : -1;
  return -1;
}

// K-slot (global slot index q within the layer's padded K, lane half h) -> raw weight column, or -1 (zero).
// A layer's K is the concatenation of segments in the order the kernel walks them.
constexpr int hid_slot_feature(int q, int h) { return 32 * (q >> 4) + acc_row(q & 15, h); }   // q in [0,128)

constexpr int slab_raw_col(int slab, int q, int h) {
  // q = slot index in [0, K/2)
  if (slab < 8) return xyz_slot_col(h, q);                                  // L0: K = xyz
  if (slab >= 32 && slab < 40) {                                            // skip: [xyz | hid]
    if (q < 32) return xyz_slot_col(h, q);
    return 63 + hid_slot_feature(q - 32, h);
  }
  if (slab >= SLAB_DIR) {                                                   // dir: [hid(final) | dir]
    if (q < 128) return hid_slot_feature(q, h);
    int c = dir_slot_col(h, q - 128);
    return c < 0 ? -1 : 256 + c;
  }
  return hid_slot_feature(q, h);                                            // hidden / fin
}
// slab -> raw weight tensor id, first raw row of the tile, number of valid rows
constexpr int slab_raw_w(int s) { return s < 64 ? 2 * (s / 8) : s < SLAB_DIR ? RAW_FIN : RAW_DIR; }
constexpr int slab_row0(int s) { return s < 64 ? 32 * (s % 8) : s < SLAB_DIR ? 32 * (s - SLAB_FIN) : 32 * (s - SLAB_DIR); }
constexpr int slab_rows(int) { return 32; }
constexpr int raw_cols(int raw_w) {
  return raw_w == 0 ? 63 : raw_w == 8 ? 319 : raw_w == RAW_DIR ? 283 : raw_w == RAW_RGB ? 128 : 256;
}

// ---- host-side table: one entry per blob element.
//  dst : byte offset into the blob
//  src : -1 -> zero (weight dtype) ; -2 -> zero (fp32) ; else
//        [SRC_F32_FLAG] | (raw_tensor_id << 20) | flat element offset in that raw fp32 tensor
//  Entries carrying SRC_F32_FLAG (biases) are stored as fp32 whatever the weight dtype.
struct PackEntry { int32_t dst; int32_t src; };
constexpr int32_t SRC_F32_FLAG = 1 << 30;
constexpr int32_t SRC_LO_FLAG = 1 << 29;         // DT_BF16X3: store RNE(w - RNE(w)) instead of RNE(w)
constexpr long table_entries() { return TOTAL_W_ELEMS + TAIL_FLOATS; }
constexpr long table_entries_dt(int dt) { return (dt == DT_BF16X3 ? 2 : 1) * TOTAL_W_ELEMS + TAIL_FLOATS; }

inline void build_pack_table(int dt, PackEntry* out) {
  long n = 0;
  const int es = esize(dt);
  for (int s = 0; s < N_SLABS; ++s) {
    const int K = slab_k(s), rw = slab_raw_w(s), row0 = slab_row0(s), rows = slab_rows(s), ncol = raw_cols(rw);
    const long base = slab_elem_offset(s);
    if (dt == DT_F32) {
      for (int g = 0; g < K / 8; ++g)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 4; ++j) {
            const int i = lane & 31, h = lane >> 5, q = 4 * g + j;
            const int col = slab_raw_col(s, q, h);
            PackEntry e;
            e.dst = (int32_t)((base + ((long)g * 64 + lane) * 4 + j) * es);
            e.src = (i < rows && col >= 0) ? ((rw << 20) | ((row0 + i) * ncol + col)) : -1;
            out[n++] = e;
          }
    } else if (dt == DT_BF16X3) {
      for (int ks = 0; ks < K / 16; ++ks)
        for (int part = 0; part < 2; ++part)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
              const int i = lane & 31, h = lane >> 5, q = 8 * ks + j;
              const int col = slab_raw_col(s, q, h);
              PackEntry e;
              e.dst = (int32_t)(base * 4 + ((((long)ks * 2 + part) * 64 + lane) * 8 + j) * 2);
              e.src = (i < rows && col >= 0) ? ((part ? SRC_LO_FLAG : 0) | (rw << 20) | ((row0 + i) * ncol + col)) : -1;
              out[n++] = e;
            }
    } else {
      for (int ks = 0; ks < K / 16; ++ks)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int i = lane & 31, h = lane >> 5, q = 8 * ks + j;
            const int col = slab_raw_col(s, q, h);
            PackEntry e;
            e.dst = (int32_t)((base + ((long)ks * 64 + lane) * 8 + j) * es);
            e.src = (i < rows && col >= 0) ? ((rw << 20) | ((row0 + i) * ncol + col)) : -1;
            out[n++] = e;
          }
    }
  }
  for (int s = 0; s < N_SLABS; ++s) {
    const int rb = slab_raw_w(s) + 1, row0 = slab_row0(s), rows = slab_rows(s);
    for (int h = 0; h < 2; ++h)
      for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, h);
        PackEntry e;
        e.dst = (int32_t)(bias_byte_offset(dt) + ((long)s * 32 + h * 16 + r) * 4);
        e.src = row < rows ? (SRC_F32_FLAG | (rb << 20) | (row0 + row)) : -2;
        out[n++] = e;
      }
  }
  // aux table: head weights in K-slot order (slot q of lane half h  <->  hid_slot_feature(q, h))
  const long aux0 = bias_byte_offset(dt) + (long)BIAS_FLOATS * 4;
  for (int a = 0; a < AUX_FLOATS; ++a) {
    PackEntry e;
    e.dst = (int32_t)(aux0 + (long)a * 4);
    e.src = -2;
    if (a < AUX_RGBW) {                                   // sigma_w[h][q], q < 128  (sigma.weight is 1 x 256)
      const int h = a / 128, q = a % 128;
      e.src = SRC_F32_FLAG | (RAW_SIG << 20) | hid_slot_feature(q, h);
    } else if (a < AUX_HEADB) {                           // rgb_w[c][h][q], q < 64   (rgb.0.weight is 3 x 128)
      const int c = (a - AUX_RGBW) / 128, h = ((a - AUX_RGBW) % 128) / 64, q = (a - AUX_RGBW) % 64;
      e.src = SRC_F32_FLAG | (RAW_RGB << 20) | (c * 128 + hid_slot_feature(q, h));
    } else if (a == AUX_HEADB) {
      e.src = SRC_F32_FLAG | ((RAW_SIG + 1) << 20) | 0;
    } else if (a < AUX_HEADB + 4) {
      e.src = SRC_F32_FLAG | ((RAW_RGB + 1) << 20) | (a - AUX_HEADB - 1);
    }
    out[n++] = e;
  }
}

// ================================================================================================
// Backward-chain blob: TRANSPOSED weights for  g_x[in_feature, point] = W^T * g_y   (same register-resident
// scheme as the forward: the masked accumulators of one transposed layer are the B operand of the next).
// Slabs in execution order (one 32-row tile of INPUT features x K of OUTPUT features in K-slot order):
//   0.. 7  DIRT  K=128  rows = final features (dir_encoding.0[:, :256]^T)   k-slots: g_y2 (128)
//   8..15  FINT  K=256  rows = h8 features (xyz_encoding_final^T)           k-slots: g_final (256)
//  16..71  LT    K=256  layers i = 7,6,5,4,3,2,1 (xyz_encoding_{i+1}^T), 8 tiles each, rows = hidden inputs of layer i
//                       (for the skip layer i=4 the hidden part = columns 63..318 of its weight, nerf.py:133)
// followed by a tail: 64 zeros (the slab pipeline's "bias" slot: these layers have no bias term) and the aux table of the
// two narrow transposed heads, applied on the VALU like in the forward:
//   rgbT[3][2][64] (rgb.0.weight[c][f], f = K-slot order of the 128 h2 features) | sigT[2][128] (sigma.weight[f]).
// fp32 (the training path of configs[1]); the bf16-operand chain has its own blob below.
constexpr int NB_SLABS = 72;
constexpr int BSLAB_FINT = 8, BSLAB_LT = 16;
constexpr int bslab_k(int s) { return s < 8 ? 128 : 256; }
constexpr long bslab_elem_offset(int s) { return s < 8 ? (long)s * 32 * 128 : 8L * 32 * 128 + (long)(s - 8) * 32 * 256; }
constexpr long B_TOTAL_ELEMS = bslab_elem_offset(NB_SLABS);
constexpr int B_ZERO_FLOATS = 64;
constexpr int B_AUX_RGBT = 0, B_AUX_SIGT = 384, B_AUX_FLOATS = 640;
constexpr int B_TAIL_FLOATS = B_ZERO_FLOATS + B_AUX_FLOATS;                      // 704 floats = 2816 B
constexpr long b_tail_byte_offset() { return B_TOTAL_ELEMS * 4; }
constexpr long bblob_bytes() { return b_tail_byte_offset() + (long)B_TAIL_FLOATS * 4; }
constexpr long b_table_entries() { return B_TOTAL_ELEMS + B_TAIL_FLOATS; }
constexpr int B_MAX_SLAB_K = 256;

inline void build_pack_table_bwd(PackEntry* out) {
  long n = 0;
  for (int s = 0; s < NB_SLABS; ++s) {
    const int K = bslab_k(s);
    const long base = bslab_elem_offset(s);
    for (int g = 0; g < K / 8; ++g)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 4; ++j) {
          const int i = lane & 31, h = lane >> 5, q = 4 * g + j;
          int32_t src;
          if (s < BSLAB_FINT) {                               // dir_encoding.0^T : W_d (128 x 283), cols 0..255
            src = (RAW_DIR << 20) | (hid_slot_feature(q, h) * 283 + 32 * s + i);
          } else if (s < BSLAB_LT) {                          // xyz_encoding_final^T
            src = (RAW_FIN << 20) | (hid_slot_feature(q, h) * 256 + 32 * (s - BSLAB_FINT) + i);
          } else {                                            // xyz_encoding_{li+1}^T, li = 7..1
            const int li = 7 - (s - BSLAB_LT) / 8, t = (s - BSLAB_LT) % 8;
            const int ncol = raw_cols(2 * li), coloff = (li == 4) ? 63 : 0;
            src = ((2 * li) << 20) | (hid_slot_feature(q, h) * ncol + coloff + 32 * t + i);
          }
          PackEntry e;
          e.dst = (int32_t)((base + ((long)g * 64 + lane) * 4 + j) * 4);
          e.src = src;
          out[n++] = e;
        }
  }
  const long tail = b_tail_byte_offset();
  for (int a = 0; a < B_TAIL_FLOATS; ++a) {
    PackEntry e;
    e.dst = (int32_t)(tail + (long)a * 4);
    e.src = -2;
    const int x = a - B_ZERO_FLOATS;
    if (x >= B_AUX_RGBT && x < B_AUX_SIGT) {                  // rgbT[c][h][q], q < 64
      const int c = x / 128, h = (x % 128) / 64, q = x % 64;
      e.src = SRC_F32_FLAG | (RAW_RGB << 20) | (c * 128 + hid_slot_feature(q, h));
    } else if (x >= B_AUX_SIGT) {                             // sigT[h][q], q < 128
      const int h = (x - B_AUX_SIGT) / 128, q = (x - B_AUX_SIGT) % 128;
      e.src = SRC_F32_FLAG | (RAW_SIG << 20) | hid_slot_feature(q, h);
    }
    out[n++] = e;
  }
}


// ================================================================================================
// Backward-chain blob, bf16 operands (mixed-precision training: sn_mlp_bwd_bf16.hip).  Same transposed scheme, 32x32x16
// fragments (8 K-slots per lane per k-step), and the two narrow transposed heads moved to the VALU like in the forward:
//   0.. 7  DIRT  K=128  rows = final features (dir_encoding.0[:, :256]^T)   k-slots: g_y2 (128)
//   8..15  FINT  K=256  rows = h8 features (xyz_encoding_final^T)           k-slots: g_final (256)
//  16..71  LT    K=256  layers i = 7..1 (xyz_encoding_{i+1}^T), 8 tiles each
// followed by an fp32 tail: 72 x 32 zeros (the slab pipeline's bias slots: these layers have no bias term) and the aux
// table  rgbT[3][2][64] (rgb.0.weight[c][f], f = K-slot order of the 128 h2 features) | sigT[2][128] (sigma.weight[f]).
constexpr int NBB_SLABS = 72;
constexpr int BBSLAB_FINT = 8, BBSLAB_LT = 16;
constexpr int bbslab_k(int s) { return s < 8 ? 128 : 256; }
constexpr long bbslab_elem_offset(int s) { return s < 8 ? (long)s * 32 * 128 : 8L * 32 * 128 + (long)(s - 8) * 32 * 256; }
constexpr long BB_TOTAL_ELEMS = bbslab_elem_offset(NBB_SLABS);
constexpr int BB_ZERO_FLOATS = NBB_SLABS * 32;
constexpr int BB_AUX_RGBT = 0, BB_AUX_SIGT = 384, BB_AUX_FLOATS = 640;
constexpr int BB_TAIL_FLOATS = BB_ZERO_FLOATS + BB_AUX_FLOATS;                   // 2944 floats = 11776 B
constexpr long bb_tail_byte_offset() { return BB_TOTAL_ELEMS * 2; }
constexpr long bbblob_bytes() { return bb_tail_byte_offset() + (long)BB_TAIL_FLOATS * 4; }
constexpr long bb_table_entries() { return BB_TOTAL_ELEMS + BB_TAIL_FLOATS; }

inline void build_pack_table_bwd_bf16(PackEntry* out) {
  long n = 0;
  for (int s = 0; s < NBB_SLABS; ++s) {
    const int K = bbslab_k(s);
    const long base = bbslab_elem_offset(s);
    for (int ks = 0; ks < K / 16; ++ks)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int i = lane & 31, h = lane >> 5, q = 8 * ks + j;
          int32_t src;
          if (s < BBSLAB_FINT) {                              // dir_encoding.0^T : W_d (128 x 283), cols 0..255
            src = (RAW_DIR << 20) | (hid_slot_feature(q, h) * 283 + 32 * s + i);
          } else if (s < BBSLAB_LT) {                         // xyz_encoding_final^T
            src = (RAW_FIN << 20) | (hid_slot_feature(q, h) * 256 + 32 * (s - BBSLAB_FINT) + i);
          } else {                                            // xyz_encoding_{li+1}^T, li = 7..1
            const int li = 7 - (s - BBSLAB_LT) / 8, t = (s - BBSLAB_LT) % 8;
            const int ncol = raw_cols(2 * li), coloff = (li == 4) ? 63 : 0;
            src = ((2 * li) << 20) | (hid_slot_feature(q, h) * ncol + coloff + 32 * t + i);
          }
          PackEntry e;
          e.dst = (int32_t)((base + ((long)ks * 64 + lane) * 8 + j) * 2);
          e.src = src;
          out[n++] = e;
        }
  }
  const long tail = bb_tail_byte_offset();
  for (int a = 0; a < BB_TAIL_FLOATS; ++a) {
    PackEntry e;
    e.dst = (int32_t)(tail + (long)a * 4);
    e.src = -2;
    const int x = a - BB_ZERO_FLOATS;
    if (x >= BB_AUX_RGBT && x < BB_AUX_SIGT) {                // rgbT[c][h][q], q < 64
      const int c = x / 128, h = (x % 128) / 64, q = x % 64;
      e.src = SRC_F32_FLAG | (RAW_RGB << 20) | (c * 128 + hid_slot_feature(q, h));
    } else if (x >= BB_AUX_SIGT) {                            // sigT[h][q], q < 128
      const int h = (x - BB_AUX_SIGT) / 128, q = (x - BB_AUX_SIGT) % 128;
      e.src = SRC_F32_FLAG | (RAW_SIG << 20) | hid_slot_feature(q, h);
    }
    out[n++] = e;
  }
}


// ================================================================================================
// Backward-chain blob, bf16x3 (sn_mlp_bwd_bf16x3.hip): the slabs of the bf16 chain blob above with every weight as a (hi, lo)
// bf16 pair -- per k-step the hi fragment then the lo fragment (DT_BF16X3) -- and the same fp32 tail (zero bias slots, rgbT, sigT).
constexpr long bbx_tail_byte_offset() { return BB_TOTAL_ELEMS * 4; }
constexpr long bbxblob_bytes() { return bbx_tail_byte_offset() + (long)BB_TAIL_FLOATS * 4; }
constexpr long bbx_table_entries() { return 2 * BB_TOTAL_ELEMS + BB_TAIL_FLOATS; }

inline void build_pack_table_bwd_bf16x3(PackEntry* out) {
  // the weight entries of the bf16 table, doubled; its tail entries, moved
  PackEntry* b16 = new PackEntry[bb_table_entries()];
  build_pack_table_bwd_bf16(b16);
  long n = 0;
  for (long e = 0; e < BB_TOTAL_ELEMS; ++e) {
    // element e of the bf16 blob sits at byte 2e = ((frag * 64 + lane) * 8 + j) * 2 with frag = 1 KB fragment index over the
    // whole stream (slabs are whole numbers of fragments): its pair goes to fragments 2 frag (hi) and 2 frag + 1 (lo)
    const long byte16 = b16[e].dst;
    const long frag = byte16 / 1024, within = byte16 % 1024;
    for (int part = 0; part < 2; ++part) {
      PackEntry x;
      x.dst = (int32_t)((2 * frag + part) * 1024 + within);
      x.src = b16[e].src >= 0 ? (b16[e].src | (part ? SRC_LO_FLAG : 0)) : b16[e].src;
      out[n++] = x;
    }
  }
  for (long e = BB_TOTAL_ELEMS; e < bb_table_entries(); ++e) {
    PackEntry x = b16[e];
    x.dst = (int32_t)(x.dst - bb_tail_byte_offset() + bbx_tail_byte_offset());
    out[n++] = x;
  }
  delete[] b16;
}

}  // namespace snl
