// Fused SHMP layer for gfx950: SAGEConv gather-aggregate for every relation slot, the per-relation
// Linear, the to_hetero sum, the update Linear and the ReLU (gnn_model.py:262-264, 273, 389-395) in
// ONE launch; the aggregates never touch HBM.
//
//   out[i] = relu( sum_{s<sm} (sum_{e in vrow(i,s)} x[vcol[e]]) * Wt[s] + x[i] * Wt[sm] + bias
//                  + sum_{sm<=s<sm+st} sum_{e in vrow(i,s)} ytab[vcol[e]-ytab_row0][(s-sm)*64 : +64] )
//
// "MFMA slots" (s < sm) are gathered into LDS and multiplied on the matrix cores; "table slots"
// are the same linear map re-associated, (sum_j x_j) W = sum_j (x_j W): their sources were already
// multiplied by W (ytab), so they are a row add in the epilogue.  The host uses the table form for
// the canonical->count relations (at most one source per row), cutting K from 320 to 192.
//
// Structure (wave-autonomous, persistent, software-pipelined):
//   * block = 8 waves = one CU (2 waves per SIMD, <= 256 VGPRs each); all (sm+1) 64x64 weight blocks
//     are loaded into LDS ONCE per block and stay resident while the block strides over 256-row tiles;
//   * every wave owns 32 destination rows of a tile end to end and never meets a block barrier in
//     the tile loop;
//   * the CSR slice of the wave's NEXT tile is prefetched under the current tile's work: its row
//     pointers into a second private LDS buffer, its first 256 source ids into registers that are
//     written over the (then dead) id buffer at the tile switch; ids 256..WCAP of a dense tile are
//     fetched at the switch;
//   * gathers are batched and branch-free: lane group g8 (8 lanes x float4) serves rows it*8+g8,
//     it = 0..3, and one step fetches two sources x both 128-B halves for all four rows = 16 loads
//     in flight together; absent sources read a row of zeros instead of being predicated away (a
//     load under a divergent branch is fenced by its own s_waitcnt and serialises the round trips);
//   * the first gather step of the next LIVE K block is issued BEFORE the MFMAs of block b, and the
//     first live block of the next tile before the epilogue stores of the current one, so
//     feature-row latency hides under the matrix work of the same wave; a relation slot without a
//     source in the wave's 32 rows is dead: no gather, no split, no MFMAs (exact zeros);
//   * rows with more than 2 sources in a slot take up to EXTRA_STEPS more batched steps; beyond that
//     (hub rows, canonical rows of dense neighborhoods, ids past WCAP) they are finished
//     cooperatively by the whole wave (8 lane groups stride over one row's sources);
//   * table slots run as one extra pseudo K block: their pre-transformed source rows are gathered
//     the same way, staged in the A image and ADDED to the accumulators in the C/D layout;
//   * HBM traffic per row and layer: one 256-B read of x, one 256-B write, ~20 B of indices
//     (neighbour re-reads hit L2 / MALL: a neighborhood's rows are contiguous);
//   * optional fused pooling (global_add_pool of the produced rows): after the stores a wave swaps
//     the two lane halves of its accumulators (v_permlane32_swap: lane = column, all 32 rows of the
//     tile in registers in row order), runs one running sum down the rows and writes it out at every
//     segment end -- one 256-B partial per (tile, segment), summed per segment by pool_reduce_kernel.
//     All control flow of that pass is wave-uniform (the segment-end bitmap of the tile is a scalar).
//
// Two arithmetic modes share the gather:
//   f32   v_mfma_f32_32x32x2_f32 on an fp32 A image [32][33] and fp32 weights [K][64];
//   x6    the fp32-accurate 6-product bf16 split of gemm_split.hip: the gathered sums are split
//         into three bf16 planes [3][32][40] when they are written to LDS, the weights arrive
//         pre-split (n-major planes) and v_mfma_f32_32x32x16_bf16 does 24 MFMAs of 32 cycles per
//         32-deep half block instead of 32 MFMAs of 64 cycles (sm <= 2: the planes of four weight
//         blocks do not fit beside eight waves).
#include "common_device.hpp"
#include "shmp_args.hpp"

namespace desco {

constexpr int WR = 32;        // rows per wave
constexpr int NW = 8;         // waves per block (2 per SIMD)
constexpr int AH = 33;        // half-K fp32 A image row stride (floats): conflict-free ds_read_b32
constexpr int APS = 40;       // half-K bf16 plane row stride (shorts, 80 B): conflict-free ds_read_b128
constexpr int MAXS = 4;       // relation slots stored per row
constexpr int RPN = WR * MAXS + 1;
constexpr int EXTRA_STEPS = 9; // batched 2-source steps after the prefetched one (<= 20 sources per row)
constexpr int WCAP = 512;     // source ids staged per wave (longer slices fall back to global);
                              // one buffer: the next tile's first 256 ids wait in registers until
                              // the switch, a denser tile fetches the rest then
constexpr int A_FLOATS = 3 * WR * APS / 2;               // A region per wave: max(32*33, 3*32*40/2) floats
constexpr int WAVE_LDS = A_FLOATS + 2 * RPN + WCAP;      // floats per wave
static_assert(A_FLOATS >= WR * AH, "the fp32 image must fit in the plane region");

// absent sources of a batched gather step read this row instead of being predicated away
__device__ __attribute__((aligned(16))) float shmp_zero_row[64] = {};


using bf16x8 = __attribute__((ext_vector_type(8))) short;

__device__ __forceinline__ void f4add(float4& a, const float4 b) {
  a.x += b.x;
  a.y += b.y;
  a.z += b.z;
  a.w += b.w;
}

// ---- gather machinery (macros: every temporary is a named register) ----------------
// They use the enclosing scope's rp, ec, ebase, grow0, nr, xb, yb, zrow, g, S, g8, l8 and the
// registers lo*/hi* (sums), u*/w* (loads in flight), c*/n* (cursors).
#define DESCO_CUR(it_, slot_)                              \
  {                                                        \
    const int v_ = ((it_) * 8 + g8) * S + (slot_);         \
    c##it_ = rp[v_] - ebase;                               \
    n##it_ = rp[v_ + 1] - ebase;                           \
  }
#define DESCO_CURS(slot_) DESCO_CUR(0, slot_) DESCO_CUR(1, slot_) DESCO_CUR(2, slot_) DESCO_CUR(3, slot_)
// two sources of row it_ (staged ids only: e < WCAP), unconditional loads
#define DESCO_ISSUE2(it_, base_, ld_)                                                 \
  {                                                                                   \
    const int m_ = n##it_ < WCAP ? n##it_ : WCAP;                                     \
    const bool k0_ = c##it_ < m_, k1_ = c##it_ + 1 < m_;                              \
    const int i0_ = ec[k0_ ? c##it_ : 0], i1_ = ec[k1_ ? c##it_ + 1 : 0];             \
    const float* p0_ = k0_ ? (base_) + (int64_t)i0_ * (ld_) : zrow;                   \
    const float* p1_ = k1_ ? (base_) + (int64_t)i1_ * (ld_) : zrow;                   \
    u##it_##0 = *reinterpret_cast<const float4*>(p0_);                                \
    u##it_##1 = *reinterpret_cast<const float4*>(p0_ + 32);                           \
    w##it_##0 = *reinterpret_cast<const float4*>(p1_);                                \
    w##it_##1 = *reinterpret_cast<const float4*>(p1_ + 32);                           \
    c##it_ += (k0_ ? 1 : 0) + (k1_ ? 1 : 0);                                          \
  }
#define DESCO_CONSUME2(it_)                                                           \
  {                                                                                   \
    f4add(lo##it_, u##it_##0);                                                        \
    f4add(hi##it_, u##it_##1);                                                        \
    f4add(lo##it_, w##it_##0);                                                        \
    f4add(hi##it_, w##it_##1);                                                        \
  }
#define DESCO_ZERO_SUMS()                                   \
  {                                                         \
    lo0 = make_float4(0.f, 0.f, 0.f, 0.f);                  \
    lo1 = lo0; lo2 = lo0; lo3 = lo0;                        \
    hi0 = lo0; hi1 = lo0; hi2 = lo0; hi3 = lo0;             \
  }
#define DESCO_ANY_STAGED()                                                                    \
  __any((c0 < (n0 < WCAP ? n0 : WCAP)) | (c1 < (n1 < WCAP ? n1 : WCAP)) |                     \
        (c2 < (n2 < WCAP ? n2 : WCAP)) | (c3 < (n3 < WCAP ? n3 : WCAP)))
// the row itself: rows beyond nr re-read the wave's last valid row (never stored)
#define DESCO_ISSUE_SELF(it_)                                                                  \
  {                                                                                            \
    const int r_ = (it_) * 8 + g8;                                                             \
    const float* p_ = xb + (grow0 + (r_ < nr ? r_ : nr - 1)) * LDX;                          \
    u##it_##0 = *reinterpret_cast<const float4*>(p_);                                          \
    u##it_##1 = *reinterpret_cast<const float4*>(p_ + 32);                                     \
  }
// table pseudo block: the first source of table slot 0 (-> u) and of table slot 1 (-> w) of row it_
#define DESCO_TAB_CUR(it_)                                                                  \
  const int v_ = ((it_) * 8 + g8) * S + g.sm;                                               \
  const int ca_ = rp[v_] - ebase, na_ = rp[v_ + 1] - ebase;                                 \
  const int nb_ = ST > 1 ? rp[v_ + 2] - ebase : na_;                                        \
  const bool k0_ = ca_ < (na_ < WCAP ? na_ : WCAP);                                         \
  const bool k1_ = ST > 1 && na_ < (nb_ < WCAP ? nb_ : WCAP);
#define DESCO_ISSUE_TAB(it_)                                                                \
  {                                                                                         \
    DESCO_TAB_CUR(it_)                                                                      \
    const int i0_ = ec[k0_ ? ca_ : 0], i1_ = ec[k1_ ? na_ : 0];                             \
    const float* p0_ = k0_ ? yb + (int64_t)i0_ * LDY : zrow;                              \
    u##it_##0 = *reinterpret_cast<const float4*>(p0_);                                      \
    u##it_##1 = *reinterpret_cast<const float4*>(p0_ + 32);                                 \
    if (ST > 1) {                                                                           \
      const float* p1_ = k1_ ? yb + 64 + (int64_t)i1_ * LDY : zrow;                       \
      w##it_##0 = *reinterpret_cast<const float4*>(p1_);                                    \
      w##it_##1 = *reinterpret_cast<const float4*>(p1_ + 32);                               \
    }                                                                                       \
  }
// consume the step; leave the cursor of table slot 0 in (c, n) and of slot 1 in (d, m)
#define DESCO_CONSUME_TAB(it_)                                                              \
  {                                                                                         \
    DESCO_TAB_CUR(it_)                                                                      \
    f4add(lo##it_, u##it_##0);                                                              \
    f4add(hi##it_, u##it_##1);                                                              \
    if (ST > 1) {                                                                           \
      f4add(lo##it_, w##it_##0);                                                            \
      f4add(hi##it_, w##it_##1);                                                            \
    }                                                                                       \
    c##it_ = ca_ + (k0_ ? 1 : 0);                                                           \
    n##it_ = na_;                                                                           \
    d##it_ = na_ + (k1_ ? 1 : 0);                                                           \
    m##it_ = nb_;                                                                           \
  }
// heavy rows (hub / canonical rows of dense neighborhoods, or ids beyond the staged WCAP): the
// whole wave cooperates on one row at a time -- lane group k takes sources c+k, c+k+8, ... and
// the 8 partial sums are folded with three xor-shuffles (lanes with equal l8 hold the same columns)
#define DESCO_COOP(it_, base_, ld_)                                                       \
  {                                                                                       \
    unsigned long long m_ = __ballot(c##it_ < n##it_);                                    \
    while (m_) {                                                                          \
      const int sl_ = __builtin_ctzll(m_);                                                \
      const int og_ = sl_ >> 3;                                                           \
      const int cc_ = __shfl(c##it_, sl_, 64), nn_ = __shfl(n##it_, sl_, 64);             \
      float4 p_ = make_float4(0.f, 0.f, 0.f, 0.f), q_ = p_;                               \
      for (int e_ = cc_ + g8; e_ < nn_; e_ += 8) {                                        \
        const int64_t j_ = e_ < WCAP ? ec[e_] : g.vcol[ebase + e_];                       \
        const float* s_ = (base_) + j_ * (ld_);                                           \
        const float4 v0_ = *reinterpret_cast<const float4*>(s_);                          \
        const float4 v1_ = *reinterpret_cast<const float4*>(s_ + 32);                     \
        f4add(p_, v0_);                                                                   \
        f4add(q_, v1_);                                                                   \
      }                                                                                   \
      _Pragma("unroll") for (int o_ = 8; o_ < 64; o_ <<= 1) {                             \
        p_.x += __shfl_xor(p_.x, o_, 64);                                                 \
        p_.y += __shfl_xor(p_.y, o_, 64);                                                 \
        p_.z += __shfl_xor(p_.z, o_, 64);                                                 \
        p_.w += __shfl_xor(p_.w, o_, 64);                                                 \
        q_.x += __shfl_xor(q_.x, o_, 64);                                                 \
        q_.y += __shfl_xor(q_.y, o_, 64);                                                 \
        q_.z += __shfl_xor(q_.z, o_, 64);                                                 \
        q_.w += __shfl_xor(q_.w, o_, 64);                                                 \
      }                                                                                   \
      if (g8 == og_) {                                                                    \
        f4add(lo##it_, p_);                                                               \
        f4add(hi##it_, q_);                                                               \
        c##it_ = n##it_;                                                                  \
      }                                                                                   \
      m_ &= ~(0xffULL << (og_ * 8));                                                      \
    }                                                                                     \
  }
// finish a gathered block whose first step is already in flight: consume it, up to EXTRA_STEPS more
// batched steps (two sources per row each: all four rows of a lane group advance together), then
// the cooperative path for rows that are heavier still (one row at a time, the whole wave on it)
#define DESCO_FINISH(base_, ld_)                                                           \
  {                                                                                        \
    DESCO_CONSUME2(0) DESCO_CONSUME2(1) DESCO_CONSUME2(2) DESCO_CONSUME2(3)                \
    for (int st_ = 0; st_ < EXTRA_STEPS && DESCO_ANY_STAGED(); ++st_) {                    \
      DESCO_ISSUE2(0, base_, ld_) DESCO_ISSUE2(1, base_, ld_)                              \
      DESCO_ISSUE2(2, base_, ld_) DESCO_ISSUE2(3, base_, ld_)                              \
      DESCO_CONSUME2(0) DESCO_CONSUME2(1) DESCO_CONSUME2(2) DESCO_CONSUME2(3)              \
    }                                                                                      \
    if (__any((c0 < n0) | (c1 < n1) | (c2 < n2) | (c3 < n3))) {                            \
      DESCO_COOP(0, base_, ld_) DESCO_COOP(1, base_, ld_) DESCO_COOP(2, base_, ld_)        \
      DESCO_COOP(3, base_, ld_)                                                            \
    }                                                                                      \
  }
// first step of block b_ (cursors + loads); nothing waits on the loads here
#define DESCO_ISSUE_BLOCK(b_)                                                              \
  {                                                                                        \
    if ((b_) < KB - 1) {                                                                   \
      DESCO_CURS(b_)                                                                       \
      DESCO_ISSUE2(0, xb, LDX) DESCO_ISSUE2(1, xb, LDX)                                \
      DESCO_ISSUE2(2, xb, LDX) DESCO_ISSUE2(3, xb, LDX)                                \
    } else if ((b_) == KB - 1) {                                                           \
      DESCO_ISSUE_SELF(0) DESCO_ISSUE_SELF(1) DESCO_ISSUE_SELF(2) DESCO_ISSUE_SELF(3)      \
    } else {                                                                               \
      DESCO_ISSUE_TAB(0) DESCO_ISSUE_TAB(1) DESCO_ISSUE_TAB(2) DESCO_ISSUE_TAB(3)          \
    }                                                                                      \
  }
// bit s of `live`: relation slot s (an MFMA slot) has at least one source among the wave's 32 rows
#define DESCO_SLOT_LIVE(s_)                                                                \
  (__any((rp[(0 * 8 + g8) * S + (s_) + 1] > rp[(0 * 8 + g8) * S + (s_)]) |                  \
         (rp[(1 * 8 + g8) * S + (s_) + 1] > rp[(1 * 8 + g8) * S + (s_)]) |                  \
         (rp[(2 * 8 + g8) * S + (s_) + 1] > rp[(2 * 8 + g8) * S + (s_)]) |                  \
         (rp[(3 * 8 + g8) * S + (s_) + 1] > rp[(3 * 8 + g8) * S + (s_)]))                   \
       ? 1 << (s_) : 0)
#define DESCO_TILE_LIVE()                                        \
  {                                                              \
    live = 0;                                                    \
    if (KB - 1 > 0) live |= DESCO_SLOT_LIVE(0);                  \
    if (KB - 1 > 1) live |= DESCO_SLOT_LIVE(1);                  \
    if (KB - 1 > 2) live |= DESCO_SLOT_LIVE(2);                  \
  }
// first step of the first LIVE block after block a_ (a_ = -1: of the tile); dead slots are left
// out of the software pipeline altogether, so the block behind one is not issued late.  The slot
// index is a wave-uniform runtime value here (one copy of the gather issue code per site).
#define DESCO_ISSUE_AFTER(a_)                                                              \
  {                                                                                        \
    if ((a_) < KB - 1) {                                                                   \
      int nb_ = KB - 1;                                                                    \
      if ((a_) + 3 < KB - 1 && ((live >> ((a_) + 3)) & 1)) nb_ = (a_) + 3;                  \
      if ((a_) + 2 < KB - 1 && ((live >> ((a_) + 2)) & 1)) nb_ = (a_) + 2;                  \
      if ((a_) + 1 < KB - 1 && ((live >> ((a_) + 1)) & 1)) nb_ = (a_) + 1;                  \
      if (nb_ < KB - 1) {                                                                  \
        DESCO_CURS(nb_)                                                                    \
        DESCO_ISSUE2(0, xb, LDX) DESCO_ISSUE2(1, xb, LDX)                                  \
        DESCO_ISSUE2(2, xb, LDX) DESCO_ISSUE2(3, xb, LDX)                                  \
      } else {                                                                             \
        DESCO_ISSUE_SELF(0) DESCO_ISSUE_SELF(1) DESCO_ISSUE_SELF(2) DESCO_ISSUE_SELF(3)    \
      }                                                                                    \
    } else if ((a_) + 1 < NB) {                                                            \
      DESCO_ISSUE_BLOCK((a_) + 1)                                                          \
    }                                                                                      \
  }
// write one fp32 half image (every lane writes: row = it*8 + g8, 4 floats at 4*l8)
#define DESCO_PUT_F32(av_, it_)                         \
  {                                                     \
    float* d_ = Aw + ((it_) * 8 + g8) * AH + 4 * l8;    \
    d_[0] = av_.x;                                      \
    d_[1] = av_.y;                                      \
    d_[2] = av_.z;                                      \
    d_[3] = av_.w;                                      \
  }
// write one half image as three bf16 planes (row = it*8 + g8, 4 bf16 at 4*l8 of every plane)
#define DESCO_PUT_X6(av_, it_)                                                  \
  {                                                                             \
    uint32_t h0_, m0_, l0_, h1_, m1_, l1_;                                      \
    split2_bf16x3(av_.x, av_.y, h0_, m0_, l0_);                                   \
    split2_bf16x3(av_.z, av_.w, h1_, m1_, l1_);                                   \
    short* d_ = Ap + ((it_) * 8 + g8) * APS + 4 * l8;                           \
    *reinterpret_cast<uint2*>(d_) = make_uint2(h0_, h1_);                       \
    *reinterpret_cast<uint2*>(d_ + WR * APS) = make_uint2(m0_, m1_);            \
    *reinterpret_cast<uint2*>(d_ + 2 * WR * APS) = make_uint2(l0_, l1_);        \
  }
// 32 f32 MFMAs on the staged half h_ of K block b_: A[i = lane&31][k = lane>>5], B[k][j = lane&31]
#define DESCO_MFMA_HALF_F32(b_, h_)                                                        \
  {                                                                                        \
    const float* as_ = Aw + (lane & 31) * AH + (lane >> 5);                                \
    const float* bs_ = Bimg + ((b_) * 64 + (h_) * 32 + (lane >> 5)) * 64 + (lane & 31);    \
    _Pragma("unroll") for (int kk = 0; kk < 16; ++kk) {                                    \
      const float a_ = as_[2 * kk];                                                        \
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_, bs_[2 * kk * 64], acc0, 0, 0, 0);    \
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_, bs_[2 * kk * 64 + 32], acc1, 0, 0, 0); \
    }                                                                                      \
  }
// 24 bf16 MFMAs (6-product split) on the staged half: lane (r = lane&31, q = lane>>5) holds
// A[row r][k = 16 s + 8 q + 0..7] and B[k = 16 s + 8 q + 0..7][col r] of every plane
#define DESCO_MFMA_HALF_X6(b_, h_)                                                                \
  {                                                                                               \
    const short* ap_ = Ap + (lane & 31) * APS + 8 * (lane >> 5);                                  \
    const short* bp_ = Wp + (lane & 31) * WST + (b_) * 64 + (h_) * 32 + 8 * (lane >> 5);          \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                            \
      const bf16x8 ah_ = *reinterpret_cast<const bf16x8*>(ap_ + 16 * s_);                         \
      const bf16x8 am_ = *reinterpret_cast<const bf16x8*>(ap_ + WR * APS + 16 * s_);              \
      const bf16x8 al_ = *reinterpret_cast<const bf16x8*>(ap_ + 2 * WR * APS + 16 * s_);          \
      const bf16x8 b0h_ = *reinterpret_cast<const bf16x8*>(bp_ + 16 * s_);                        \
      const bf16x8 b0m_ = *reinterpret_cast<const bf16x8*>(bp_ + WPL + 16 * s_);                  \
      const bf16x8 b0l_ = *reinterpret_cast<const bf16x8*>(bp_ + 2 * WPL + 16 * s_);              \
      const bf16x8 b1h_ = *reinterpret_cast<const bf16x8*>(bp_ + 32 * WST + 16 * s_);             \
      const bf16x8 b1m_ = *reinterpret_cast<const bf16x8*>(bp_ + 32 * WST + WPL + 16 * s_);       \
      const bf16x8 b1l_ = *reinterpret_cast<const bf16x8*>(bp_ + 32 * WST + 2 * WPL + 16 * s_);   \
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al_, b0h_, acc0, 0, 0, 0);                   \
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al_, b1h_, acc1, 0, 0, 0);                   \
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_, b0l_, acc0, 0, 0, 0);                   \
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_, b1l_, acc1, 0, 0, 0);                   \
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am_, b0m_, acc0, 0, 0, 0);                   \
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am_, b1m_, acc1, 0, 0, 0);                   \
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am_, b0h_, acc0, 0, 0, 0);                   \
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am_, b1h_, acc1, 0, 0, 0);                   \
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_, b0m_, acc0, 0, 0, 0);                   \
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_, b1m_, acc1, 0, 0, 0);                   \
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_, b0h_, acc0, 0, 0, 0);                   \
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_, b1h_, acc1, 0, 0, 0);                   \
    }                                                                                             \
  }
// add the staged (fp32) table half rows in the C/D layout: acc[reg] += stage[row(reg)][lane&31]
#define DESCO_TAB_HALF(acc_)                                                               \
  {                                                                                        \
    _Pragma("unroll") for (int reg = 0; reg < 16; ++reg)                                   \
        acc_[reg] += Aw[((reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) * AH + cl];         \
  }

// KB = sm + 1 resident weight blocks (1..4), ST table slots (0..2), X6: bf16 6-product arithmetic,
// LD64: x rows are 64 floats and ytab rows 64*ST floats apart (the product path's layouts): source-row
// addresses then need a shift instead of a 64-bit multiply per gathered row
// POOL: fused pooling epilogue (instantiated for the count-row launches of the x6 form only)
template <int KB, int ST, bool X6, bool LD64, bool POOL = false>
__global__ __launch_bounds__(NW * 64) void shmp_layer_f32_kernel(ShmpArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int WST = KB * 64 + 8;                         // weight plane row stride (shorts)
  constexpr int WPL = 64 * WST;                            // shorts per weight plane
  constexpr int W_FLOATS = X6 ? 3 * WPL / 2 : KB * 64 * 64;
  float* Bimg = lds;                                       // f32: [KB*64][64]
  short* Wp = reinterpret_cast<short*>(lds);               // x6:  [3][64 n][WST]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* Aw = lds + W_FLOATS + wave * WAVE_LDS;            // fp32 half image [32][33] (f32 mode, table block)
  short* Ap = reinterpret_cast<short*>(Aw);                // x6: bf16 planes [3][32][40] of a half image
  int* rpb = reinterpret_cast<int*>(Aw + A_FLOATS);        // 2 x [32*S+1] row pointers (absolute)
  int* ec = rpb + 2 * RPN;                                 // [WCAP] source ids of the current tile
  (void)Bimg; (void)Wp; (void)Ap;

  // ---- resident weights -------------------------------------------------------------------
  if (X6) {
    // global planes [3][64][KB*64] -> LDS [3][64][WST], 16 bytes at a time
    constexpr int CH = KB * 8;                             // uint4 chunks per row
    for (int i = tid; i < 3 * 64 * CH; i += NW * 64) {
      const int row = i / CH, ch = i - row * CH;           // row = plane*64 + n
      *reinterpret_cast<uint4*>(Wp + row * WST + 8 * ch) =
          *reinterpret_cast<const uint4*>(g.wplanes + (int64_t)row * (KB * 64) + 8 * ch);
    }
  } else {
    for (int i = tid; i < KB * 1024; i += NW * 64)
      *reinterpret_cast<float4*>(Bimg + 4 * i) = *reinterpret_cast<const float4*>(g.wt + 4 * i);
  }
  __syncthreads();

  const int g8 = lane >> 3, l8 = lane & 7;                 // 8 groups of 8 lanes: one half row each
  const int cl = lane & 31;
  const int S = g.S;
  const int nslot = WR * S + 1;                            // <= 129: at most 3 per lane
  const int64_t ntiles = (g.num_rows + NW * WR - 1) / (NW * WR);
  constexpr int NB = KB + (ST > 0 ? 1 : 0);                // K blocks incl. the table pseudo block
  const int64_t LDX = LD64 ? 64 : g.ldx, LDY = LD64 ? 64 * (ST > 0 ? ST : 1) : g.ldy;
  const float* xb = g.x + 4 * l8;
  const float* yb = ST > 0 ? g.ytab + 4 * l8 - g.ytab_row0 * LDY : nullptr;
  const float* zrow = shmp_zero_row + 4 * l8;
  (void)yb;

  // ---- this wave's first tile ---------------------------------------------------------------
  // XCD-aware tile order: blocks b, b+8, b+16, ... share an XCD (round-robin dispatch) and its 4 MB L2,
  // so a neighborhood's rows -- the sources of all its tiles -- should be gathered by ONE XCD.  Measured
  // (profiles/r2_f_ab_xcd_order.log): +0.5 % on Syn_1827 / MSRC+IMDB shapes, 0 on COX2 shapes (the
  // gathers are not what bounds the kernel); contiguous eighths per XCD were 8 % SLOWER on Syn shapes
  // (the dataset is ordered by graph size: the XCD with the dense end finishes last).  Speed only: any
  // block -> XCD placement gives the same result.
  int64_t tile, tend = ntiles;
  int tstride = gridDim.x;
  if ((gridDim.x & 7) == 0) {
    // chunks of (grid / 8) consecutive tiles go round robin over the XCDs: XCD x works on the 32
    // neighbouring tiles of chunk 8 j + x in sweep j (locality), heavy and light regions of the dataset
    // are spread over all XCDs (balance)
    tile = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  } else {
    tile = blockIdx.x;
  }
  int64_t w0 = tile * (NW * WR) + wave * WR;               // first row of this wave (relative)
  if (tile >= tend || w0 >= g.num_rows) return;            // no barrier below: idle waves may leave
  int nr = (int)((g.num_rows - w0) < WR ? (g.num_rows - w0) : WR);
  int64_t grow0 = g.row0 + w0;
  int cur = 0;
  int* rp = rpb;
  if (S > 0) {                                             // S == 0: no CSR at all (plain row-wise Linear)
    const int nptr = nr * S + 1;
    for (int i = lane; i < nslot; i += 64)
      rp[i] = g.vrowptr[grow0 * S + (i < nptr ? i : nptr - 1)];
    const int eb = rp[0], ecnt = rp[WR * S] - eb;
    for (int i = lane; i < ecnt && i < WCAP; i += 64) ec[i] = g.vcol[eb + i];
  } else if (lane == 0) {
    rp[0] = 0;
    rpb[RPN] = 0;
  }
  int ebase = rp[0];

  float4 lo0, lo1, lo2, lo3, hi0, hi1, hi2, hi3;           // gathered sums of the current block
  float4 u00, u01, u10, u11, u20, u21, u30, u31;           // in flight: first source (lo, hi) of row it
  float4 w00, w01, w10, w11, w20, w21, w30, w31;           // in flight: second source
  w00 = w01 = w10 = w11 = w20 = w21 = w30 = w31 = make_float4(0.f, 0.f, 0.f, 0.f);
  int c0 = 0, c1 = 0, c2 = 0, c3 = 0, n0 = 0, n1 = 0, n2 = 0, n3 = 0;   // cursors [c, n) rel. to ebase
  // bit b: relation slot b has at least one source among this wave's 32 rows.  A slot that is empty
  // for the whole wave tile (triangle edges in molecule graphs, tride edges in clique unions) is an
  // all-zero K block: it is left out of the tile's block sequence (wave-uniform; it would add exact
  // zeros), and the first gather step of the block behind it is issued in its place
  int live = 0;
  DESCO_TILE_LIVE()
  DESCO_ISSUE_AFTER(-1)

  for (;;) {
    int* rpn = rpb + (cur ^ 1) * RPN;
    // ---- prefetch the row pointers of the next tile (registers now, LDS later) --------------
    const int64_t tn = tile + tstride;
    const int64_t w0n = tn * (NW * WR) + wave * WR;
    const bool has_next = tn < tend && w0n < g.num_rows;
    const int nrn = has_next ? (int)((g.num_rows - w0n) < WR ? (g.num_rows - w0n) : WR) : 0;
    int p0 = 0, p1 = 0, p2 = 0;
    if (has_next && S > 0) {
      const int nptr = nrn * S + 1;
      const int32_t* src = g.vrowptr + (g.row0 + w0n) * S;
      p0 = src[lane < nptr ? lane : nptr - 1];
      if (lane + 64 < nslot) p1 = src[lane + 64 < nptr ? lane + 64 : nptr - 1];
      if (lane + 128 < nslot) p2 = src[lane + 128 < nptr ? lane + 128 : nptr - 1];
    }
    int qn0 = 0, qn1 = 0, qn2 = 0, qn3 = 0;   // source ids of the next tile (registers until the tile ends)
    int ebn = 0, ecntn = 0;
    // fused pooling: this tile's segment-end bitmap and first partial slot (wave-uniform address:
    // scalar loads, in flight under the whole tile)
    uint32_t pool_e = 0;
    int pool_s = 0;
    if constexpr (POOL) {
      const int t32 = __builtin_amdgcn_readfirstlane((int)(grow0 >> 5));
      pool_e = g.pool_bits[t32];
      pool_s = g.pool_slot[t32];
    }

    // ---- accumulator init: bias ----------------------------------------------------------------
    f32x16 acc0, acc1;
    {
      const float bv0 = g.bias ? g.bias[cl] : 0.f, bv1 = g.bias ? g.bias[32 + cl] : 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc0[i] = bv0;
        acc1[i] = bv1;
      }
    }
    // K blocks: b < KB-1 = relation slot b (gathered x rows), b == KB-1 = the row itself,
    // b == KB (ST > 0) = table pseudo block (gathered ytab rows, added in the C/D layout)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b < KB - 1 && !((live >> b) & 1)) continue;      // empty relation slot (wave-uniform)
      // ---- complete the gather of block b ------------------------------------------------------
      DESCO_ZERO_SUMS()
      if (b < KB - 1) {
        DESCO_FINISH(xb, LDX)
      } else if (b == KB - 1) {
        lo0 = u00; hi0 = u01; lo1 = u10; hi1 = u11;
        lo2 = u20; hi2 = u21; lo3 = u30; hi3 = u31;
      } else {
        // canonical->count relations have at most one source per row: one step covers both table
        // slots; anything beyond that (general inputs) takes the cooperative path
        int d0, d1, d2, d3, m0, m1, m2, m3;
        DESCO_CONSUME_TAB(0) DESCO_CONSUME_TAB(1) DESCO_CONSUME_TAB(2) DESCO_CONSUME_TAB(3)
        if (__any((c0 < n0) | (c1 < n1) | (c2 < n2) | (c3 < n3))) {
          DESCO_COOP(0, yb, LDY) DESCO_COOP(1, yb, LDY) DESCO_COOP(2, yb, LDY)
          DESCO_COOP(3, yb, LDY)
        }
        if (ST > 1 && __any((d0 < m0) | (d1 < m1) | (d2 < m2) | (d3 < m3))) {
          c0 = d0; c1 = d1; c2 = d2; c3 = d3;
          n0 = m0; n1 = m1; n2 = m2; n3 = m3;
          DESCO_COOP(0, yb + 64, LDY) DESCO_COOP(1, yb + 64, LDY) DESCO_COOP(2, yb + 64, LDY)
          DESCO_COOP(3, yb + 64, LDY)
        }
      }
      if (b == KB - 1 && has_next) {
        // next tile's row pointers have landed: publish them, then fetch its source ids
        rpn[lane] = p0;
        if (lane + 64 < nslot) rpn[lane + 64] = p1;
        if (lane + 128 < nslot) rpn[lane + 128] = p2;
        ebn = rpn[0];
        ecntn = rpn[WR * S] - ebn;
        if (lane < ecntn) qn0 = g.vcol[ebn + lane];
        if (lane + 64 < ecntn) qn1 = g.vcol[ebn + lane + 64];
        if (lane + 128 < ecntn) qn2 = g.vcol[ebn + lane + 128];
        if (lane + 192 < ecntn) qn3 = g.vcol[ebn + lane + 192];
      }
      // ---- the two 32-column halves of block b; the first gather step of block b+1 goes out
      //      under this block's MFMAs (after the low halves have left their registers)
      {
        if (X6 && b < KB) {
          DESCO_PUT_X6(lo0, 0) DESCO_PUT_X6(lo1, 1) DESCO_PUT_X6(lo2, 2) DESCO_PUT_X6(lo3, 3)
        } else {
          DESCO_PUT_F32(lo0, 0) DESCO_PUT_F32(lo1, 1) DESCO_PUT_F32(lo2, 2) DESCO_PUT_F32(lo3, 3)
        }
      }
      DESCO_ISSUE_AFTER(b)
      {
        if (b >= KB) {
          DESCO_TAB_HALF(acc0)
        } else if (X6) {
          DESCO_MFMA_HALF_X6(b, 0)
        } else {
          DESCO_MFMA_HALF_F32(b, 0)
        }
        if (X6 && b < KB) {
          DESCO_PUT_X6(hi0, 0) DESCO_PUT_X6(hi1, 1) DESCO_PUT_X6(hi2, 2) DESCO_PUT_X6(hi3, 3)
        } else {
          DESCO_PUT_F32(hi0, 0) DESCO_PUT_F32(hi1, 1) DESCO_PUT_F32(hi2, 2) DESCO_PUT_F32(hi3, 3)
        }
        if (b >= KB) {
          DESCO_TAB_HALF(acc1)
        } else if (X6) {
          DESCO_MFMA_HALF_X6(b, 1)
        } else {
          DESCO_MFMA_HALF_F32(b, 1)
        }
      }
    }

    // ---- switch to the next tile: publish its source ids and launch its first gather step ------
    const int64_t grow_out = grow0;
    const int nr_out = nr;
    if (has_next) {
      if (lane < ecntn) ec[lane] = qn0;            // (the current tile's ids are dead by now)
      if (lane + 64 < ecntn) ec[lane + 64] = qn1;
      if (lane + 128 < ecntn) ec[lane + 128] = qn2;
      if (lane + 192 < ecntn) ec[lane + 192] = qn3;
      if (ecntn > 256) {                           // dense tile (wave-uniform): ids 256..511 now
        for (int i = 256 + lane; i < ecntn && i < WCAP; i += 64) ec[i] = g.vcol[ebn + i];
      }
      cur ^= 1;
      rp = rpn;
      ebase = ebn;
      tile = tn;
      w0 = w0n;
      nr = nrn;
      grow0 = g.row0 + w0n;
      DESCO_TILE_LIVE()
      DESCO_ISSUE_AFTER(-1)
    }

    // ---- epilogue: relu + store; C/D map col = lane&31, row = (reg&3)+8*(reg>>2)+4*(lane>>5) ----
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      acc0[reg] = apply_act(acc0[reg], g.act, g.slope);
      acc1[reg] = apply_act(acc1[reg], g.act, g.slope);
    }
    if (!POOL || g.out) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (r < nr_out) {
          float* o = g.out + (grow_out + r) * g.ldo + cl;
          o[0] = acc0[reg];
          o[32] = acc1[reg];
          if (g.out2) {        // e.g. the canonical rows' column block of the anchor-MLP operand
            float* o2 = g.out2 + (grow_out - g.row0 + r) * g.ldo2 + cl;
            o2[0] = acc0[reg];
            o2[32] = acc1[reg];
          }
        }
      }
    }
    if constexpr (POOL) {
      // lane halves swapped: acc0[reg] = row (reg&3)+8*(reg>>2), acc1[reg] = that row + 4, column = lane
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 t_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc0[reg]), __float_as_uint(acc1[reg]),
                                                          false, false);
        acc0[reg] = __uint_as_float(t_[0]);
        acc1[reg] = __uint_as_float(t_[1]);
      }
      const uint32_t E = __builtin_amdgcn_readfirstlane(pool_e);
      int slot = __builtin_amdgcn_readfirstlane(pool_s);
      const int nru = __builtin_amdgcn_readfirstlane(nr_out);
      float* pp = g.pool_part + lane;
      float run = 0.f;
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        const int reg = (r & 3) + 4 * (r >> 3);
        if (r < nru) {                                        // (wave-uniform)
          run += ((r >> 2) & 1) ? acc1[reg] : acc0[reg];
          if ((E >> r) & 1u) {                                // row r ends its segment (wave-uniform)
            pp[(int64_t)slot * 64] = run;
            ++slot;
            run = 0.f;
          }
        }
      }
      // the last segment of the tile continues in the next tile: its partial so far
      if (nru > 0 && !((E >> (nru - 1)) & 1u)) pp[(int64_t)slot * 64] = run;
    }
    if (!has_next) break;
  }
}

// ---- streaming row-wise Linear, K = 64 (desco_linear64_bf16x6_f32) ---------------------------------
// The layer kernel's self block on its own: no CSR, NJ resident 64x64 weight blocks (NJ output
// column blocks per pass over x, so x is read once per NJ*64 outputs), rows of the next tile in
// flight under the current tile's MFMAs and stores.
struct Lin64Args {
  const float* x;
  int64_t ldx;
  const short* wplanes;     // [NJ][3][64 n][64 k]
  const float* bias;        // [NJ*64] or null
  float* out;               // first of the NJ column blocks
  int64_t ldo;
  int64_t num_rows;
  int act;
  float slope;
};

template <int NJ>
__global__ __launch_bounds__(NW * 64) void linear64_kernel(Lin64Args g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int WST = 72, WPL = 64 * WST;                  // one weight block: [3][64][72] shorts
  constexpr int W_FLOATS = NJ * 3 * WPL / 2;
  short* wbase = reinterpret_cast<short*>(lds);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* Aw = lds + W_FLOATS + wave * A_FLOATS;
  short* Ap = reinterpret_cast<short*>(Aw);
  for (int i = tid; i < NJ * 3 * 64 * 8; i += NW * 64) {   // [NJ*3*64 rows][8 chunks of 8 shorts]
    const int row = i >> 3, ch = i & 7;
    *reinterpret_cast<uint4*>(wbase + row * WST + 8 * ch) =
        *reinterpret_cast<const uint4*>(g.wplanes + (int64_t)row * 64 + 8 * ch);
  }
  __syncthreads();
  const int g8 = lane >> 3, l8 = lane & 7, cl = lane & 31;
  const int64_t ntiles = (g.num_rows + NW * WR - 1) / (NW * WR);
  int64_t tile = blockIdx.x;
  int64_t w0 = tile * (NW * WR) + wave * WR;
  if (tile >= ntiles || w0 >= g.num_rows) return;
  const float* xb = g.x + 4 * l8;
  float4 u00, u01, u10, u11, u20, u21, u30, u31;
#define DESCO_LIN_LOAD(it_)                                                        \
  {                                                                                \
    const int64_t r_ = w0 + (it_) * 8 + g8;                                        \
    const float* p_ = xb + (r_ < g.num_rows ? r_ : g.num_rows - 1) * g.ldx;        \
    u##it_##0 = *reinterpret_cast<const float4*>(p_);                              \
    u##it_##1 = *reinterpret_cast<const float4*>(p_ + 32);                         \
  }
  // bias of the lane's two columns per output block, once: a load inside the tile loop is waited for at
  // the accumulator init, and vector memory returns in order -- the wait would also drain the prefetch of
  // the next tile's rows issued just before it
  float bv[NJ][2];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    bv[j][0] = g.bias ? g.bias[64 * j + cl] : 0.f;
    bv[j][1] = g.bias ? g.bias[64 * j + 32 + cl] : 0.f;
  }
  DESCO_LIN_LOAD(0) DESCO_LIN_LOAD(1) DESCO_LIN_LOAD(2) DESCO_LIN_LOAD(3)
  for (;;) {
    const float4 lo0 = u00, hi0 = u01, lo1 = u10, hi1 = u11, lo2 = u20, hi2 = u21, lo3 = u30, hi3 = u31;
    const int64_t row_out = w0;
    const int nr = (int)((g.num_rows - w0) < WR ? (g.num_rows - w0) : WR);
    const int64_t tn = tile + gridDim.x;
    const int64_t w0n = tn * (NW * WR) + wave * WR;
    const bool has_next = tn < ntiles && w0n < g.num_rows;
    if (has_next) {
      tile = tn;
      w0 = w0n;
      DESCO_LIN_LOAD(0) DESCO_LIN_LOAD(1) DESCO_LIN_LOAD(2) DESCO_LIN_LOAD(3)
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const short* Wp = wbase + j * 3 * WPL;
      f32x16 acc0, acc1;
      {
        const float bv0 = bv[j][0], bv1 = bv[j][1];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          acc0[i] = bv0;
          acc1[i] = bv1;
        }
      }
      DESCO_PUT_X6(lo0, 0) DESCO_PUT_X6(lo1, 1) DESCO_PUT_X6(lo2, 2) DESCO_PUT_X6(lo3, 3)
      DESCO_MFMA_HALF_X6(0, 0)
      DESCO_PUT_X6(hi0, 0) DESCO_PUT_X6(hi1, 1) DESCO_PUT_X6(hi2, 2) DESCO_PUT_X6(hi3, 3)
      DESCO_MFMA_HALF_X6(0, 1)
      if (nr == WR) {                       // whole tile: one base address, no per-row tests
        float* o = g.out + (row_out + 4 * (lane >> 5)) * g.ldo + 64 * j + cl;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int r = (reg & 3) + 8 * (reg >> 2);
          __builtin_nontemporal_store(apply_act(acc0[reg], g.act, g.slope), o + r * g.ldo);
          __builtin_nontemporal_store(apply_act(acc1[reg], g.act, g.slope), o + r * g.ldo + 32);
        }
      } else {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int r = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
          if (r < nr) {
            float* o = g.out + (row_out + r) * g.ldo + 64 * j + cl;
            o[0] = apply_act(acc0[reg], g.act, g.slope);
            o[32] = apply_act(acc1[reg], g.act, g.slope);
          }
        }
      }
    }
    if (!has_next) break;
  }
#undef DESCO_LIN_LOAD
}

#undef DESCO_CUR
#undef DESCO_CURS
#undef DESCO_ISSUE2
#undef DESCO_CONSUME2
#undef DESCO_ZERO_SUMS
#undef DESCO_ANY_STAGED
#undef DESCO_ISSUE_SELF
#undef DESCO_TAB_CUR
#undef DESCO_ISSUE_TAB
#undef DESCO_CONSUME_TAB
#undef DESCO_COOP
#undef DESCO_FINISH
#undef DESCO_ISSUE_BLOCK
#undef DESCO_ISSUE_AFTER
#undef DESCO_TILE_LIVE
#undef DESCO_SLOT_LIVE
#undef DESCO_PUT_F32
#undef DESCO_PUT_X6
#undef DESCO_MFMA_HALF_F32
#undef DESCO_MFMA_HALF_X6
#undef DESCO_TAB_HALF

template <int KB, int ST, bool X6, bool LD64, bool POOL = false>
static void shmp_launch_one(const ShmpArgs& g, unsigned grid, hipStream_t st) {
  constexpr int WST = KB * 64 + 8;
  constexpr size_t w_floats = X6 ? (size_t)3 * 64 * WST / 2 : (size_t)KB * 64 * 64;
  constexpr size_t shmem = sizeof(float) * (w_floats + (size_t)NW * WAVE_LDS);
  static_assert(shmem <= 160 * 1024, "SHMP layer: LDS budget exceeded");
  static DeviceOnce attr_once;        // function attributes are per device
  if (!attr_once.done()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(shmp_layer_f32_kernel<KB, ST, X6, LD64, POOL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_once.mark();
  }
  hipLaunchKernelGGL((shmp_layer_f32_kernel<KB, ST, X6, LD64, POOL>), dim3(grid), dim3(NW * 64), shmem, st, g);
}

template <int KB, bool X6>
static void shmp_launch_st(const ShmpArgs& g, unsigned grid, hipStream_t st) {
  const bool ld64 = g.ldx == 64 && (g.st == 0 || g.ldy == 64 * g.st);
  if constexpr (KB == 3 && X6) {
    if (g.pool_part) {       // count-row launches of the product path (validated by shmp_launch)
      if (ld64)
        shmp_launch_one<3, 2, true, true, true>(g, grid, st);
      else
        shmp_launch_one<3, 2, true, false, true>(g, grid, st);
      return;
    }
  }
#define DESCO_ONE(ST_)                                         \
  if (ld64)                                                    \
    shmp_launch_one<KB, ST_, X6, true>(g, grid, st);           \
  else                                                         \
    shmp_launch_one<KB, ST_, X6, false>(g, grid, st);
  switch (g.st) {
    case 0: DESCO_ONE(0) break;
    case 1: DESCO_ONE(1) break;
    default:
      if constexpr (KB <= 3) {           // sm + st <= 4 slots: four weight blocks leave room for one table slot
        DESCO_ONE(2)
      }
      break;
  }
#undef DESCO_ONE
}

// Tile form of the x6 launches: 16-row wave tiles (shmp_layer16.hip) unless DESCO_SHMP_ROWS=32 asks for
// the 32-row kernel of this file (A/B runs; read once per process).
static int shmp_tile_rows() {
  static const int rows = [] {
    const char* e = getenv("DESCO_SHMP_ROWS");
    return e && atoi(e) == 32 ? 32 : 16;
  }();
  return rows;
}

static int shmp_launch(const char* who, bool x6, const float* x, int64_t ldx, const int32_t* vrowptr,
                       const int32_t* vcol, int64_t row0, int64_t num_rows, int slots_stored,
                       int slots_mfma, int slots_table, const void* weights, const float* bias,
                       const float* ytab, int64_t ldy, int64_t ytab_row0, float* out, int64_t ldo,
                       float* out2, int64_t ldo2, int act, float slope, desco_stream_t stream,
                       const uint32_t* pool_bits = nullptr, const int32_t* pool_slot = nullptr,
                       float* pool_part = nullptr, const float* wscale = nullptr, float* row_absmax = nullptr,
                       const float* xself = nullptr, int64_t ldxs = 0, const float* self_coef = nullptr) {
  if (num_rows == 0) return 0;
  auto mis16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  const int max_mfma = x6 ? 2 : 3;
  const bool pool = pool_part != nullptr;
  const int tile_rows = wscale ? 16 : x6 ? shmp_tile_rows() : 32;      // (the fp16 form exists for 16-row tiles only)
  if (pool && (!pool_bits || !pool_slot || row0 % tile_rows || mis16(pool_part) || out2 || !x6 || slots_mfma != 2 ||
               slots_table != 2))
    return fail(DESCO_EINVAL, "desco_shmp_layer_pool_bf16x6_f32: bad pooling argument (row0 % tile rows, no out2, "
                              "slots_mfma == 2, slots_table == 2)");
  if (xself && (!(x6 && (wscale || shmp_tile_rows() == 16)) || ldxs % 4 || mis16(xself)))
    return fail(DESCO_EINVAL, "desco_shmp_layer_*: xself is implemented by the 16-row form only (16-byte rows)");
  if (self_coef && (xself || !pool || !wscale || mis16(self_coef)))
    return fail(DESCO_EINVAL, "desco_shmp_layer_*: self_coef is implemented by the pooled f16x3 launch only (no xself)");
  if (!x || (!vrowptr && slots_stored > 0) || !weights || (!out && !pool && !out2) || row0 < 0 || num_rows < 0 || slots_mfma < 0 ||
      slots_mfma > max_mfma || slots_table < 0 || slots_mfma + slots_table > slots_stored ||
      slots_stored < 0 || slots_stored > MAXS || (slots_stored == 0 && (slots_mfma || slots_table)) || slots_table > 2 || (slots_table > 0 && !ytab) ||
      ldx % 4 || (slots_table > 0 && ldy % 4) || mis16(x) || mis16(weights) ||
      (slots_table > 0 && mis16(ytab)) || x == out || x == out2 ||
      (out && (mis16(out) || ldo % 4)) || (out2 && (mis16(out2) || ldo2 % 4)))      // float4 stores
    return fail(DESCO_EINVAL,
                x6 ? "desco_shmp_layer_bf16x6_f32: bad argument (slots_mfma <= 2, slots_table <= 2)"
                   : "desco_shmp_layer_f32: bad argument (slots_mfma <= 3, slots_table <= 2)");
  const int64_t ntiles = (num_rows + NW * WR - 1) / (NW * WR);
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      cus = v;
  }
  const unsigned grid = (unsigned)(ntiles < cus ? ntiles : cus);
  ShmpArgs g{x,
             ldx,
             vrowptr,
             vcol,
             row0,
             num_rows,
             slots_stored,
             slots_mfma,
             slots_table,
             x6 ? nullptr : static_cast<const float*>(weights),
             x6 ? static_cast<const short*>(weights) : nullptr,
             wscale,
             bias,
             ytab,
             ldy,
             ytab_row0,
             out,
             ldo,
             out2,
             ldo2,
             row_absmax,
             act,
             slope,
             pool_bits,
             pool_slot,
             pool_part,
             tile_rows,
             xself,
             ldxs,
             self_coef};
  if (!out && !pool && !(x6 && tile_rows == 16))
    return fail(DESCO_EINVAL, "desco_shmp_layer_*: out == NULL (rows to out2 alone) is implemented by the 16-row form only");
  hipStream_t st = (hipStream_t)stream;
  if (x6 && tile_rows == 16) {
    if (!shmp16_launch(g, cus, stream)) return fail(DESCO_EINVAL, "desco_shmp_layer_bf16x6_f32: shape not built");
    return launch_status(who);
  }
  if (x6) {
    switch (slots_mfma) {
      case 0: shmp_launch_st<1, true>(g, grid, st); break;
      case 1: shmp_launch_st<2, true>(g, grid, st); break;
      default: shmp_launch_st<3, true>(g, grid, st); break;
    }
  } else {
    switch (slots_mfma) {
      case 0: shmp_launch_st<1, false>(g, grid, st); break;
      case 1: shmp_launch_st<2, false>(g, grid, st); break;
      case 2: shmp_launch_st<3, false>(g, grid, st); break;
      default: shmp_launch_st<4, false>(g, grid, st); break;
    }
  }
  return launch_status(who);
}

}  // namespace desco

extern "C" int desco_shmp_layer_f32(const float* x, int64_t ldx, const int32_t* vrowptr,
                                    const int32_t* vcol, int64_t row0, int64_t num_rows,
                                    int slots_stored, int slots_mfma, int slots_table,
                                    const float* wt, const float* bias, const float* ytab,
                                    int64_t ldy, int64_t ytab_row0, float* out, int64_t ldo,
                                    float* out2, int64_t ldo2, desco_stream_t stream) {
  return desco::shmp_launch("desco_shmp_layer_f32", false, x, ldx, vrowptr, vcol, row0, num_rows,
                            slots_stored, slots_mfma, slots_table, wt, bias, ytab, ldy, ytab_row0, out,
                            ldo, out2, ldo2, DESCO_ACT_RELU, 0.f, stream);
}

extern "C" int desco_shmp_layer_bf16x6_f32(const float* x, int64_t ldx, const int32_t* vrowptr,
                                           const int32_t* vcol, int64_t row0, int64_t num_rows,
                                           int slots_stored, int slots_mfma, int slots_table,
                                           const int16_t* wt_planes, const float* bias,
                                           const float* ytab, int64_t ldy, int64_t ytab_row0,
                                           float* out, int64_t ldo, float* out2, int64_t ldo2,
                                           desco_stream_t stream) {
  return desco::shmp_launch("desco_shmp_layer_bf16x6_f32", true, x, ldx, vrowptr, vcol, row0,
                            num_rows, slots_stored, slots_mfma, slots_table, wt_planes, bias, ytab,
                            ldy, ytab_row0, out, ldo, out2, ldo2, DESCO_ACT_RELU, 0.f, stream);
}

extern "C" int desco_shmp_layer_pool_bf16x6_f32(const float* x, int64_t ldx, const int32_t* vrowptr,
                                                const int32_t* vcol, int64_t row0, int64_t num_rows,
                                                int slots_stored, int slots_mfma, int slots_table,
                                                const int16_t* wt_planes, const float* bias,
                                                const float* ytab, int64_t ldy, int64_t ytab_row0,
                                                float* out, int64_t ldo, const uint32_t* pool_bits,
                                                const int32_t* pool_slot, float* pool_part,
                                                desco_stream_t stream) {
  if (!pool_part)
    return desco::fail(DESCO_EINVAL, "desco_shmp_layer_pool_bf16x6_f32: pool_part is null");
  return desco::shmp_launch("desco_shmp_layer_pool_bf16x6_f32", true, x, ldx, vrowptr, vcol, row0,
                            num_rows, slots_stored, slots_mfma, slots_table, wt_planes, bias, ytab,
                            ldy, ytab_row0, out, ldo, nullptr, 0, DESCO_ACT_RELU, 0.f, stream,
                            pool_bits, pool_slot, pool_part);
}

// The same layer in the three-product fp16 form (16-row tiles): wt_planes[2][64 n][(slots_mfma+1)*64 k] and
// w_scale[2] = {scale, 1/scale} (device) from desco_split_f16x2_f32; row scales are found in the kernel.
extern "C" int desco_shmp_layer_f16x3_f32(const float* x, int64_t ldx, const int32_t* vrowptr,
                                          const int32_t* vcol, int64_t row0, int64_t num_rows,
                                          int slots_stored, int slots_mfma, int slots_table,
                                          const int16_t* wt_planes, const float* w_scale, const float* bias,
                                          const float* ytab, int64_t ldy, int64_t ytab_row0,
                                          float* out, int64_t ldo, float* out2, int64_t ldo2,
                                          float* row_absmax, const float* xself, int64_t ldxs,
                                          desco_stream_t stream) {
  if (!w_scale) return desco::fail(DESCO_EINVAL, "desco_shmp_layer_f16x3_f32: w_scale is null");
  return desco::shmp_launch("desco_shmp_layer_f16x3_f32", true, x, ldx, vrowptr, vcol, row0,
                            num_rows, slots_stored, slots_mfma, slots_table, wt_planes, bias, ytab,
                            ldy, ytab_row0, out, ldo, out2, ldo2, DESCO_ACT_RELU, 0.f, stream, nullptr, nullptr,
                            nullptr, w_scale, row_absmax, xself, ldxs);
}

extern "C" int desco_shmp_layer_pool_f16x3_f32(const float* x, int64_t ldx, const int32_t* vrowptr,
                                               const int32_t* vcol, int64_t row0, int64_t num_rows,
                                               int slots_stored, int slots_mfma, int slots_table,
                                               const int16_t* wt_planes, const float* w_scale, const float* bias,
                                               const float* ytab, int64_t ldy, int64_t ytab_row0,
                                               float* out, int64_t ldo, const uint32_t* pool_bits,
                                               const int32_t* pool_slot, float* pool_part,
                                               desco_stream_t stream) {
  if (!pool_part || !w_scale)
    return desco::fail(DESCO_EINVAL, "desco_shmp_layer_pool_f16x3_f32: pool_part / w_scale is null");
  return desco::shmp_launch("desco_shmp_layer_pool_f16x3_f32", true, x, ldx, vrowptr, vcol, row0,
                            num_rows, slots_stored, slots_mfma, slots_table, wt_planes, bias, ytab,
                            ldy, ytab_row0, out, ldo, nullptr, 0, DESCO_ACT_RELU, 0.f, stream,
                            pool_bits, pool_slot, pool_part, w_scale);
}

extern "C" int desco_shmp_layer_pool_table_f16x3_f32(const float* x, int64_t ldx, const int32_t* vrowptr,
                                                     const int32_t* vcol, int64_t row0, int64_t num_rows,
                                                     int slots_stored, int slots_mfma, int slots_table,
                                                     const int16_t* wt_planes, const float* w_scale, const float* bias,
                                                     const float* ytab, int64_t ldy, int64_t ytab_row0,
                                                     float* out, int64_t ldo, const uint32_t* pool_bits,
                                                     const int32_t* pool_slot, float* pool_part,
                                                     const float* self_coef, desco_stream_t stream) {
  if (!pool_part || !w_scale || !self_coef)
    return desco::fail(DESCO_EINVAL, "desco_shmp_layer_pool_table_f16x3_f32: pool_part / w_scale / self_coef is null");
  return desco::shmp_launch("desco_shmp_layer_pool_table_f16x3_f32", true, x, ldx, vrowptr, vcol, row0,
                            num_rows, slots_stored, slots_mfma, slots_table, wt_planes, bias, ytab,
                            ldy, ytab_row0, out, ldo, nullptr, 0, DESCO_ACT_RELU, 0.f, stream,
                            pool_bits, pool_slot, pool_part, w_scale, nullptr, nullptr, 0, self_coef);
}

extern "C" int desco_shmp_pool_tile_rows(void) { return desco::shmp_tile_rows(); }

// Row-wise Linear with K = 64 inputs (see linear64_kernel): out[i, 0:64*nb] = act(x[i, 0:64] * W^T + bias),
// w_planes = nb blocks [3][64 n][64 k].  Two column blocks per pass over x where possible.
extern "C" int desco_linear64_bf16x6_f32(const float* x, int64_t ldx, const int16_t* w_planes,
                                         int num_blocks, const float* bias, int act, float slope,
                                         float* out, int64_t ldo, int64_t num_rows,
                                         desco_stream_t stream) {
  using namespace desco;
  if (num_rows == 0 || num_blocks == 0) return 0;
  auto mis16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (!x || !w_planes || !out || num_rows < 0 || num_blocks < 0 || ldx % 4 || mis16(x) ||
      mis16(w_planes) || x == out || mis16(out) || ldo % 4)
    return fail(DESCO_EINVAL, "desco_linear64_bf16x6_f32: bad argument");
  const int64_t ntiles = (num_rows + NW * WR - 1) / (NW * WR);
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      cus = v;
  }
  const unsigned grid = (unsigned)(ntiles < cus ? ntiles : cus);
  static DeviceOnce attr_once;        // function attributes are per device
  if (!attr_once.done()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linear64_kernel<1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linear64_kernel<2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_once.mark();
  }
  constexpr size_t wave_bytes = sizeof(float) * (size_t)NW * A_FLOATS;
  for (int j = 0; j < num_blocks;) {
    const int nj = num_blocks - j >= 2 ? 2 : 1;
    Lin64Args a{x, ldx, reinterpret_cast<const short*>(w_planes) + (int64_t)j * 3 * 64 * 64,
                bias ? bias + 64 * j : nullptr, out + 64 * j, ldo, num_rows, act, slope};
    const size_t shmem = (size_t)nj * 3 * 64 * 72 * sizeof(short) + wave_bytes;
    if (nj == 2)
      hipLaunchKernelGGL(linear64_kernel<2>, dim3(grid), dim3(NW * 64), shmem, (hipStream_t)stream, a);
    else
      hipLaunchKernelGGL(linear64_kernel<1>, dim3(grid), dim3(NW * 64), shmem, (hipStream_t)stream, a);
    j += nj;
  }
  return launch_status("desco_linear64_bf16x6_f32");
}
